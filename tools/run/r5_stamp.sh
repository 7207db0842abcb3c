set -o pipefail
STATS=0 timeout -k 10 500 python tools/stamp_conv.py 64 64 27 fwd 2>&1 | tail -12
