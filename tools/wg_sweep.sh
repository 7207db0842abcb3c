for layer in "48 16 92" "32 64 29" "64 64 27" "64 32 27" "16 16 90" "32 32 48"; do
  for cfg in "0 0" "0 128" "0 256" "0 384" "4 256" "4 384" "8 256"; do
    set -- $cfg
    if [ "$2" = "0" ]; then SP_WGRAD_ROWS=$1 python tools/bench_wgrad.py $layer 2>&1 | tail -1 | sed "s/^/rows=$1 /"
    else SP_WGRAD_ROWS=$1 SP_WGRAD_BLOCKS=$2 python tools/bench_wgrad.py $layer 2>&1 | tail -1 | sed "s/^/rows=$1 /"; fi
  done
done
