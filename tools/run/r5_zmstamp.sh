set -o pipefail
bash tools/build_variant.sh zmstamps "-DSP_ZM_STAMPS" sp_conv_zm.hip 2>&1 | tail -1
export SP_LIB_PATH=$PWD/stroke-prediction_amd/lib/variants/zmstamps.so
for a in "48 16 92 fwd" "32 32 60 fwd" "16 16 126 fwd" "48 16 92 dgrad" "16 32 62 fwd" "32 32 60 dgrad"; do
  timeout -k 10 200 python tools/stamp_zm.py $a 2>&1 | tail -8
done
