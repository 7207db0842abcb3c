set -e
mkdir -p gpurun_out
python tools/f8_dirderiv.py 156 f32 fp8 > gpurun_out/f8cos_default.txt 2>&1
SP_F8_DGRAD=0 python tools/f8_dirderiv.py 156 f32 fp8 > gpurun_out/f8cos_nodgrad.txt 2>&1
SP_F8_WGRAD=0 python tools/f8_dirderiv.py 156 f32 fp8 > gpurun_out/f8cos_nowgrad.txt 2>&1
SP_F8_DGRAD=0 SP_F8_WGRAD=0 python tools/f8_dirderiv.py 156 f32 fp8 > gpurun_out/f8cos_fwdonly.txt 2>&1
SP_F8_DZ=e4m3 python tools/f8_dirderiv.py 156 f32 fp8 > gpurun_out/f8cos_dze4m3.txt 2>&1
python tools/f8_dirderiv.py 156 f32 bf16 > gpurun_out/f8cos_bf16.txt 2>&1
grep -h "cos" gpurun_out/f8cos_default.txt | tail -25
