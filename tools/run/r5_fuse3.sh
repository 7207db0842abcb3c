set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_kernels.py tests/test_gpu_unet.py -x -q > gpurun_out/t5.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/t5.log
for v in "" ""; do
  echo "== $v"; env $v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
export TMPDIR=/tmp
SP_OVERLAP=0 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary --no-parity > gpurun_out/s.log 2>&1 && python tools/rocpd_sequence.py $(find gpurun_out/prof_s -name "*.db" | head -1) > gpurun_out/r5_seq.txt; rm -rf gpurun_out/prof_s
grep -v "^#" gpurun_out/r5_seq.txt | grep "prep\|finish\|head_grad\|Fill\|finalize" | awk '{printf "%s %s\n", $2, substr($0, index($0,$3), 50)}' | sort -k2 | uniq -c -f1 | head; grep -v "^#" gpurun_out/r5_seq.txt | grep "prep_folded" | awk '{printf "%s ", $2}'
