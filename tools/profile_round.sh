#!/bin/bash
# Round profiles (run on the GPU box from the repo root): kernel-trace summaries of the default bench command and of
# its serial twin, HBM traffic counters (separate --pmc passes) and the MFMA-busy counters of the conv kernels.
# usage: tools/profile_round.sh r04 [unet|cae|unet4fp8|x3]   (unet4fp8: kernel-trace summary of the fp8 4-scale step only;
#        x3: kernel-trace summaries of the headline step in the pair modes f16x3 / bf16x3 + their per-layer tables)
set -o pipefail
TAG=${1:-r02}; WL=${2:-unet}
export TMPDIR=/tmp
OUT="gpurun_out/prof_${TAG}_${WL}"; mkdir -p "$OUT"
db() { find $1 -name "*.db" | head -1; }
csvf() { find $1 -name "*counter_collection.csv" | head -1; }
if [ "$WL" = unet ]; then
  rocprofv3 --kernel-trace --stats -d $OUT/g -o g -- python bench.py --steps 10 --warmup 3 --no-parity --no-cpu-baseline --no-secondary > $OUT/g.log 2>&1 && \
    python tools/rocpd_stats.py $(db $OUT/g) gpurun_out/${TAG}_bench_kernel_stats.csv > gpurun_out/${TAG}_bench_kernel_stats.txt
  python tools/rocpd_sequence.py $(db $OUT/g) > gpurun_out/${TAG}_headline_step_sequence.txt      # ordered kernels of the last step
  SP_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $OUT/s -o s -- python bench.py --steps 10 --warmup 3 --no-parity --no-cpu-baseline --no-secondary > $OUT/s.log 2>&1 && \
    python tools/rocpd_stats.py $(db $OUT/s) gpurun_out/${TAG}_bench_kernel_stats_serial.csv > gpurun_out/${TAG}_bench_kernel_stats_serial.txt
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python bench.py --steps 2 --warmup 1 --no-graph --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python bench.py --steps 2 --warmup 1 --no-graph --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/w.log 2>&1
  python tools/pmc_traffic.py $(csvf $OUT/f) $(csvf $OUT/w) $TAG > gpurun_out/${TAG}_pmc_traffic.txt
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $OUT/q1 -o q -- python bench.py --steps 2 --warmup 1 --no-graph --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/q1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/q2 -o q -- python bench.py --steps 2 --warmup 1 --no-graph --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/q2.log 2>&1
  python tools/pmc_sq.py $(csvf $OUT/q1) $(csvf $OUT/q2) > gpurun_out/${TAG}_mfma_busy.txt
  SP_LAYER_ORDER=$OUT/layers.json python bench.py --steps 3 --warmup 1 --no-parity --no-cpu-baseline --no-secondary --layers > $OUT/l.log 2>&1
  SP_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $OUT/s2 -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary --no-parity > $OUT/s2.log 2>&1 && python tools/rocpd_sequence.py $(db $OUT/s2) > $OUT/serial_seq.txt
  python tools/layer_table.py $OUT/serial_seq.txt $(csvf $OUT/f) $(csvf $OUT/w) $(csvf $OUT/q1) $OUT/layers.json > gpurun_out/${TAG}_bench_layers.txt 2>&1      # per conv launch: us, TFLOP/s, HBM MB, MFMA busy
  MIN_US=12 bash tools/probes/serial_traffic.sh bf16 unet > gpurun_out/${TAG}_bench_serial_traffic.txt 2>&1      # every kernel of the step: serial duration next to its HBM bytes
elif [ "$WL" = x3 ]; then
  for DT in f16x3 bf16x3; do
    SP_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $OUT/$DT -o s -- python bench.py --dtype $DT --steps 10 --warmup 3 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/$DT.log 2>&1 && \
      python tools/rocpd_stats.py $(db $OUT/$DT) gpurun_out/${TAG}_bench_${DT}_kernel_stats_serial.csv > gpurun_out/${TAG}_bench_${DT}_kernel_stats_serial.txt
    python bench.py --dtype $DT --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-parity --layers 2>&1 | grep -E "conv_igemm|conv_wgrad|ms_per_step" | cut -c1-400 > gpurun_out/${TAG}_bench_${DT}_layers.txt
  done
elif [ "$WL" = unet4fp8 ]; then
  rocprofv3 --kernel-trace --stats -d $OUT/g -o g -- python bench.py --workload unet4 --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/g.log 2>&1 && \
    python tools/rocpd_stats.py $(db $OUT/g) gpurun_out/${TAG}_unet4_fp8_kernel_stats.csv > gpurun_out/${TAG}_unet4_fp8_kernel_stats.txt
  python tools/rocpd_sequence.py $(db $OUT/g) > gpurun_out/${TAG}_unet4_fp8_step_sequence.txt      # ordered kernels of the last step
  python bench.py --workload unet4 --dtype fp8 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline --layers 2>&1 | grep -E "conv_igemm|conv_wgrad|quantize|partial" > gpurun_out/${TAG}_unet4_fp8_layers.txt
  bash tools/probes/serial_traffic.sh fp8 unet4 > gpurun_out/${TAG}_unet4_fp8_serial_traffic.txt 2>&1
  python tools/probes/finish_probe.py > gpurun_out/${TAG}_unet4_fp8_finish_isolated.txt 2>&1
else
  rocprofv3 --kernel-trace --stats -d $OUT/g -o g -- python bench.py --workload cae --steps 5 --warmup 2 --no-cpu-baseline > $OUT/g.log 2>&1 && \
    python tools/rocpd_stats.py $(db $OUT/g) gpurun_out/${TAG}_cae_kernel_stats.csv > gpurun_out/${TAG}_cae_kernel_stats.txt
  python tools/rocpd_sequence.py $(db $OUT/g) > gpurun_out/${TAG}_cae_step_sequence.txt      # ordered kernels of the last step
  MIN_US=30 bash tools/probes/serial_traffic.sh bf16 cae > gpurun_out/${TAG}_cae_serial_traffic.txt 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python bench.py --workload cae --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing > $OUT/f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python bench.py --workload cae --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing > $OUT/w.log 2>&1
  python tools/pmc_traffic.py $(csvf $OUT/f) $(csvf $OUT/w) ${TAG}_cae cae_ "bench.py --workload cae --steps 1 --warmup 1 --no-graph, CAE B=4 1x28x128x128 bf16" > gpurun_out/${TAG}_cae_pmc_traffic.txt
fi
rm -rf "$OUT"
ls -la gpurun_out | grep ${TAG}_ | tail -12
