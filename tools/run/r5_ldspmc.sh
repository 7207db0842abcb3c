set -o pipefail
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/prof_lds; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/q1 -o q -- python bench.py --steps 2 --warmup 1 --no-graph --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/q1.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/q2 -o q -- python bench.py --steps 2 --warmup 1 --no-graph --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/q2.log 2>&1
python tools/pmc_sq.py $(find $OUT/q1 -name "*counter_collection.csv" | head -1) $(find $OUT/q2 -name "*counter_collection.csv" | head -1) --match conv_zm3 > gpurun_out/r05_lds_counters.txt 2>&1
tail -3 $OUT/q2.log
rm -rf $OUT
head -80 gpurun_out/r05_lds_counters.txt | cut -c1-150
