#!/bin/bash
# Profiling recipe for one round (run on the GPU box through gpurun).  Outputs land in gpurun_out/.
# usage: tools/profile_round.sh <tag>
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd $PWD
# 1) per-kernel time of the SAME command as the judged bench
timeout 900 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_bench -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-engine > $OUT/prof_${TAG}_bench.log 2>&1
# 2) counters: separate passes, --kernel-trace only
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
for K in roofline attention10; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_${TAG}_${K}_sq -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_${TAG}_${K}_fetch -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_${TAG}_${K}_write -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_write.log 2>&1
done
find $OUT -name "*.csv" | head -40
