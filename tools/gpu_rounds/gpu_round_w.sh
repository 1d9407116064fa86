#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
TAG=r02w
timeout 900 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_bench -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/prof_${TAG}_bench.log 2>&1; echo "prof rc=$?"
for K in roofline attention6; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/pmc_${TAG}_${K}_sq -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_${TAG}_${K}_fetch -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_${TAG}_${K}_write -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_write.log 2>&1
done
timeout 1500 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-700
