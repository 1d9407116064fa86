#!/bin/bash
# Closing pass of a round 2/2 (first used in round 5): the bench as the driver runs it, then rocprofv3 --kernel-trace --stats of the bench command and the PMC
# passes (separate runs) of the roofline launch and the attention launch, all on the library that ships.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r05}
timeout 1500 python bench.py > $OUT/${TAG}_bench_n1_default.json 2> $OUT/${TAG}_bench_n1_default.err
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_bench -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-engine > $OUT/prof_${TAG}_bench.log 2>&1
for K in roofline attention10; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_${TAG}_${K}_sq -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_sq.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_${TAG}_${K}_fetch -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_${TAG}_${K}_write -o pmc -- python tools/run_kernel.py $K 5 > $OUT/pmc_${TAG}_${K}_write.log 2>&1
done
sha256sum vllm_omni_amd/libomni_cdna4.so | cut -c1-16 > $OUT/${TAG}_library_sha.txt
head -c 1500 $OUT/${TAG}_bench_n1_default.json; echo; tail -2 $OUT/${TAG}_bench_n1_default.err
