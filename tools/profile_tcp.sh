#!/bin/bash
# TCP (L1) request counters for GEMM variants.  usage: tools/profile_tcp.sh <tag>
TAG=${1:-r01d}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
for V in 0 1; do
  OMNI_GEMM_VARIANT=$V timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum -d $OUT/pmc_${TAG}_gemm_v${V}_tcp -o pmc -- python tools/run_kernel.py gemm_mlp_up 5 > $OUT/pmc_${TAG}_gemm_v${V}_tcp.log 2>&1
done
