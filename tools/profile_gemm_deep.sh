#!/bin/bash
# Deep counter passes on the default GEMM kernel (MLP-up shape).  usage: tools/profile_gemm_deep.sh <tag>
TAG=${1:-r01c}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_${TAG}_gemm_${name} -o pmc -- python tools/run_kernel.py gemm_mlp_up 5 > $OUT/pmc_${TAG}_gemm_${name}.log 2>&1; }
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
run tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run sq SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS
