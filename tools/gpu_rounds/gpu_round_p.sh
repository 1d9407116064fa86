#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
A=vllm_omni_amd/csrc/build/abl
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_properties.py -q -x --timeout 600 -k "attention or attn or forward" > gpurun_out/r02p_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02p_pytest.log
AB_ROUNDS=4 timeout 900 python tools/bench_libs.py attention $A/libomni_a11.so@OMNI_ATTN_MFMA=32 $A/libomni_a00.so $A/libomni_a10.so $A/libomni_a11.so 2>&1 | tee gpurun_out/r02p_attn.log
for M in 32 16; do
  OMNI_ATTN_MFMA=$M timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/pmc_r02p_attn${M}_sq -o pmc -- python tools/run_kernel.py attention6 5 > $OUT/pmc_r02p_attn${M}_sq.log 2>&1
  OMNI_ATTN_MFMA=$M timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES -d $OUT/pmc_r02p_attn${M}_b -o pmc -- python tools/run_kernel.py attention6 5 > $OUT/pmc_r02p_attn${M}_b.log 2>&1
done
python tools/summarize_prof.py $OUT/pmc_r02p_attn32_sq $OUT/pmc_r02p_attn16_sq $OUT/pmc_r02p_attn32_b $OUT/pmc_r02p_attn16_b 2>&1 | grep -v "^==" | grep "flash" 
