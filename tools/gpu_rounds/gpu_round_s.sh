#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=vllm_omni_amd/libomni_cdna4.so
A=vllm_omni_amd/csrc/build/abl
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_properties.py -q -x --timeout 600 -k "attention or attn or forward" > gpurun_out/r02s_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02s_pytest.log
AB_ROUNDS=4 timeout 900 python tools/bench_libs.py attention $A/libomni_a11.so@OMNI_ATTN_MFMA=32 $L@OMNI_ATTN_PP=0 $L@OMNI_ATTN_PP=1 2>&1 | tee gpurun_out/r02s_attn.log
