#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=vllm_omni_amd/libomni_cdna4.so
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_properties.py tests/test_gpu_dit_forward.py -q -x --timeout 600 > gpurun_out/r02o_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r02o_pytest.log
AB_ROUNDS=4 timeout 600 python tools/bench_libs.py attention $L@OMNI_ATTN_MFMA=32 $L@OMNI_ATTN_MFMA=16 2>&1 | tee gpurun_out/r02o_attn.log
