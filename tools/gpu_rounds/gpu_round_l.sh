#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_properties.py -q -x --timeout 600 > gpurun_out/r02l_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02l_pytest.log
AB_ROUNDS=4 timeout 600 python tools/bench_libs.py gemm $A/libomni_pp32.so $A/libomni_pp16.so 2>&1 | tee gpurun_out/r02l_gemm.log
AB_SKIP_ATTN=1 AB_ROUNDS=3 timeout 600 python tools/bench_ab.py > gpurun_out/r02l_ab.log 2>&1; tail -14 gpurun_out/r02l_ab.log
