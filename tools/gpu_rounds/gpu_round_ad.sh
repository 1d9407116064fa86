#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize_properties.py tests/test_gpu_dit_forward.py -q -x --timeout 600 > gpurun_out/r02ad_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02ad_pytest.log
AB_ROUNDS=4 timeout 900 python tools/bench_libs.py qkv $A/libomni_qbase.so $A/libomni_qskipv.so 2>&1 | tee gpurun_out/r02ad_qkv.log
