#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
AB_ROUNDS=4 timeout 600 python tools/bench_libs.py gemm $A/libomni_ppbase.so $A/libomni_ppmfma16.so 2>&1 | tee gpurun_out/r02k_gemm.log
