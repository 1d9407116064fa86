#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
AB_ROUNDS=4 timeout 900 python tools/bench_libs.py gemm $A/libomni_g0.so $A/libomni_gagpr.so $A/libomni_gord1.so $A/libomni_gord2.so $A/libomni_gnoprio.so 2>&1 | tee gpurun_out/r02u_gemm.log
