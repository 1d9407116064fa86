#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
timeout 900 python tools/bench_libs.py gemm $A/libomni_ppfull.so $A/libomni_ppnodma.so $A/libomni_ppnovm.so $A/libomni_ppnobar.so $A/libomni_ppnodmanobar.so $A/libomni_ppnodmanoread.so 2>&1 | tee gpurun_out/r02h_gemm.log
