#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
B=$A/libomni_gbase.so
AB_ROUNDS=3 timeout 900 python tools/bench_libs.py gemm $B $B@OMNI_GEMM_GROUP_M=2 $B@OMNI_GEMM_GROUP_M=6 $B@OMNI_GEMM_GROUP_M=8 $B@OMNI_GEMM_GROUP_M=16 $A/libomni_gaux1.so $A/libomni_gaux16.so $A/libomni_gaux17.so 2>&1 | tee gpurun_out/r02ac_gemm.log
