#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
AB_ROUNDS=3 timeout 900 python tools/bench_libs.py qkv $A/libomni_q0.so $A/libomni_qnold.so $A/libomni_qnomath.so 2>&1 | tee gpurun_out/r02ae_qkv.log
