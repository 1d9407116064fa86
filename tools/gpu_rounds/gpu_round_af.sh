#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
AB_ROUNDS=4 timeout 900 python tools/bench_libs.py attention $A/libomni_w0.so $A/libomni_wpair.so 2>&1 | tee gpurun_out/r02af_attn.log
