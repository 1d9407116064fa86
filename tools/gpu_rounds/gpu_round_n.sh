#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
AB_ROUNDS=4 timeout 600 python tools/bench_libs.py attention $A/libomni_atbase.so $A/libomni_atmfma16.so 2>&1 | tee gpurun_out/r02n_attn.log
