#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
timeout 600 python tools/bench_libs.py attention $A/libomni_base.so $A/libomni_stag.so $A/libomni_spread.so 2>&1 | tee gpurun_out/r02g_attn.log
timeout 600 python tools/bench_libs.py gemm $A/libomni_ppfull.so $A/libomni_ppnomfma.so $A/libomni_ppnodma.so 2>&1 | tee gpurun_out/r02g_gemm.log
