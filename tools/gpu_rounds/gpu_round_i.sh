#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A=vllm_omni_amd/csrc/build/abl
for v in ppdmamma ppbal ppboth; do
  OMNI_CDNA4_LIB=$PWD/$A/libomni_$v.so timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "pingpong or gemm" 2>&1 | tail -2
done
timeout 900 python tools/bench_libs.py gemm $A/libomni_ppbase.so $A/libomni_ppdmamma.so $A/libomni_ppbal.so $A/libomni_ppboth.so 2>&1 | tee gpurun_out/r02i_gemm.log
