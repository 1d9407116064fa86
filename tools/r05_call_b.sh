#!/bin/bash
# Round 5, GPU call B: the new 2048^2 parity tests, the N-rank bench control flow (2 and 8 ranks on one device), fp8 GEMMs on the
# round-4 loop (product) vs the restructured loop (variant fp8new).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_res2048_parity.py -m gpu -x -q -s > $OUT/r05_res2048_parity.log 2>&1
timeout 900 python -m pytest tests/test_gpu_multirank_bench.py -m gpu -x -q -k "bench" > $OUT/r05_multirank.log 2>&1
L=vllm_omni_amd/csrc/build/abl
( echo "== product (fp8 instance on the round-2..4 loop)"; timeout 120 python tools/bench_fp8_gemm.py 20
  echo "== fp8new (-DOMNI_PP_SCHED=17: fp8 instance on the restructured loop)"; OMNI_DEV_LIB=$L/libomni_fp8new.so timeout 120 python tools/bench_fp8_gemm.py 20
  echo "== product again"; timeout 120 python tools/bench_fp8_gemm.py 20
  echo "== fp8new again"; OMNI_DEV_LIB=$L/libomni_fp8new.so timeout 120 python tools/bench_fp8_gemm.py 20 ) > $OUT/r05_fp8_gemm_old_vs_new_loop.log 2>&1
tail -15 $OUT/r05_res2048_parity.log; tail -5 $OUT/r05_multirank.log; cat $OUT/r05_fp8_gemm_old_vs_new_loop.log | grep -v amdgpu.ids
