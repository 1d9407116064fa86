#!/bin/bash
# Round 5, GPU call C: the cleaned gemm.hip (ablation switches removed, dev families under csrc/dev/) against the round-4 shipped
# library: bit identity over the 17 shape classes, timing; fp8 GEMMs on the old vs the restructured loop; the GEMM / fp8 tests.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=vllm_omni_amd/csrc/build/abl
P=vllm_omni_amd/libomni_cdna4.so
( timeout 60 ./tools/probe/pp_probe --sweep $L/libomni_r4ref.so $P
  timeout 40 ./tools/probe/pp_probe --iters 20 $L/libomni_r4ref.so $P $L/libomni_r4ref.so $P ) > $OUT/r05_gemm_cleanup_identity.log 2>&1
( echo "== r4ref (fp8 instance on the round-2..4 loop)"; OMNI_DEV_LIB=$L/libomni_r4ref.so timeout 120 python tools/bench_fp8_gemm.py 20
  echo "== fp8new (fp8 instance on the restructured loop)"; OMNI_DEV_LIB=$L/libomni_fp8new.so timeout 120 python tools/bench_fp8_gemm.py 20
  echo "== r4ref again"; OMNI_DEV_LIB=$L/libomni_r4ref.so timeout 120 python tools/bench_fp8_gemm.py 20
  echo "== fp8new again"; OMNI_DEV_LIB=$L/libomni_fp8new.so timeout 120 python tools/bench_fp8_gemm.py 20 ) > $OUT/r05_fp8_gemm_old_vs_new_loop.log 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -m gpu -x -q -k "not benchmarked_width" > $OUT/r05_gemm_tests_cleaned.log 2>&1
grep -c " ok" $OUT/r05_gemm_cleanup_identity.log; grep "DIFFERS\|sweep\|round" $OUT/r05_gemm_cleanup_identity.log | tail -12
grep -v amdgpu.ids $OUT/r05_fp8_gemm_old_vs_new_loop.log; tail -3 $OUT/r05_gemm_tests_cleaned.log
