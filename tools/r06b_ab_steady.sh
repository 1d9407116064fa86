#!/bin/bash
# Same-box A/B: the whole-launch split-K instance on the general K-loop (0) vs the steady-state K-loop (1); the product library
# first re-runs the round-6b tests (opt-in in-launch reduce, paired AdaLN).  Variant libraries:
#   tools/build_variants.sh gemm steady0 "-DOMNI_SPLITK_STEADY_MIN_KT=1000000" steady1 "-DOMNI_SPLITK_STEADY_MIN_KT=1"
# (the product: 12 — pieces of at least 12 K-tiles take the steady-state loop)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06b}
timeout 900 python -m pytest tests/test_gpu_splitk_inlaunch.py tests/test_gpu_ops.py -x -q -m gpu > $OUT/${TAG}_new_tests_optin.log 2>&1
tail -3 $OUT/${TAG}_new_tests_optin.log
L=$OUT/${TAG}_ab_splitk_steady_loop.log; : > $L
for rep in 1 2 3; do
  for v in 0 1; do
    export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_steady$v.so
    for spec in "256 1" "384 1" "512 1" "256 2" "256 4"; do
      set -- $spec
      echo "px $1 R $2 splitk_steady $v (rep $rep): $(timeout 300 python tools/time_step.py $1 60 10 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c64-230)" >> $L
    done
    echo "config1 splitk_steady $v (rep $rep): $(timeout 300 python tools/time_config1.py 2>&1 | grep -v amdgpu.ids | tail -1)" >> $L
  done
done
cat $L
