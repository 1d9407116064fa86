#!/bin/bash
# Same-box A/B: cache policy of the split-K partial stores (0 plain, 1 nt, 2 sc1) x of the finish kernels' partial loads (0 plain, 1 nt)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06b}
L=$OUT/${TAG}_ab_splitk_partial_cache_policy.log; : > $L
for rep in 1 2 3; do
  for v in 00 10 20 01 11 21; do
    export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_pol$v.so
    for spec in "256 1" "512 1" "256 4" "1024 2"; do
      set -- $spec
      echo "px $1 R $2 store/load policy $v (rep $rep): $(timeout 300 python tools/time_step.py $1 60 8 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c64-230)" >> $L
    done
    echo "config1 store/load policy $v (rep $rep): $(timeout 300 python tools/time_config1.py 2>&1 | grep -v amdgpu.ids | tail -1)" >> $L
  done
done
cat $L
