#!/bin/bash
# Round 6, third session: the finish kernels' partial-load width follows the split factor (variant library libomni_finw.so = the
# product objects with gemm.hip rebuilt) against the product library, same box: config 1 (warm table) and one 256^2 / 384^2 step.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
L=$OUT/r06i_ab_finish_width.log; : > $L
for rep in 1 2 3; do
  for v in product finw; do
    if [ $v = finw ]; then export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_finw.so; else unset OMNI_DEV_LIB; fi
    echo "config1 warm $v (rep $rep): $(timeout 300 python tools/time_config1.py warm 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-60)" >> $L
    for spec in "256 1" "512 1"; do
      set -- $spec
      echo "px $1 R $2 $v (rep $rep): $(timeout 300 python tools/time_step.py $1 60 10 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120)" >> $L
    done
  done
done
cat $L
