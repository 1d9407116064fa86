#!/bin/bash
# dev: 3-way K-split over TWO rounds for a long-K launch of 129 .. 170 tiles (the MLP down-projection of one 576^2 / 640^2 request or two
# 448^2 ones), OMNI_GEMM_SPLITK_LONGK = 0 / 1, -DOMNI_DEV gemm.hip, same box, 60 layers.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
L=$OUT/r06n_ab_long_k_split.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs5.so
for rep in 1 2 3; do
  for k in 0 1; do
    for spec in "576 1" "640 1" "448 2" "512 1"; do
      set -- $spec
      echo "px $1 R $2 long_k_split $k (rep $rep): $(OMNI_GEMM_SPLITK_LONGK=$k timeout 300 python tools/time_step.py $1 60 6 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200)" >> $L
    done
  done
done
cat $L
