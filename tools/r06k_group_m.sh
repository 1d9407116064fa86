#!/bin/bash
# dev: GROUP_M (row tiles per L2 band of the GEMM block remap) at the small-step-batch shapes; -DOMNI_DEV library, same box.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
L=$OUT/r06k_group_m.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs3.so
for rep in 1 2; do
  for gm in 4 2 8 16; do
    for spec in "1024 1" "1024 2" "512 1" "2048 1"; do
      set -- $spec
      echo "px $1 R $2 group_m $gm (rep $rep): $(OMNI_GEMM_GROUP_M=$gm timeout 300 python tools/time_step.py $1 60 4 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-140)" >> $L
    done
  done
done
cat $L
