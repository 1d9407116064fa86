#!/bin/bash
# dev: MLP-up's thin-tail split when the launch is ONE full round + a thin tail (1280-row batches: 288 tiles), -DOMNI_DEV dit_forward, same box.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
L=$OUT/r06l_ab_mlpup_tail.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs4.so
for rep in 1 2 3; do
  for k in 0 1; do
    for spec in "384 1" "256 2" "256 1" "512 1"; do
      set -- $spec
      echo "px $1 R $2 mlpup_tail $k (rep $rep): $(OMNI_DIT_MLPUP_TAIL=$k timeout 300 python tools/time_step.py $1 60 10 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200)" >> $L
    done
  done
done
cat $L
