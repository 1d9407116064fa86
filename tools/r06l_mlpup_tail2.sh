#!/bin/bash
# dev: MLP-up's thin-tail split by the number of full rounds in front of the tail (OMNI_DIT_MLPUP_TAIL = largest number of rounds taken).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
L=$OUT/r06l_ab_mlpup_tail_rounds.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs4.so
for rep in 1 2 3; do
  for k in 0 1 2 3; do
    for spec in "384 1" "576 1" "704 1"; do
      set -- $spec
      echo "px $1 R $2 mlpup_tail_rounds $k (rep $rep): $(OMNI_DIT_MLPUP_TAIL=$k timeout 300 python tools/time_step.py $1 60 6 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200)" >> $L
    done
  done
done
cat $L
