#!/bin/bash
# Round 6: same-box A/B of the two thin-last-round splits (GEMM tail split, attention short-block split) on whole 60-layer
# forwards, through a -DOMNI_DEV build of gemm.hip / attention_w64.hip whose knobs read the environment (the product has none).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=${1:-$OUT/r06_ab_splits.log}
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs.so
: > $L
for spec in ${SPECS:-"2048 1" "1024 1" "1024 2" "1024 3" "1024 4" "1024 5"}; do
  set -- $spec
  for rep in 1 2; do
    for mode in "0 0" "1 0" "0 1" "1 1"; do
      set -- $spec $mode
      echo "px $1 R $2 gemm_tail_split $3 attn_split $4 (rep $rep): $(OMNI_GEMM_TAILSPLIT=$3 OMNI_ATTN_SPLIT=$4 timeout 600 python tools/time_step.py $1 60 3 $2 2>&1 | grep -v amdgpu.ids | tail -1)" >> $L
    done
  done
done
cat $L
