#!/bin/bash
# Same-box A/B of the MID tail split (tails of a quarter to half a round of tiles: 2 or 3 pieces per tile, one sub-round).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06b}
timeout 900 python -m pytest tests/test_gpu_tail_split.py -x -q -m gpu > $OUT/${TAG}_tail_tests.log 2>&1
tail -3 $OUT/${TAG}_tail_tests.log
L=$OUT/${TAG}_ab_mid_tail_split.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs2.so
for rep in 1 2; do
  for v in 0 1; do
    for spec in "512 1 8" "512 2 8" "768 4 4" "768 5 4" "1024 3 4" "1024 4 4" "1024 5 3" "1536 1 4" "2048 3 2"; do
      set -- $spec
      echo "px $1 R $2 mid_tail_split $v (rep $rep): $(OMNI_GEMM_TAIL_MID=$v timeout 400 python tools/time_step.py $1 60 $3 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c64-230)" >> $L
    done
  done
done
cat $L
