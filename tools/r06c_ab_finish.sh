#!/bin/bash
# Round 6, third session: same-box A/B of the split-K finish kernels' row batching (OMNI_FIN_BATCH = rows per thread: 4 = the
# product, 2, 1 -> 8 / 16 / 32 workgroups per tile) and load order (OMNI_FIN_REORDER = 1: a row's partial loads are issued before
# its map-dependent gate / residual loads).  Variant libraries: tools/build_variants.sh gemm fin<B><R> "-DOMNI_FIN_BATCH=B -DOMNI_FIN_REORDER=R".
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06c}
L=$OUT/${TAG}_ab_finish_variants.log; : > $L
for rep in 1 2 3; do
  for v in fin40 fin41 fin20 fin10 fin11; do
    export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_$v.so
    echo "config1 $v (rep $rep): $(timeout 300 python tools/time_config1.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-60)" >> $L
    for spec in "256 2" "384 1"; do
      set -- $spec
      echo "px $1 R $2 $v (rep $rep): $(timeout 300 python tools/time_step.py $1 60 10 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120)" >> $L
    done
  done
done
cat $L
