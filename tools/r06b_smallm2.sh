#!/bin/bash
# Same-box A/B of the in-launch split-K reduce by the largest split factor it takes (dev knob; 0 = the two-kernel path everywhere).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06b}
L=$OUT/${TAG}_ab_inlaunch_by_factor.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs2.so
for rep in 1 2 3; do
  for ns in 0 2 4 8; do
    for spec in "256 1" "384 1" "512 1" "256 2" "256 4"; do
      set -- $spec
      echo "px $1 R $2 in-launch up to nsplit $ns (rep $rep): $(OMNI_GEMM_SPLITK_INLAUNCH=$ns timeout 300 python tools/time_step.py $1 60 10 $2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150)" >> $L
    done
    echo "config1 in-launch up to nsplit $ns (rep $rep): $(OMNI_GEMM_SPLITK_INLAUNCH=$ns timeout 300 python tools/time_config1.py 2>&1 | grep -v amdgpu.ids | tail -1)" >> $L
  done
done
cat $L
