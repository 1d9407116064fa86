#!/bin/bash
# Round 6, second session: the small-batch launch savers (in-launch split-K reduce, paired AdaLN) — new GPU tests, then a same-box
# A/B of whole 60-layer forwards / config-1 images through a -DOMNI_DEV build whose knobs read the environment (the product has
# none), then the per-kernel trace of the config-1 step with the product library.  The dev library = gemm.hip + dit_forward.hip compiled
# with -DOMNI_DEV (tools/build_variants.sh, both objects linked into libomni_devknobs2.so); OMNI_GEMM_SPLITK_INLAUNCH = the largest
# split factor reduced in the launch (0 = never, 8 = always), OMNI_DIT_ADALN_PAIR = 0 / 1.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06b}
timeout 1500 python -m pytest tests/test_gpu_splitk_inlaunch.py tests/test_gpu_ops.py tests/test_gpu_dit_forward.py tests/test_gpu_pipeline.py -x -q -m gpu > $OUT/${TAG}_new_tests.log 2>&1
tail -5 $OUT/${TAG}_new_tests.log
L=$OUT/${TAG}_ab_smallm.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs2.so
for rep in 1 2 3; do
  for mode in "0 0" "8 0" "0 1" "8 1"; do
    set -- $mode
    for spec in "256 1" "512 1" "256 4"; do
      set -- $mode $spec
      echo "px $3 R $4 splitk_inlaunch $1 adaln_pair $2 (rep $rep): $(OMNI_GEMM_SPLITK_INLAUNCH=$1 OMNI_DIT_ADALN_PAIR=$2 timeout 300 python tools/time_step.py $3 60 10 $4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200)" >> $L
    done
    set -- $mode
    echo "config1 splitk_inlaunch $1 adaln_pair $2 (rep $rep): $(OMNI_GEMM_SPLITK_INLAUNCH=$1 OMNI_DIT_ADALN_PAIR=$2 timeout 300 python tools/time_config1.py 2>&1 | grep -v amdgpu.ids | tail -1)" >> $L
  done
done
unset OMNI_DEV_LIB
cat $L
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_step_256px_R1 -o p -- python tools/time_step.py 256 60 5 1 > $OUT/prof_${TAG}_step_256px_R1.log 2>&1
python tools/step_profile_table.py $OUT/prof_${TAG}_step_256px_R1/*.db 256 1 > $OUT/${TAG}_step_table_256px_R1.txt 2>&1
python tools/summarize_prof.py $OUT $TAG > $OUT/${TAG}_profiles_summary.txt 2>&1
find $OUT -name "*.db" -size +30M -delete
cat $OUT/${TAG}_step_table_256px_R1.txt | head -40
