#!/bin/bash
# Round 6, third session: the split-K finish folded into the following AdaLN (ABI v13) — new GPU tests and the tests of the paths it
# touches, then a same-box A/B of whole 60-layer forwards / config-1 images through a -DOMNI_DEV build of gemm.hip + dit_forward.hip
# (libomni_devknobs3.so): OMNI_DIT_FUSE_FINISH = 0 off / 1 out-projection -> norm2 / 2 MLP-down -> next norm1 / 3 both (the product),
# OMNI_GEMM_SPLITK_MIN_KT = K-tiles per split-K piece at least (4 = the product; 12: the out-projection splits 4-way instead of 6).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06c}
timeout 1500 python -m pytest tests/test_gpu_finish_adaln.py tests/test_gpu_splitk_inlaunch.py tests/test_gpu_teacache.py tests/test_gpu_dit_forward.py tests/test_gpu_pipeline.py tests/test_gpu_ops.py -x -q -m gpu > $OUT/${TAG}_fused_finish_tests.log 2>&1
tail -15 $OUT/${TAG}_fused_finish_tests.log
L=$OUT/${TAG}_ab_fused_finish.log; : > $L
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs3.so
for rep in 1 2 3; do
  for mode in "0 4" "1 4" "2 4" "3 4" "3 12"; do
    set -- $mode
    echo "config1 fuse_finish $1 min_kt $2 (rep $rep): $(OMNI_DIT_FUSE_FINISH=$1 OMNI_GEMM_SPLITK_MIN_KT=$2 timeout 300 python tools/time_config1.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-60)" >> $L
    for spec in "256 2" "384 1" "512 1"; do
      set -- $mode $spec
      echo "px $3 R $4 fuse_finish $1 min_kt $2 (rep $rep): $(OMNI_DIT_FUSE_FINISH=$1 OMNI_GEMM_SPLITK_MIN_KT=$2 timeout 300 python tools/time_step.py $3 60 10 $4 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130)" >> $L
    done
  done
done
unset OMNI_DEV_LIB
cat $L
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_step_256px_R1 -o p -- python tools/time_step.py 256 60 5 1 > $OUT/prof_${TAG}_step_256px_R1.log 2>&1
python tools/step_profile_table.py $OUT/prof_${TAG}_step_256px_R1/*.db 256 1 > $OUT/${TAG}_step_table_256px_R1.txt 2>&1
find $OUT -name "*.db" -size +30M -delete
cat $OUT/${TAG}_step_table_256px_R1.txt | head -40
