#!/bin/bash
# Round 6: calibration sweep of the GEMM tail-split factor per block-GEMM shape (same box, dev library).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
L=$OUT/r06_tail_split_sweep.log
export OMNI_DEV_LIB=$PWD/vllm_omni_amd/csrc/build/abl/libomni_devknobs.so
: > $L
for rows in "8192 128" "16384 256" "24576 384" "32768 512" "40960 640" "32768 128"; do
  for nk in "3072 3072" "3072 12288" "9216 3072" "12288 3072"; do
    timeout 300 python tools/bench_tail_split.py $rows $nk 2>&1 | grep "^M " >> $L
  done
done
cat $L
