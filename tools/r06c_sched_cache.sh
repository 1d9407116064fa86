#!/bin/bash
# Round 6, third session: the per-schedule modulation-table cache + the GEMV piece-count template — tests, then config 1 cold / warm.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06c}
timeout 1200 python -m pytest tests/test_gpu_dit_forward.py tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_engine.py -x -q -m gpu > $OUT/${TAG}_sched_cache_tests.log 2>&1
tail -6 $OUT/${TAG}_sched_cache_tests.log
L=$OUT/${TAG}_config1_cold_warm.log; : > $L
for rep in 1 2 3; do
  for m in cold warm; do
    echo "config1 $m (rep $rep): $(timeout 300 python tools/time_config1.py $m 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-90)" >> $L
  done
done
cat $L
