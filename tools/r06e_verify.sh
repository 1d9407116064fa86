#!/bin/bash
# Round 6, third session, last library change (GEMV: DPP wave reductions): the GEMV / forward / pipeline tests, config 1 cold and warm,
# the per-forward-GEMV step (no table), then the full suite and the default bench as the driver runs it.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06e}
L=$OUT/${TAG}_config1_and_steps.log; : > $L
for rep in 1 2 3; do
  for m in cold warm; do
    echo "config1 $m (rep $rep): $(timeout 300 python tools/time_config1.py $m 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-90)" >> $L
  done
  echo "px 256 R 1 (rep $rep): $(timeout 300 python tools/time_step.py 256 60 10 1 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130)" >> $L
done
cat $L
bash tools/closing_pass_tests.sh $TAG > /dev/null 2>&1
tail -8 $OUT/${TAG}_pytest_gpu_full.log
timeout 1500 python bench.py > $OUT/${TAG}_bench_n1_default.json 2> $OUT/${TAG}_bench_n1_default.err
sha256sum vllm_omni_amd/libomni_cdna4.so | cut -c1-16 > $OUT/${TAG}_library_sha.txt
head -c 600 $OUT/${TAG}_bench_n1_default.json
