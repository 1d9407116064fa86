#!/bin/bash
# Round 6, first GPU call: per-kernel breakdowns (rocprofv3 --kernel-trace --stats) of the three step shapes the round-5 verdict
# names as unproven or far from their roof: a 2048^2 true-CFG step (config 5 geometry), a lone 1024^2 request (R = 1), and
# BASELINE config 1 (256^2, 4 steps, batch 1).  Outputs under gpurun_out/.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r06}
for spec in "2048 60 2 1" "1024 60 3 1" "1024 60 2 5" "256 60 5 1"; do
  set -- $spec
  name=step_$1px_R$4
  timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_$name -o p -- python tools/time_step.py $1 $2 $3 $4 > $OUT/prof_${TAG}_$name.log 2>&1
  tail -1 $OUT/prof_${TAG}_$name.log
done
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_${TAG}_config1 -o p -- python tools/run_config1.py 5 > $OUT/prof_${TAG}_config1.log 2>&1
python tools/summarize_prof.py $OUT $TAG > $OUT/${TAG}_first_profiles_summary.txt 2>&1
