#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout 1500 python bench.py > gpurun_out/r02t_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r02t_bench.log | cut -c1-3000
timeout 900 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_r02t_bench -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/prof_r02t_bench.log 2>&1; echo "prof rc=$?"
python tools/summarize_prof.py gpurun_out r02t 2>&1 | head -30
