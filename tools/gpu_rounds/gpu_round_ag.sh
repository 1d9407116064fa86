#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_r02ag_c1 -o c1 -- python tools/run_config1.py 3 > $OUT/prof_r02ag_c1.log 2>&1; echo rc=$?
python tools/summarize_prof.py gpurun_out r02ag 2>&1 | grep -v "^W2026" | head -40
