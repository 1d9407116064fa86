#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dit_forward.py tests/test_gpu_pipeline.py tests/test_gpu_teacache.py tests/test_gpu_sequence_parallel.py -q -x --timeout 600 > gpurun_out/r02ai_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02ai_pytest.log
timeout 600 rocprofv3 --kernel-trace --stats -T -d $OUT/prof_r02ai_c1 -o c1 -- python tools/run_config1.py 3 > $OUT/prof_r02ai_c1.log 2>&1; echo rc=$?
python tools/summarize_prof.py gpurun_out r02ai 2>&1 | grep -v "^W2026" | grep "gemm\|total kernel" | head -14
