#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_teacache.py -q -s -x --timeout 300 > gpurun_out/r02d_tc.log 2>&1; echo "teacache rc=$?"; tail -40 gpurun_out/r02d_tc.log
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_gpu_teacache.py > gpurun_out/r02d_all.log 2>&1; echo "all rc=$?"; tail -5 gpurun_out/r02d_all.log
