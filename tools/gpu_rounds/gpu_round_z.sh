#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_edit.py tests/test_gpu_pipeline.py -q --timeout 600 -s > gpurun_out/r02z_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^$" gpurun_out/r02z_pytest.log | tail -14
