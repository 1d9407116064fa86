#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_teacache.py tests/test_gpu_engine.py tests/test_gpu_pipeline.py -q -s --timeout 600 > gpurun_out/r02e.log 2>&1; echo "rc=$?"; grep -E "graph=|passed|failed|Error|error|assert" gpurun_out/r02e.log | tail -40
