#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for hw in 256 512; do for k in 0 1; do OMNI_GEMM_SPLITK=$k python tools/time_config1.py $hw 2>/dev/null | tail -1; done; done | tee gpurun_out/r02aj_c1.log
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dit_forward.py tests/test_gpu_pipeline.py tests/test_gpu_engine.py -q -x --timeout 600 2>&1 | tail -2
