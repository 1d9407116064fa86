#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/time_config1.py 256 2>/dev/null | tail -1
OMNI_GEMM_SPLITK_OVERSUB=1.2 OMNI_GEMM_SPLITK_TILES=160 python tools/time_config1.py 256 2>/dev/null | tail -1
OMNI_GEMM_SPLITK_OVERSUB=2.3 OMNI_GEMM_SPLITK_TILES=160 python tools/time_config1.py 256 2>/dev/null | tail -1
