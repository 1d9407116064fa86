#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 120 tools/probe/overlap 2>&1 | tee gpurun_out/r02q_overlap.log
