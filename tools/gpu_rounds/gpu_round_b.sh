#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -q -k pingpong > gpurun_out/r02b_pp.log 2>&1; PP=$?
echo "pingpong test rc=$PP"; tail -3 gpurun_out/r02b_pp.log
timeout 600 python -m pytest tests/test_gpu_bench_shape_parity.py -q -s -k "60_layers" > gpurun_out/r02b_60.log 2>&1; echo "60-layer rc=$?"; grep -E "60 layers|product vs|passed|failed" gpurun_out/r02b_60.log
AB_SKIP_ATTN=1 timeout 900 python tools/bench_ab.py > gpurun_out/r02b_ab.log 2>&1; echo "ab rc=$?"; tail -12 gpurun_out/r02b_ab.log
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02b_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r02b_bench.log
