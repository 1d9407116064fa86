#!/bin/bash
# round-2 GPU call A: new-kernel safety check first, then the whole gpu suite, the A/B microbench and a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 240 python -m pytest tests/test_gpu_ops.py -q -k pingpong > gpurun_out/r02a_pp.log 2>&1; PP=$?
echo "pingpong test rc=$PP"; tail -3 gpurun_out/r02a_pp.log
if [ $PP -ne 0 ]; then export OMNI_GEMM_VARIANT=1; echo "FALLING BACK TO RING KERNEL"; fi
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -s > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|rel_l2|layers|vae decode|diffuse vs|config0" gpurun_out/r02a_pytest.log | tail -40
timeout 900 python tools/bench_ab.py > gpurun_out/r02a_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r02a_ab.log | tail -30
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02a_bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/r02a_bench.log
