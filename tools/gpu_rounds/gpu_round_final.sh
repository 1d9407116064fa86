#!/bin/bash
# final confirmation of a round: the three things the driver runs (GPU tests, smoke, default bench)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > gpurun_out/final_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/final_bench.log | cut -c1-400
