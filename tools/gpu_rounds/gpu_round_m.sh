#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r02m_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r02m_pytest.log
timeout 1500 python bench.py > gpurun_out/r02m_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r02m_bench.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
