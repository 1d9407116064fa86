#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r02y_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r02y_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
