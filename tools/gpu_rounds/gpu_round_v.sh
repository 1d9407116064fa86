#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for R in 3 5 7 3 5; do
  timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --requests $R 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('R=$R', d['value'], d['dit_mfma_roofline_frac'])"
done 2>&1 | tee gpurun_out/r02v_R.log
