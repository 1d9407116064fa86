#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
LOG=gpurun_out/r02aa_clocks.log
echo "# rocm-smi sampled once per second while tools/run_kernel.py loops the bench roofline GEMM / the B=6 attention" > $LOG
rocm-smi --showmaxpower 2>&1 | grep -i "max" >> $LOG
for K in roofline attention6; do
  echo "## $K" >> $LOG
  python tools/run_kernel.py $K 8000 > /dev/null 2>&1 &
  PID=$!
  for i in $(seq 1 40); do
    if ! kill -0 $PID 2>/dev/null; then break; fi
    echo "t=${i}s $(rocm-smi --showpower --showclocks 2>&1 | grep -i 'sclk\|Power (W)' | sed 's/GPU\[0\]\s*: //' | tr '\n' ';')" >> $LOG
    sleep 1
  done
  wait $PID
done
grep -v "97Mhz\|96Mhz\|95Mhz" $LOG | cut -c1-200 | head -60
