#!/bin/bash
# Closing pass of a round 1/2 (first used in round 5): the full GPU suite + smoke on the library that ships (flags: vllm_omni_amd/csrc/build.py FLAGS).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r05}
( echo "library: vllm_omni_amd/libomni_cdna4.so  sha256 $(sha256sum vllm_omni_amd/libomni_cdna4.so | cut -c1-16)  built with: hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -fvisibility=hidden (csrc/build.py; no -D switches)"
  echo "git: $(git rev-parse HEAD 2>/dev/null || echo n/a)"
  timeout 1700 python -m pytest tests -m gpu -q -rs --durations=15 2>&1 | grep -v amdgpu.ids
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > $OUT/${TAG}_pytest_gpu_full.log 2>&1
tail -30 $OUT/${TAG}_pytest_gpu_full.log
