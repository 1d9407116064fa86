#!/bin/bash
# Dev tool: builds tools/probe/pp_probe (torch-free GEMM timer + phase-probe reader) and the two libraries it is usually given:
# the probe build of the library (-DOMNI_DEV -DOMNI_PP_PROBE=1 on gemm.hip) and a plain dev build of the same sources as the
# reference for the probe's own cost.  Run from anywhere; needs the product objects (python vllm_omni_amd/csrc/build.py).
#   then on a GPU box:  tools/probe/pp_probe vllm_omni_amd/libomni_cdna4.so vllm_omni_amd/csrc/build/abl/libomni_probe.so
set -e
cd "$(dirname "$0")/../.."
[ -f vllm_omni_amd/csrc/build/gemm.o ] || python vllm_omni_amd/csrc/build.py
tools/build_variants.sh gemm probe "-DOMNI_PP_PROBE=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probe/pp_probe.cpp -o tools/probe/pp_probe -ldl
echo built tools/probe/pp_probe
