#!/bin/bash
# Dev tool: builds libomni_cdna4 variants of attention.hip with extra -D flags into vllm_omni_amd/csrc/build/abl/, for
# same-box A/B runs of tools/bench_attn.py via OMNI_CDNA4_LIB (boxes differ by +-5 %, so only same-run comparisons count).
#   usage: build_attn_ablations.sh name1 "-DFOO=1 -DBAR=2" [name2 "flags2" ...]
# OMNI_ATTN_ABL=<mask> variants are timing-only ablations (results WRONG by construction).
set -e
cd "$(dirname "$0")/.."
B=vllm_omni_amd/csrc/build
mkdir -p $B/abl
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Ivllm_omni_amd/csrc $f \
      -c vllm_omni_amd/csrc/attention.hip -o $B/abl/attention_$n.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/gemm.o $B/abl/attention_$n.o $B/elementwise.o $B/vae.o \
      $B/dit_forward.o -o $B/abl/libomni_$n.so
  echo built $B/abl/libomni_$n.so
done
