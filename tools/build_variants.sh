#!/bin/bash
# Dev tool: builds libomni_cdna4 variants with extra -D flags for ONE translation unit into vllm_omni_amd/csrc/build/abl/,
# for same-box A/B runs via OMNI_DEV_LIB + tools/devlib.py (boxes differ by +-5 %, so only same-run comparisons count).
#   usage: build_variants.sh <attention|gemm|elementwise|vae|dit_forward> name1 "-DFOO=1" [name2 "flags2" ...]
# OMNI_ATTN_ABL=<mask> variants of attention are timing-only ablations (results WRONG by construction).
set -e
cd "$(dirname "$0")/.."
B=vllm_omni_amd/csrc/build
tu=$1; shift
mkdir -p $B/abl
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fno-gpu-rdc -fvisibility=hidden -DOMNI_DEV -Iinclude -Ivllm_omni_amd/csrc $f \
      -c vllm_omni_amd/csrc/$tu.hip -o $B/abl/${tu}_$n.o 2>/dev/null
  objs=""
  for t in gemm attention attention_w64 attention_general elementwise vae dit_forward; do
    if [ $t = $tu ]; then objs="$objs $B/abl/${tu}_$n.o"; else objs="$objs $B/$t.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $B/abl/libomni_$n.so
  echo built $B/abl/libomni_$n.so
done
