#!/bin/bash
# Dev tool (GPU box): tools/bench_conv.py for the product library and the named dev variants (same box, same run).
SH=${SHAPES:-"384_384_128 384_384_256 192_192_512 96_96_1024 96_96_1024_n"}
echo -n "product: "; timeout 100 python tools/bench_conv.py $SH 2>/dev/null
for v in "$@"; do echo -n "$v: "; OMNI_DEV_LIB=vllm_omni_amd/csrc/build/abl/libomni_$v.so timeout 100 python tools/bench_conv.py $SH 2>/dev/null; done
