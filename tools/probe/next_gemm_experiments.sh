#!/bin/bash
# The GEMM experiments queued at the end of round 4 (DESIGN.md 7 item 29 / "Open leads"): builds the variant libraries and the
# torch-free harnesses here (no GPU needed), then prints the ONE gpurun command that measures them (30-60 s of box time).
#   sched0  = the round-2..4 K-loop with s_setprio (the reference point of every earlier log)
#   product = restructured loop, no s_setprio (what ships)
#   big     = two big phases per K-tile, 32 MFMAs per cluster (-DOMNI_PP_SCHED=9): NOT yet run on hardware
#   p_*     = the same with the phase probe (-DOMNI_PP_PROBE=1)
set -e
cd "$(dirname "$0")/../.."
python vllm_omni_amd/csrc/build.py > /dev/null
rm -f vllm_omni_amd/csrc/build/abl/*
tools/build_variants.sh gemm sched0 "-DOMNI_PP_SCHED=0 -DOMNI_PP_SETPRIO=1" big "-DOMNI_PP_SCHED=9" \
    p_base "-DOMNI_PP_PROBE=1" p_big "-DOMNI_PP_PROBE=1 -DOMNI_PP_SCHED=9"
rm -f vllm_omni_amd/csrc/build/abl/*.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probe/pp_probe.cpp -o tools/probe/pp_probe -ldl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/probe/attn_bench.cpp -o tools/probe/attn_bench -ldl
L=vllm_omni_amd/csrc/build/abl
P=vllm_omni_amd/libomni_cdna4.so
cat <<CMD

run (one call; the sweep first: a variant that is not bit-identical is not worth timing):

gpurun --timeout 150 -- 'mkdir -p gpurun_out; (
  timeout 60 ./tools/probe/pp_probe --sweep $P $L/libomni_sched0.so $L/libomni_big.so;
  timeout 30 ./tools/probe/pp_probe --iters 20 $L/libomni_sched0.so $P $L/libomni_big.so $L/libomni_p_base.so $L/libomni_p_big.so;
  for s in "--n 3072 --k 3072 --epi 2" "--n 9216 --k 3072 --epi 4" "--n 3072 --k 12288 --epi 2"; do echo "== \$s"; timeout 20 ./tools/probe/pp_probe --iters 20 \$s $L/libomni_sched0.so $P $L/libomni_big.so; done;
  timeout 20 ./tools/probe/attn_bench --iters 20 $P
) > gpurun_out/r05_gemm_first.log 2>&1; tail -40 gpurun_out/r05_gemm_first.log'
CMD
