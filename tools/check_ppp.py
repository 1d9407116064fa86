#!/usr/bin/env python3
"""Dev tool (GPU; a -DOMNI_DEV library through OMNI_DEV_LIB + tools/devlib.py): the PERSISTENT ping-pong GEMM kernel (dev family 8,
gemm_bf16_ppp_kernel) against the one-shot ping-pong kernel (family 3) — bit for bit — on the shapes that stress its seam logic:
ragged M and N, two groups, no bias, K of two K-tiles, skipped row tiles, more and fewer tiles per workgroup.
    OMNI_DEV_LIB=vllm_omni_amd/csrc/build/abl/libomni_ppp.so python tools/check_ppp.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.devlib  # noqa: E402,F401
import torch  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
BF16 = torch.bfloat16
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(BF16)  # noqa: E731

CASES = [  # (M0, M1, N, K, epilogue, bias?, skip?)
    (40960, 640, 12288, 3072, ops.EPI_BIAS_GELU_TANH, True, False),      # the roofline launch
    (66000, 0, 1056, 128, ops.EPI_BIAS, True, False),                    # ragged M and N, two K-tiles
    (20000, 300, 3072, 256, ops.EPI_BIAS_GELU_TANH, False, False),       # no bias, ragged second group
    (40960, 640, 3072, 512, ops.EPI_BIAS_GELU_TANH, True, True),         # skipped row tiles (TeaCache predicate)
    (8192, 0, 8448, 192, ops.EPI_BIAS, True, False),                     # 32 x 33 tiles: just over one round
]
bad = 0
for M0, M1, N, K, epi, has_bias, skip in CASES:
    groups = {3: [], 8: []}
    outs = {3: [], 8: []}
    for M in (M0, M1):
        if not M:
            continue
        a = ops.w_to_k32_blocked(rn(M, K))
        w = ops.w_to_k32_blocked(rn(N, K, sc=0.05))
        b = rn(N, sc=0.2) if has_bias else None
        sk = None
        if skip:
            sk = (torch.arange((M + 255) // 256, device=dev) % 3 == 1).to(torch.int32)
        for f in (3, 8):
            o = torch.full((M, N), 7.0, dtype=BF16, device=dev)            # skipped tiles must stay untouched
            outs[f].append(o)
            groups[f].append(ops.GemmGroupArgs(a, w, b, o, a_k32_blocked=True, out_k32_blocked=True, tile_skip=sk))
    for f in (3, 8):
        ops.gemm(groups[f], epi, w_k32_blocked=True, kernel_hint=16 + f)
    torch.cuda.synchronize()
    eq = all(bool(torch.equal(x, y)) for x, y in zip(outs[3], outs[8]))
    fin = all(bool(torch.isfinite(x.float()).all()) for x in outs[8])
    ndiff = sum(int((x != y).sum()) for x, y in zip(outs[3], outs[8]))
    print(f"M={M0}+{M1} N={N} K={K} epi={epi} bias={has_bias} skip={skip}: bit-equal {eq} finite {fin} differing elements {ndiff}", flush=True)
    bad += 0 if eq else 1
print("PPP CHECK", "OK" if not bad else f"FAILED ({bad} cases)")
sys.exit(1 if bad else 0)
