#!/usr/bin/env python3
"""Dev tool (GPU, needs a -DOMNI_DEV library: OMNI_DEV_LIB=... through tools/devlib.py): the GEMM kernel FAMILIES side by side
in ONE process at the bench's shapes — ping-pong (3, the product kernel), Q4 (4), V4 (7: the vendor's schedule), and
hipBLASLt through torch.mm (no epilogue) — interleaved rounds, median TF/s, and a bit-for-bit comparison of every family's
result with the ping-pong kernel's.   python tools/bench_gemm_families.py [--families 3,7] [--rounds 5]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.devlib  # noqa: E402,F401  (points the binding at OMNI_DEV_LIB)
import torch  # noqa: E402

from tools.bench_kernels import timeit  # noqa: E402
from vllm_omni_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--families", default="3,4,7")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--shapes", default="mlp_up,qkv_plain,out_proj,mlp_down,sq8192")
args = ap.parse_args()
fams = [int(f) for f in args.families.split(",")]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
BF16 = torch.bfloat16
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(BF16)  # noqa: E731
D, Mi, Mt = 3072, 40960, 640                      # the bench step-batch: 10 x (4096 + 64) rows

SHAPES = {  # name: (N, K, epilogue, blocked operands?, two groups?)
    "mlp_up": (4 * D, D, ops.EPI_BIAS_GELU_TANH, True, True),
    "mlp_up_k32out_bias": (4 * D, D, ops.EPI_BIAS, True, True),        # the same launch without the GELU arithmetic (K32-blocked out)
    "qkv_plain": (3 * D, D, ops.EPI_BIAS, True, True),
    "out_proj": (D, D, ops.EPI_BIAS, True, True),
    "mlp_down": (D, 4 * D, ops.EPI_BIAS, True, True),
    "sq8192": (8192, 8192, ops.EPI_BIAS, False, False),
    # the real epilogues of the two N = 3072 GEMMs of a block: res + gate[item] * (acc + bias), in place on the residual stream
    "out_gate": (D, D, ops.EPI_BIAS_GATE_RES, True, True),
    "down_gate": (D, 4 * D, ops.EPI_BIAS_GATE_RES, True, True),
}
for name in args.shapes.split(","):
    N, K, epi, blk, two = SHAPES[name]
    k32out = blk and (epi == ops.EPI_BIAS_GELU_TANH or "k32out" in name)
    m_i, m_t = (Mi, Mt) if two else (8192, 0)
    xi_rm = rn(m_i, K)
    wi_rm = rn(N, K, sc=0.02)
    b = rn(N, sc=0.1)
    xi, wi = (ops.w_to_k32_blocked(xi_rm), ops.w_to_k32_blocked(wi_rm)) if blk else (xi_rm, wi_rm)
    groups_of = []
    outs = {}
    for f in fams:
        oi = torch.zeros(m_i, N, dtype=BF16, device=dev)
        extra_i, extra_t = {}, {}
        if epi == ops.EPI_BIAS_GATE_RES:                 # 10 items: rows_per_item 4096 / 64; the residual is a separate buffer so
            gate = rn(10, N, sc=0.3)                     # that repeated launches compute the same thing
            extra_i = dict(res=rn(m_i, N), gate=gate, gate_item_stride=N, rows_per_item=4096)
            extra_t = dict(res=rn(m_t, N), gate=gate, gate_item_stride=N, rows_per_item=64)
        grp = [ops.GemmGroupArgs(xi, wi, b, oi, a_k32_blocked=blk, out_k32_blocked=k32out, **extra_i)]
        if two:
            xt, wt = ops.w_to_k32_blocked(rn(m_t, K)), ops.w_to_k32_blocked(rn(N, K, sc=0.02))
            grp.append(ops.GemmGroupArgs(xt, wt, b, torch.zeros(m_t, N, dtype=BF16, device=dev), a_k32_blocked=blk,
                                         out_k32_blocked=k32out, **extra_t))
        outs[f] = (oi, grp)
    flops = 2.0 * (m_i + m_t) * N * K
    run = {f: (lambda f=f: ops.gemm(outs[f][1], epi, w_k32_blocked=blk, kernel_hint=16 + f)) for f in fams}
    run["hipblaslt"] = lambda: torch.mm(xi_rm, wi_rm.t())
    for f in fams:                                   # correctness first: every family vs the ping-pong kernel, bit for bit
        run[f]()
    torch.cuda.synchronize()
    base = outs[fams[0]][0]
    eq = {f: bool(torch.equal(outs[f][0], base)) for f in fams}
    ref = (xi_rm[:512].float() @ wi_rm.float().t() + b.float())
    if epi == ops.EPI_BIAS_GELU_TANH:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    got = base[:512].float() if epi == ops.EPI_BIAS else None
    err = float((got - ref).norm() / ref.norm()) if got is not None else float("nan")
    res = {k: [] for k in run}
    for _ in range(args.rounds):
        for k, fn in run.items():
            res[k].append(flops / timeit(fn, iters=8, warmup=2) / 1e12)
    line = "  ".join(f"{k}: {statistics.median(v):7.1f}" for k, v in res.items())
    print(f"{name:10s} N={N} K={K}  TF/s  {line}   bit-equal to family {fams[0]}: {eq}  rel_l2 vs fp32 {err:.2e}", flush=True)
