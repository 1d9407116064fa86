#!/usr/bin/env python3
"""Launch ONE hot kernel a few times (for rocprofv3 --kernel-trace / --pmc passes).
   python tools/run_kernel.py roofline|torchmm_mlp_up|gemm_mlp_up|gemm_mlp_down|gemm_qkv|gemm_out|attention|attention6|attention10|
                              conv<Cin>_<Cout>_<side>[_n] [iters]     (the VAE's bordered 3x3 conv; _n = with the fused norm output)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "gemm_mlp_up"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = torch.Generator(device=dev).manual_seed(0)
D, Mi, Mt = 3072, 8192, 128


def rn(*shape, s=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * s).to(BF16)


if which.startswith("gemm"):
    N, K, epi = {"gemm_mlp_up": (4 * D, D, ops.EPI_BIAS_GELU_TANH), "gemm_mlp_down": (D, 4 * D, ops.EPI_BIAS),
                 "gemm_qkv": (3 * D, D, ops.EPI_BIAS), "gemm_out": (D, D, ops.EPI_BIAS)}[which]
    xi, xt, wi, wt, b = rn(Mi, K), rn(Mt, K), rn(N, K, s=0.02), rn(N, K, s=0.02), rn(N)
    oi, ot = torch.empty(Mi, N, dtype=BF16, device=dev), torch.empty(Mt, N, dtype=BF16, device=dev)
    fn = lambda: ops.gemm([ops.GemmGroupArgs(xi, wi, b, oi), ops.GemmGroupArgs(xt, wt, b, ot)], epi)  # noqa: E731
elif which == "roofline":
    # exactly bench.py's roofline launch: MLP-up + GELU at the step-batch shape (R = bench.py's default 5), production layouts
    R = int(os.environ.get("BENCH_R", "5"))
    Mi, Mt, N, K = 2 * R * 4096, 2 * R * 64, 4 * D, D
    xi, xt = ops.w_to_k32_blocked(rn(Mi, K)), ops.w_to_k32_blocked(rn(Mt, K))
    wi, wt, b = ops.w_to_k32_blocked(rn(N, K, s=0.02)), ops.w_to_k32_blocked(rn(N, K, s=0.02)), rn(N)
    oi, ot = torch.empty(Mi, N, dtype=BF16, device=dev), torch.empty(Mt, N, dtype=BF16, device=dev)
    fn = lambda: ops.gemm([ops.GemmGroupArgs(xi, wi, b, oi, a_k32_blocked=True, out_k32_blocked=True),  # noqa: E731
                           ops.GemmGroupArgs(xt, wt, b, ot, a_k32_blocked=True, out_k32_blocked=True)],
                          ops.EPI_BIAS_GELU_TANH, w_k32_blocked=True)
elif which == "torchmm_mlp_up":
    # the vendor kernel (hipBLASLt through torch.mm: no bias, no GELU) on the roofline launch's shape, for counter passes
    R = int(os.environ.get("BENCH_R", "5"))
    M, N, K = 2 * R * 4160, 4 * D, D
    xa, wa = rn(M, K), rn(N, K, s=0.02)
    oa = torch.empty(M, N, dtype=BF16, device=dev)
    fn = lambda: torch.mm(xa, wa.t(), out=oa)  # noqa: E731
elif which in ("attention", "attention6", "attention10"):
    B, H, S = {"attention": 2, "attention6": 6, "attention10": 10}[which], 24, 4160      # attention10 = the bench's step-batch
    q, k, v = rn(B * S, H * 128), rn(B * S, H * 128), rn(B * S, H * 128)
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
    fn = lambda: ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128))  # noqa: E731
elif which.startswith("conv"):
    parts = which[4:].split("_")
    cin, cout, side = int(parts[0]), int(parts[1]), int(parts[2])
    x = torch.nn.functional.pad(rn(1, side, side, cin), (0, 0, 1, 1, 1, 1))
    w, b, gm = rn(cout, 3, 3, cin, s=0.05), rn(cout), rn(cout)
    res = torch.nn.functional.pad(rn(1, side, side, cout), (0, 0, 1, 1, 1, 1))
    kw = dict(norm_gamma=gm) if len(parts) > 3 else {}
    fn = lambda: ops.vae_conv2d(x, w, b, res=res, x_bordered=True, y_bordered=True, **kw)  # noqa: E731
else:
    raise SystemExit("unknown kernel")
for _ in range(iters):
    fn()
torch.cuda.synchronize()
print("done", which, iters)
