#!/usr/bin/env python3
"""Per-kernel microbenchmark on one MI355X (dev tool, not the judged bench.py).

Times each hot kernel at the BASELINE workload shapes (B=2 CFG pair, 1024^2: 8192 image + 128 text rows) with
HIP events on the launch stream, reports TFLOP/s or GB/s, and — as a known-good on-hardware reference
(cdna guide rule 10) — the same GEMM/attention through torch (hipBLASLt / SDPA).
"""
import argparse
import json
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        fn()
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    res = {}
    D, Mi, Mt = 3072, 8192, 128
    g = torch.Generator(device=dev).manual_seed(0)

    def rn(*shape, s=1.0):
        return (torch.randn(*shape, device=dev, generator=g) * s).to(BF16)

    # ---------------- GEMMs (grouped img+txt) ----------------
    for name, N, K, epi in (("qkv", 3 * D, D, ops.EPI_BIAS), ("out_proj", D, D, ops.EPI_BIAS),
                            ("mlp_up_gelu", 4 * D, D, ops.EPI_BIAS_GELU_TANH), ("mlp_down", D, 4 * D, ops.EPI_BIAS)):
        xi, xt = rn(Mi, K), rn(Mt, K)
        wi, wt, b = rn(N, K, s=0.02), rn(N, K, s=0.02), rn(N)
        oi, ot = torch.empty(Mi, N, dtype=BF16, device=dev), torch.empty(Mt, N, dtype=BF16, device=dev)
        # production layouts: K32-blocked A and W (row-major output), same values
        xib, xtb, wib, wtb = (ops.w_to_k32_blocked(z) for z in (xi, xt, wi, wt))
        fn = lambda: ops.gemm([ops.GemmGroupArgs(xib, wib, b, oi, a_k32_blocked=True),
                               ops.GemmGroupArgs(xtb, wtb, b, ot, a_k32_blocked=True)], epi, w_k32_blocked=True)
        t = timeit(fn)
        fl = 2.0 * (Mi + Mt) * N * K
        t_ref = timeit(lambda: (torch.mm(xi, wi.t()), torch.mm(xt, wt.t())))
        res[f"gemm_{name}"] = dict(ms=t * 1e3, tflops=fl / t / 1e12, torch_mm_ms=t_ref * 1e3, torch_mm_tflops=fl / t_ref / 1e12)
        print(f"gemm {name:12s} M={Mi}+{Mt} N={N} K={K}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s | torch.mm {t_ref*1e3:8.3f} ms {fl/t_ref/1e12:7.1f} TF/s", flush=True)
        del xi, xt, wi, wt, oi, ot
    # single big square for comparison with the guide's numbers
    for n in (4096, 8192):
        a, w = rn(n, n), rn(n, n, s=0.02)
        o = torch.empty(n, n, dtype=BF16, device=dev)
        t = timeit(lambda: ops.gemm([ops.GemmGroupArgs(a, w, None, o)], ops.EPI_BIAS), iters=10)
        t_ref = timeit(lambda: torch.mm(a, w.t()), iters=10)
        print(f"gemm square {n}: {2*n**3/t/1e12:7.1f} TF/s | torch.mm {2*n**3/t_ref/1e12:7.1f} TF/s", flush=True)
        res[f"gemm_sq{n}"] = dict(tflops=2 * n ** 3 / t / 1e12, torch_mm_tflops=2 * n ** 3 / t_ref / 1e12)
        del a, w, o

    # ---------------- attention ----------------
    H, S = 24, 4096 + 64
    for B in (2, 6):
        q, k, v = rn(B * S, H * 128), rn(B * S, H * 128), rn(B * S, H * 128)
        cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
        t = timeit(lambda: ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128)), iters=10)
        fl = 4.0 * B * H * S * S * 128
        q4, k4, v4 = (x.view(B, S, H, 128).permute(0, 2, 1, 3) for x in (q, k, v))
        t_ref = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4), iters=5)
        res[f"attention_B{B}"] = dict(ms=t * 1e3, tflops=fl / t / 1e12, sdpa_ms=t_ref * 1e3, sdpa_tflops=fl / t_ref / 1e12)
        print(f"attention B={B} H={H} S={S}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TF/s | torch SDPA {t_ref*1e3:8.3f} ms {fl/t_ref/1e12:7.1f} TF/s", flush=True)
    B = 2

    # ---------------- HBM-bound ----------------
    x, mod = rn(Mi, D), rn(2, 6 * D)
    item = (torch.arange(Mi, device=dev) // 4096).int()
    y = torch.empty_like(x)
    t = timeit(lambda: ops.adaln_modulate(x, mod[:, D:], mod, mod_item_stride=6 * D, row_item_map=item, out=y))
    print(f"adaln {Mi}x{D}: {t*1e6:8.1f} us  {2*Mi*D*2/t/1e9:7.0f} GB/s", flush=True)
    res["adaln"] = dict(us=t * 1e6, gbps=2 * Mi * D * 2 / t / 1e9)
    qk = rn(B * S, D)
    pos = torch.arange(B * S, device=dev, dtype=torch.int32) % S
    cos, sin, wn = rn(S, 64), rn(S, 64), rn(128)
    t = timeit(lambda: ops.qk_norm_rope_(qk, H, wn, wn, cos, sin, pos, 64))
    print(f"qk_norm_rope {B*S}x{D}: {t*1e6:8.1f} us  {2*B*S*D*2/t/1e9:7.0f} GB/s", flush=True)
    res["qk_norm_rope"] = dict(us=t * 1e6, gbps=2 * B * S * D * 2 / t / 1e9)
    temb, wm, bm = rn(1, D), rn(6 * D, D, s=0.02), rn(6 * D)
    t = timeit(lambda: ops.linear_smallbatch(temb, wm, bm, act_in=1))
    print(f"modulation gemv 6D x D (B=1): {t*1e6:8.1f} us  {6*D*D*2/t/1e9:7.0f} GB/s", flush=True)
    res["mod_gemv"] = dict(us=t * 1e6, gbps=6 * D * D * 2 / t / 1e9)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_kernels.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
