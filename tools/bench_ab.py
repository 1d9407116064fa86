#!/usr/bin/env python3
"""Same-process interleaved A/B of kernel variants on one MI355X (dev tool; cdna guide rule 24: N variants x M rounds).

  GEMM: ring (OMNI_GEMM_VARIANT 1) vs ping-pong (3) vs torch.mm (hipBLASLt, no epilogue) at the bench step-batch shapes
        (M = 24576 + 384) and the B=2 CFG-pair shapes (M = 8192 + 128), production K32-blocked layouts.
  attention: block order 0 (heads fastest) vs 1 (XCD-aware head-major) at B = 2 / 6, S = 4160, H = 24.
"""
import ctypes
import json
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import _native as N, ops  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device("cuda:0")
lib = N.lib()
raw = ctypes.CDLL(N.LIB_PATH)


def timeit(fn, iters=10, warmup=2):
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


VARIANTS = {"ring": 1, "pp4": 3}
if os.environ.get("AB_DEV_LIB"):          # a -DOMNI_DEV build (tools/build_variants.sh) also carries the two-phase variants
    VARIANTS.update({"pp2": 5, "pp2_dmafirst": 6})


def main():
    rounds = int(os.environ.get("AB_ROUNDS", "5"))
    res = {}
    g = torch.Generator(device=dev).manual_seed(0)

    def rn(*shape, s=1.0):
        return (torch.randn(*shape, device=dev, generator=g) * s).to(BF16)

    D = 3072
    for Mi, Mt in ((24576, 384), (8192, 128)):
        for name, Nn, K, epi in (("qkv", 3 * D, D, ops.EPI_BIAS), ("out_proj", D, D, ops.EPI_BIAS),
                                 ("mlp_up_gelu", 4 * D, D, ops.EPI_BIAS_GELU_TANH), ("mlp_down", D, 4 * D, ops.EPI_BIAS)):
            xi, xt = rn(Mi, K), rn(Mt, K)
            wi, wt, b = rn(Nn, K, s=0.02), rn(Nn, K, s=0.02), rn(Nn)
            oi, ot = torch.empty(Mi, Nn, dtype=BF16, device=dev), torch.empty(Mt, Nn, dtype=BF16, device=dev)
            xib, xtb, wib, wtb = (ops.w_to_k32_blocked(z) for z in (xi, xt, wi, wt))

            def ours():
                ops.gemm([ops.GemmGroupArgs(xib, wib, b, oi, a_k32_blocked=True),
                          ops.GemmGroupArgs(xtb, wtb, b, ot, a_k32_blocked=True)], epi, w_k32_blocked=True)

            def ref():
                torch.mm(xi, wi.t())
                torch.mm(xt, wt.t())

            # bit-identity of the kernel families
            raw.omni_dev_gemm_set_variant(1); ours(); o1 = oi.clone(); t1 = ot.clone()
            same = True
            for v in VARIANTS.values():
                raw.omni_dev_gemm_set_variant(v); ours()
                torch.cuda.synchronize()
                same = same and bool(torch.equal(o1, oi) and torch.equal(t1, ot))
            fl = 2.0 * (Mi + Mt) * Nn * K
            ts = {k: [] for k in list(VARIANTS) + ["mm"]}
            for _ in range(rounds):
                for k, v in VARIANTS.items():
                    raw.omni_dev_gemm_set_variant(v); ts[k].append(timeit(ours))
                ts["mm"].append(timeit(ref))
            line = {k: fl / statistics.median(v) / 1e12 for k, v in ts.items()}
            res[f"gemm_{name}_M{Mi}"] = dict(tflops=line, bit_identical=same)
            print(f"gemm {name:12s} M={Mi}+{Mt} N={Nn} K={K}: " + "  ".join(f"{k} {x:7.1f}" for k, x in line.items())
                  + f" TF/s  bit-identical={same}", flush=True)
            del xi, xt, wi, wt, oi, ot, xib, xtb, wib, wtb
    for n in (4096, 8192):
        a, w = rn(n, n), rn(n, n, s=0.02)
        o = torch.empty(n, n, dtype=BF16, device=dev)
        fn = lambda: ops.gemm([ops.GemmGroupArgs(a, w, None, o)], ops.EPI_BIAS)
        ts = {k: [] for k in list(VARIANTS) + ["mm"]}
        for _ in range(rounds):
            for k, v in VARIANTS.items():
                raw.omni_dev_gemm_set_variant(v); ts[k].append(timeit(fn))
            ts["mm"].append(timeit(lambda: torch.mm(a, w.t())))
        line = {k: 2 * n ** 3 / statistics.median(v) / 1e12 for k, v in ts.items()}
        res[f"gemm_sq{n}_rowmajor"] = line
        print(f"gemm square {n} (row-major operands): " + "  ".join(f"{k} {x:7.1f}" for k, x in line.items()) + " TF/s", flush=True)
        del a, w, o
    raw.omni_dev_gemm_set_variant(-1)

    H, S = 24, 4096 + 64
    for B in (() if os.environ.get('AB_SKIP_ATTN') else (2, 6)):
        q, k, v = rn(B * S, H * 128), rn(B * S, H * 128), rn(B * S, H * 128)
        cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(dev)
        fn = lambda: ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128))
        raw.omni_dev_attn_set_block_order(0); o0 = fn().clone()
        raw.omni_dev_attn_set_block_order(1); o1 = fn()
        torch.cuda.synchronize()
        same = bool(torch.equal(o0, o1))
        fl = 4.0 * B * H * S * S * 128
        ts = {0: [], 1: []}
        for _ in range(rounds):
            for od in (0, 1):
                raw.omni_dev_attn_set_block_order(od); ts[od].append(timeit(fn, iters=6))
        line = {f"order{od}": fl / statistics.median(v) / 1e12 for od, v in ts.items()}
        res[f"attention_B{B}"] = dict(tflops=line, bit_identical=same)
        print(f"attention B={B}: heads-fastest {line['order0']:7.1f}  XCD head-major {line['order1']:7.1f} TF/s  bit-identical={same}", flush=True)
    raw.omni_dev_attn_set_block_order(-1)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_ab.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
