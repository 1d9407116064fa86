#!/usr/bin/env python3
"""Dev probe: does the GEMM rate depend on the row stride (L2 channel camping)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for (M, N, K) in ((8192, 8192, 8192), (8192, 8192, 8256), (8192, 8192, 8128), (8320, 12288, 3072), (8320, 12288, 3136), (8320, 3072, 12288), (8320, 3072, 12352), (24960, 12288, 3072), (24960, 12288, 3136)):
    a = torch.randn(M, K, device=dev).to(BF16); w = (torch.randn(N, K, device=dev) * 0.02).to(BF16)
    o = torch.empty(M, N, device=dev, dtype=BF16)
    t = timeit(lambda: ops.gemm([ops.GemmGroupArgs(a, w, None, o)]))
    t2 = timeit(lambda: torch.mm(a, w.t()))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} (row stride {K*2} B): ours {fl/t/1e12:7.1f} TF/s | torch.mm {fl/t2/1e12:7.1f} TF/s", flush=True)
    del a, w, o
