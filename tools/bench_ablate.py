#!/usr/bin/env python3
"""Dev tool: time the 2-stage GEMM kernel with parts removed (template ablation, cdna guide §5 'ablate first').
modes: 0 full | 1 no DMA in loop | 2 no MFMA | 3 no fragment reads | 4 no vmcnt+barrier | 5 MFMA only"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import _native as N  # noqa: E402

lib = N.lib()
lib.omni_dev_gemm_ablate.restype = C.c_int
lib.omni_dev_gemm_ablate.argtypes = [C.POINTER(N.GemmParams), C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
names = {0: "full", 1: "no DMA in loop", 2: "no MFMA", 3: "no LDS fragment reads", 4: "no vmcnt+barrier", 5: "MFMA only"}
for (M, Nn, K) in ((8320, 12288, 3072), (8192, 8192, 8192), (8320, 3072, 12288)):
    a = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(Nn, K, device=dev) * 0.02).to(BF16)
    o = torch.empty(M, Nn, device=dev, dtype=BF16)
    p = N.GemmParams()
    p.ngroups, p.N, p.K, p.epilogue = 1, Nn, K, 0
    g = p.g[0]
    g.A, g.lda, g.M, g.W, g.out, g.ldo = a.data_ptr(), K, M, w.data_ptr(), o.data_ptr(), Nn
    fl = 2.0 * M * Nn * K
    print(f"--- M={M} N={Nn} K={K}  ({fl/1e9:.0f} GFLOP)")
    for mode in range(6):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            assert lib.omni_dev_gemm_ablate(C.byref(p), mode, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.omni_dev_gemm_ablate(C.byref(p), mode, st)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"  mode {mode} {names[mode]:24s}: {t*1e3:8.3f} ms   ({fl/t/1e12:7.1f} TF/s-equivalent)")

# ---- E2/E3: everything cache-resident: A rows all alias row 0 (lda = 0), one W panel (N = 256) shared by all tiles
print("--- cache-resident operands: M=262144 (1024 tiles) N=256 K=8192, lda=0")
M, Nn, K = 262144, 256, 8192
a = torch.randn(256, K, device=dev).to(BF16)
w = (torch.randn(Nn, K, device=dev) * 0.02).to(BF16)
o = torch.empty(M, Nn, device=dev, dtype=BF16)
p = N.GemmParams()
p.ngroups, p.N, p.K, p.epilogue = 1, Nn, K, 0
g = p.g[0]
g.A, g.lda, g.M, g.W, g.out, g.ldo = a.data_ptr(), 0, M, w.data_ptr(), o.data_ptr(), Nn
fl = 2.0 * M * Nn * K
for mode in (0, 2, 5):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.omni_dev_gemm_ablate(C.byref(p), mode, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.omni_dev_gemm_ablate(C.byref(p), mode, st)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"  mode {mode} {names[mode]:24s}: {t*1e3:8.3f} ms   ({fl/t/1e12:7.1f} TF/s-equivalent)")
# same shape through the product entry point with the ring variant for comparison
os.environ["OMNI_GEMM_VARIANT"] = "1"

# ---- product (ring) kernel: real data vs cache-resident operands
lib.omni_gemm_bf16.restype = C.c_int
for label, (M, Nn, K, lda0) in {"ring real 8192^3": (8192, 8192, 8192, False), "ring cache-resident": (262144, 256, 8192, True)}.items():
    a = torch.randn(256 if lda0 else M, K, device=dev).to(BF16)
    w = (torch.randn(Nn, K, device=dev) * 0.02).to(BF16)
    o = torch.empty(M, Nn, device=dev, dtype=BF16)
    p = N.GemmParams()
    p.ngroups, p.N, p.K, p.epilogue = 1, Nn, K, 0
    g = p.g[0]
    g.A, g.lda, g.M, g.W, g.out, g.ldo = a.data_ptr(), (0 if lda0 else K), M, w.data_ptr(), o.data_ptr(), Nn
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert lib.omni_gemm_bf16(C.byref(p), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.omni_gemm_bf16(C.byref(p), st)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"--- {label}: {t*1e3:8.3f} ms  {2.0*M*Nn*K/t/1e12:7.1f} TF/s")
