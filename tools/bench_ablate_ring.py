#!/usr/bin/env python3
"""Dev tool: ablate the default (ring) GEMM kernel: 0 full | 1 no DMA in loop | 3 no fragment reads | 4 no vmcnt+barrier,
on real data (8192^3, MLP-up shape) and on cache-resident operands (lda = 0, single W panel)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd import _native as N
lib = N.lib(); lib.omni_dev_gemm_ring_ablate.restype = C.c_int
lib.omni_dev_gemm_ring_ablate.argtypes = [C.POINTER(N.GemmParams), C.c_int, C.c_void_p]
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
names = {0: "full", 1: "no DMA in loop", 3: "no fragment reads", 4: "no vmcnt+barrier"}
for label, (M, Nn, K, lda0) in {"real 8192^3": (8192, 8192, 8192, False), "real mlp_up": (8320, 12288, 3072, False), "cache-resident": (262144, 256, 8192, True)}.items():
    a = torch.randn(256 if lda0 else M, K, device=dev).to(BF16); w = (torch.randn(Nn, K, device=dev) * 0.02).to(BF16)
    o = torch.empty(M, Nn, device=dev, dtype=BF16)
    p = N.GemmParams(); p.ngroups, p.N, p.K, p.epilogue = 1, Nn, K, 0
    g = p.g[0]; g.A, g.lda, g.M, g.W, g.out, g.ldo = a.data_ptr(), (0 if lda0 else K), M, w.data_ptr(), o.data_ptr(), Nn
    print(f"--- {label}: M={M} N={Nn} K={K}")
    for mode in (0, 1, 3, 4):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3): assert lib.omni_dev_gemm_ring_ablate(C.byref(p), mode, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): lib.omni_dev_gemm_ring_ablate(C.byref(p), mode, st)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        print(f"  mode {mode} {names[mode]:20s}: {t*1e3:8.3f} ms  {2.0*M*Nn*K/t/1e12:7.1f} TF/s-eq", flush=True)
