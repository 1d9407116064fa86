#!/usr/bin/env python3
"""Dev tool (GPU, -DOMNI_DEV library through OMNI_DEV_LIB): one block GEMM shape, forced tail-split factor from the environment
(OMNI_GEMM_TAIL_NS, read once per process), us per launch.   python tools/bench_tail_split.py <rows_img> <rows_txt> <N> <K> [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.devlib  # noqa: E402,F401
from vllm_omni_amd import ops  # noqa: E402

Mi, Mt, N, K = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda r, c, s=1.0: (torch.randn(r, c, device=dev, generator=g) * s).to(BF)  # noqa: E731
blk = ops.w_to_k32_blocked
xi, xt, wi, wt = blk(rn(Mi, K)), blk(rn(Mt, K)), blk(rn(N, K, 0.02)), blk(rn(N, K, 0.02))
b = torch.zeros(N, device=dev, dtype=BF)
oi, ot = torch.empty(Mi, N, device=dev, dtype=BF), torch.empty(Mt, N, device=dev, dtype=BF)
ws = torch.empty(512 * 256 * 256, dtype=torch.float32, device=dev)
# a second, unrelated GEMM between the timed launches keeps the clocks / caches where a layer's kernel mix has them
y, wy, oy = blk(rn(Mi, 3072)), blk(rn(3072, 3072, 0.02)), torch.empty(Mi, 3072, device=dev, dtype=BF)
by = torch.zeros(3072, device=dev, dtype=BF)


def timed():
    ops.gemm([ops.GemmGroupArgs(xi, wi, b, oi, a_k32_blocked=True), ops.GemmGroupArgs(xt, wt, b, ot, a_k32_blocked=True)],
             ops.EPI_BIAS, w_k32_blocked=True, splitk_ws=ws)


def other():
    ops.gemm([ops.GemmGroupArgs(y, wy, by, oy, a_k32_blocked=True)], ops.EPI_BIAS, w_k32_blocked=True)


tiles = ((Mi + 255) // 256 + (Mt + 255) // 256) * (N // 256)
for ns in os.environ.get("SWEEP_NS", "1 2 3 4 6 8 12 16").split():
    os.environ["OMNI_GEMM_TAIL_NS"] = ns                      # the dev library reads it on every call
    for _ in range(3):
        other(); timed()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for e0, e1 in evs:
        other()
        e0.record()
        timed()
        e1.record()
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    print(f"M {Mi}+{Mt} N {N} K {K}: tiles {tiles} (= {tiles // 256} rounds + {tiles % 256}), tail ns {ns}: "
          f"median {t[len(t) // 2]:.1f} us, mean {sum(t) / len(t):.1f} us", flush=True)
