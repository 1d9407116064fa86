#!/usr/bin/env python3
"""Dev experiment (GPU): do the kernel TAILS of a denoise step fill up when the step-batch runs as TWO ragged forwards on two HIP
streams (the hardware dispatches workgroups of both grids; when one kernel's grid runs dry the other's fills the idle CUs)?
   python tools/exp_dual_stream.py [layers] [R]
Prints ms per step: one forward over all 2R items vs two concurrent forwards over a (R - R//2, R//2) split, each with its own workspace."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_omni_amd.diffusion.batch import build_ragged_batch  # noqa: E402
from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 60
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
BF = torch.bfloat16
m = QwenImageTransformer2DModel(num_layers=layers, device=dev).init_random_(seed=1234)
g = torch.Generator(device=dev).manual_seed(0)
grid, S, T = (1, 64, 64), 4096, 64


def inputs(n_items):
    return (torch.randn(n_items * S, 64, device=dev, generator=g).to(BF), torch.randn(n_items * T, 3584, device=dev, generator=g).to(BF),
            torch.full((1,), 0.6015625, device=dev), m.prepare_batch(build_ragged_batch([T] * n_items, grid, temb_rows=[0] * n_items)))


def fwd(inp, out):
    lat, txt, sig, prep = inp
    m.forward_ragged(prep, lat, txt, sig, out=out)


def timed(fn, n=4):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = inputs(2 * R)
out_full = torch.empty(2 * R * S, 64, dtype=BF, device=dev)
fwd(full, out_full)
ws_full = m._workspace
t_one = timed(lambda: fwd(full, out_full))
na, nb = 2 * (R - R // 2), 2 * (R // 2)
A, B = inputs(na), inputs(nb)
oa, ob = torch.empty(na * S, 64, dtype=BF, device=dev), torch.empty(nb * S, 64, dtype=BF, device=dev)
m._workspace = None
fwd(A, oa)
ws_a = m._workspace
m._workspace = None
fwd(B, ob)
ws_b = m._workspace
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def dual():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    m._workspace = ws_a
    with torch.cuda.stream(s1):
        fwd(A, oa)
    m._workspace = ws_b
    with torch.cuda.stream(s2):
        fwd(B, ob)
    cur.wait_stream(s1); cur.wait_stream(s2)


def serial():
    m._workspace = ws_a
    fwd(A, oa)
    m._workspace = ws_b
    fwd(B, ob)


t_dual = timed(dual)
t_serial = timed(serial)
m._workspace = ws_full
t_one2 = timed(lambda: fwd(full, out_full))
print(f"{layers} layers, {2 * R} items: one forward {t_one:.1f} / {t_one2:.1f} ms; split {na}+{nb} serial {t_serial:.1f} ms; "
      f"split {na}+{nb} on two streams {t_dual:.1f} ms ({100 * (t_one2 / t_dual - 1):+.2f} % vs one forward)")
