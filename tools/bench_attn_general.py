#!/usr/bin/env python3
"""Dev tool (GPU): TF/s of `omni_flash_attn_general` (csrc/attention_general.hip) at the shapes of the in-tree callers it serves, next
to the tuned self-attention kernel on the Qwen-Image shape.   python tools/bench_attn_general.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.devlib  # noqa: E402,F401
from vllm_omni_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def case(name, B, Sq, Sk, H, dh, causal=False, mask=None):
    q = torch.randn(B * Sq, H * dh, device=dev, generator=g).to(BF)
    k = torch.randn(B * Sk, H * dh, device=dev, generator=g).to(BF)
    v = torch.randn(B * Sk, H * dh, device=dev, generator=g).to(BF)
    cq = (torch.arange(B + 1, dtype=torch.int32) * Sq).to(dev)
    ck = (torch.arange(B + 1, dtype=torch.int32) * Sk).to(dev)
    m, st = None, None
    if mask == "keypad":
        m = torch.ones(B, 1, 1, Sk, dtype=torch.bool, device=dev)
        m[..., Sk - 37:] = False
        st = tuple(m.expand(B, H, Sq, Sk).stride())
    t = timed(lambda: ops.flash_attn_general(q, k, v, cq, ck, H, H, Sq, Sk, 1 / math.sqrt(dh), causal=causal, mask=m, mask_strides=st))
    flop = 4.0 * B * H * Sq * Sk * dh * (0.5 if causal else 1.0)
    print(f"{name:58s} B {B} Sq {Sq:6d} Sk {Sk:6d} H {H} dh {dh:3d}: {t * 1e6:9.1f} us = {flop / t / 1e12:7.1f} TF/s = {flop / t / 2.5e15:.3f} of peak")
    if name.startswith("Qwen-Image"):
        t2 = timed(lambda: ops.flash_attn_varlen(q, k, v, cq, H, Sq, 1 / math.sqrt(dh)))
        print(f"{'   the tuned kernel (omni_flash_attn_fwd) on the same call':58s} {'':47s}{t2 * 1e6:9.1f} us = {flop / t2 / 1e12:7.1f} TF/s = {flop / t2 / 2.5e15:.3f} of peak")


case("Qwen-Image joint self-attention through the general path", 2, 4160, 4160, 24, 128)
case("wan2.2 cross-attention (video queries, 512 text keys)", 1, 32760, 512, 40, 128)
case("wan2.2 self-attention with a key-padding mask", 1, 8190, 8190, 40, 128, mask="keypad")
case("sd3 joint attention (head size 64)", 2, 4096 + 154, 4096 + 154, 24, 64)
case("causal, head size 128", 2, 4096, 4096, 24, 128, causal=True)
