#!/usr/bin/env python3
"""Dev tool (GPU): the VAE decoder's bordered 3x3 convs one by one — 20 back-to-back launches per shape between two events.
   [OMNI_DEV_LIB=...abl/libomni_<variant>.so] python tools/bench_conv.py [Cin_Cout_side[_n] ...]       (_n: with the fused norm output)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("OMNI_DEV_LIB"):
    import tools.devlib  # noqa: F401
import torch  # noqa: E402

from vllm_omni_amd import ops  # noqa: E402

BF16, dev = torch.bfloat16, torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
shapes = sys.argv[1:] or ["384_384_128", "384_384_256", "384_192_512", "192_192_512", "192_96_1024", "96_96_1024", "96_96_1024_n", "192_192_512_n"]
line = []
for sh in shapes:
    parts = sh.split("_")
    cin, cout, side = int(parts[0]), int(parts[1]), int(parts[2])
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(BF16)  # noqa: E731
    x = torch.nn.functional.pad(rn(1, side, side, cin), (0, 0, 1, 1, 1, 1))
    w, b, gm = rn(cout, 3, 3, cin, sc=0.05), rn(cout), rn(cout)
    res = torch.nn.functional.pad(rn(1, side, side, cout), (0, 0, 1, 1, 1, 1))
    kw = dict(norm_gamma=gm) if len(parts) > 3 else {}
    fn = lambda: ops.vae_conv2d(x, w, b, res=res, x_bordered=True, y_bordered=True, **kw)  # noqa: E731
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    line.append(f"{sh} {us:7.1f} us {2.0 * side * side * cin * cout * 9 / us / 1e6:6.0f} TF/s")
print(" | ".join(line))
