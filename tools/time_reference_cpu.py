#!/usr/bin/env python3
"""AUTHORING CONTAINER ONLY (needs /root/reference): the reference's own DiT blocks (unmodified file, shim-imported through
oracle/ref_shims.py) and the oracle port timed side by side on THIS host's cores, same weights, same input, same 4 full-width
blocks at the headline shape (4096 image + 64 text tokens, fp32).  bench.py's `cpu_baseline` is the oracle port (kind "port":
/root/reference does not exist on the GPU box); this ties that port to the reference by one measured ratio.

    python tools/time_reference_cpu.py [--layers 4] [--threads 8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import qwen_image_oracle as O  # noqa: E402
import ref_shims  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
args = ap.parse_args()
torch.set_num_threads(args.threads)
L, S, T, D = args.layers, 4096, 64, 3072
P = O.make_dit_params(L, seed=1234)
model, cfg = ref_shims.build_reference_model(L, dtype=torch.float32)
model.load_state_dict(P, strict=True)
g = torch.Generator().manual_seed(0)
lat, txt = torch.randn(1, S, 64, generator=g), torch.randn(1, T, 3584, generator=g)
sig = torch.tensor([0.5])
kw = dict(hidden_states=lat, encoder_hidden_states=txt, encoder_hidden_states_mask=torch.ones(1, T, dtype=torch.long),
          timestep=sig, img_shapes=[[(1, 64, 64)]], txt_seq_lens=[T])


def best(fn, n=2):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), out


with torch.no_grad():
    t_ref, o_ref = best(lambda: ref_shims.reference_forward(model, cfg, **kw))
    t_orc, o_orc = best(lambda: O.dit_forward(P, lat, txt, sig, (1, 64, 64), num_heads=24))
    O.FUSED_SDPA = True                      # what bench.py's cpu_baseline times: the reference's own attention op
    t_fus, o_fus = best(lambda: O.dit_forward(P, lat, txt, sig, (1, 64, 64), num_heads=24))
    O.FUSED_SDPA = False
err = float((o_ref - o_orc).norm() / o_ref.norm())
err_f = float((o_ref - o_fus).norm() / o_ref.norm())
print(f"{L} full-width blocks (+ in/out projections), 4096+64 tokens, fp32, {args.threads} threads on {os.cpu_count()} CPUs")
print(f"reference (shim-imported vllm_omni DiT): {t_ref:.2f} s = {t_ref / L:.2f} s/block")
print(f"oracle port (oracle/qwen_image_oracle.py): {t_orc:.2f} s = {t_orc / L:.2f} s/block")
print(f"oracle port with FUSED_SDPA (= bench.py cpu_baseline): {t_fus:.2f} s = {t_fus / L:.2f} s/block")
print(f"ratio port / reference = {t_orc / t_ref:.3f} (checker form), {t_fus / t_ref:.3f} (timed form);  outputs agree to rel_l2 {err:.2e} / {err_f:.2e}")
