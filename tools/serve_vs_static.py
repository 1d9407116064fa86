#!/usr/bin/env python3
"""Dev tool (GPU): where does the serving path lose against the static loop?  Same process, same 5 requests:
  (a) pipeline.generate (static step-batch, what bench.py's headline times),
  (b) ContinuousStepBatcher with all 5 admitted at once (no ramps, one composition): per-step path cost only,
  (c) the same with staggered admissions (one request per step): re-composition cost,
each with and without VAE decode of the results.   python tools/serve_vs_static.py [--layers 60]"""
import argparse
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vllm_omni_amd.diffusion.data import OmniDiffusionConfig, TransformerConfig  # noqa: E402
from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline  # noqa: E402
from vllm_omni_amd.diffusion.request import OmniDiffusionRequest  # noqa: E402
from vllm_omni_amd.diffusion.step_batcher import ContinuousStepBatcher  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=60)
args = ap.parse_args()
dev = torch.device("cuda:0")
R = 5
cfg = OmniDiffusionConfig(model="x", max_step_batch=R, tf_model_config=TransformerConfig.from_dict({"num_layers": args.layers}))
pipe = QwenImagePipeline(od_config=cfg, device=dev)
pipe.transformer.init_random_(seed=1234)
pipe.vae.init_random_(seed=4321)
g = torch.Generator().manual_seed(5)


def req(out="latent"):
    return OmniDiffusionRequest(height=1024, width=1024, num_inference_steps=20, true_cfg_scale=4.0, output_type=out,
                                latents=torch.randn(1, 4096, 64, generator=g).to(torch.bfloat16),
                                prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(torch.bfloat16),
                                negative_prompt_embeds=torch.randn(1, 64, 3584, generator=g).to(torch.bfloat16))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


reqs = [req() for _ in range(R)]
t_static = timed(lambda: pipe.generate([copy.deepcopy(r) for r in reqs], output_type="latent"))
print(f"(a) static generate, {R} requests x 20 steps, latents only: {t_static:.3f} s", flush=True)


def serve(stagger: bool, out: str):
    b = ContinuousStepBatcher(pipe, max_items=R)
    rs = [copy.deepcopy(r) for r in reqs]
    for r in rs:
        r.output_type = out
    if not stagger:
        for i, r in enumerate(rs):
            b.add(r, i)
        return b.drain()
    done = []
    for i, r in enumerate(rs):
        b.add(r, i)
        b.wait_ready()
        done += b.step()
    return done + b.drain()


t_b = timed(lambda: serve(False, "latent"))
print(f"(b) batcher, all admitted at once, latents only: {t_b:.3f} s  ({t_b / t_static:.4f} of static)", flush=True)
st = ContinuousStepBatcher(pipe, max_items=R)
t_c = timed(lambda: serve(True, "latent"))
print(f"(c) batcher, one admission per step (24 steps, 100 sample-steps), latents only: {t_c:.3f} s", flush=True)
t_sd = timed(lambda: [pipe.decode_latents(o.output, 1024, 1024) for o in pipe.generate([copy.deepcopy(r) for r in reqs], output_type="latent")])
t_bd = timed(lambda: serve(False, "pt"))
print(f"(a') static + decode: {t_sd:.3f} s;  (b') batcher + decode inside step(): {t_bd:.3f} s  ({t_bd / t_sd:.4f})", flush=True)
