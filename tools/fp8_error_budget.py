#!/usr/bin/env python3
"""Dev tool (GPU): where does the fp8 mode's error come from, at the BENCHMARKED width and depth?

60 full-width layers, ONE 1024^2 item (4096 image + 64 text rows), random N(0, 0.02^2) weights with non-zero biases and
jittered norm weights (the parity tests' model).  Checker: the fp32 oracle on the GPU.  Prints

  1. per-layer growth of the image residual stream's error (module-level walk, one native block per step) for the bf16 path,
     the all-fp8 path and the 'accurate' recipe;
  2. one forward: error vs the fp32 oracle and vs the bf16 product path for every single GEMM class in fp8 and for the
     candidate recipes;
  3. the 4-step true-CFG loop's final latent for bf16 / all-fp8 / each recipe;
  4. what each recipe costs: ms per forward at the bench's 10-item step-batch.

round-4 use: choose the recipe the bench reports.   python tools/fp8_error_budget.py [--layers 60]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import qwen_image_oracle as O  # noqa: E402  (dev tool: the oracle is the checker here, never the product)
from _util import cosine, rel_l2  # noqa: E402

BF16, DEV = torch.bfloat16, "cuda:0"


def model(layers, seed=1234):
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    m = QwenImageTransformer2DModel(num_layers=layers, device=DEV)
    m.init_random_(seed=seed)
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    for n, p in m.named_parameters():
        if p.dim() == 1 and "norm" in n:
            p.data.add_(0.1 * torch.randn(p.shape, device=DEV, generator=g).to(BF16))
        elif p.dim() == 1:
            p.data.copy_((0.02 * torch.randn(p.shape, device=DEV, generator=g)).to(BF16))
    return m


def walk(m, lat, pos, sig, grid):
    """Module-level forward (the surface cache hooks walk): returns (noise_pred, [hidden_img after each block])."""
    hidden = m.img_in(lat)
    enc = m.txt_in(m.txt_norm(pos))
    temb = m.time_text_embed(sig.to(hidden.dtype), hidden, None)
    rot = m.pos_embed([[grid]], [pos.shape[1]], device=hidden.device)
    hs = []
    for blk in m.transformer_blocks:
        enc, hidden = blk(hidden_states=hidden, encoder_hidden_states=enc, temb=temb, image_rotary_emb=rot)
        hs.append(hidden.clone())
    return m.proj_out(m.norm_out(hidden, temb)), hs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=60)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--loop-recipes", default="", help="';'-separated class lists (e.g. 'qkv;qkv,out;all'): only the loop part, for these")
    args = ap.parse_args()
    only_loop = [tuple(r.split(",")) for r in args.loop_recipes.split(";") if r]
    torch.backends.cuda.matmul.allow_tf32 = False
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    L = args.layers
    m = model(L)
    grid, S, T = (1, 64, 64), 4096, 64
    g = torch.Generator(device=DEV).manual_seed(42)
    lat = torch.randn(1, S, 64, device=DEV, generator=g).to(BF16)
    pos = torch.randn(1, T, 3584, device=DEV, generator=g).to(BF16)
    neg = torch.randn(1, 48, 3584, device=DEV, generator=g).to(BF16)
    sig = torch.tensor([0.6015625], device=DEV)
    P32 = {n: p.detach().float() for n, p in m.named_parameters()}
    with torch.no_grad():
        taps = {}
        ref = O.dit_forward(P32, lat.float(), pos.float(), sig, grid, num_heads=24, taps=taps)
        ref_h = [taps[f"block{i}"]["hidden"] for i in range(L)]
        del taps
    recipes = {"bf16": (), "fp8 all": m.FP8_CLASSES, "fp8 qkv": ("qkv",), "fp8 out": ("out",), "fp8 mlp_up": ("mlp_up",),
               "fp8 mlp_down": ("mlp_down",), "fp8 qkv+mlp_up": ("qkv", "mlp_up"), "fp8 qkv+mlp_up+mlp_down": ("qkv", "mlp_up", "mlp_down"),
               "fp8 qkv+out+mlp_up": ("qkv", "out", "mlp_up"), "fp8 mlp_up+mlp_down": ("mlp_up", "mlp_down")}
    if only_loop:
        recipes = {"bf16": ()}
        for r in only_loop:
            recipes["fp8 " + "+".join(r)] = m.FP8_CLASSES if r == ("all",) else r
    print(f"== {L} layers, one 1024^2 item (4096+{T} rows): one forward vs the fp32 oracle", flush=True)
    out16 = None
    for name, cls in ({} if only_loop else recipes).items():
        m.enable_fp8(cls) if cls else m.enable_fp8(False)
        out, hs = walk(m, lat, pos, sig, grid)
        torch.cuda.synchronize()
        if out16 is None:
            out16 = out.clone()
        growth = [rel_l2(hs[i], ref_h[i]) for i in (0, 1, 3, 7, 15, 29, 44, L - 1) if i < L]
        print(f"   {name:26s} pred vs fp32 {rel_l2(out, ref):.3e} (cos {cosine(out, ref):.5f})  vs bf16 path {rel_l2(out, out16):.3e}   "
              f"hidden_img err after layer 1/2/4/8/16/30/45/{L}: " + " ".join(f"{v:.2e}" for v in growth), flush=True)
        del hs
    del ref_h, ref
    # ---- the CFG loop
    print(f"== {args.steps}-step true-CFG loop, final latent vs the fp32 oracle loop", flush=True)
    ts, sg = O.flow_match_sigmas(args.steps, S)
    with torch.no_grad():
        x = lat.float()
        for i, t in enumerate(ts):
            s_in = (t.bfloat16() / 1000).bfloat16().float().expand(1).to(DEV)
            p = O.dit_forward(P32, x, pos.float(), s_in, grid, num_heads=24)
            n = O.dit_forward(P32, x, neg.float(), s_in, grid, num_heads=24)
            x = O.euler_step(x, O.cfg_combine(p, n, 4.0), float(sg[i]), float(sg[i + 1])).bfloat16().float()
        ref_final = x
    del P32
    torch.cuda.empty_cache()
    pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(use_hip_graph=False), device=DEV, transformer=m)
    req = OmniDiffusionRequest(height=1024, width=1024, num_inference_steps=args.steps, true_cfg_scale=4.0, latents=lat,
                               prompt_embeds=pos, negative_prompt_embeds=neg, output_type="latent")
    fin16 = None
    loop_names = list(recipes) if only_loop else ["bf16", "fp8 all", "fp8 qkv+mlp_up", "fp8 qkv+mlp_up+mlp_down", "fp8 qkv+out+mlp_up",
                                                  "fp8 mlp_up+mlp_down", "fp8 mlp_up"]
    for name in loop_names:
        cls = recipes[name]
        m.enable_fp8(cls) if cls else m.enable_fp8(False)
        fin = pipe.generate([req], output_type="latent")[0].output
        torch.cuda.synchronize()
        if fin16 is None:
            fin16 = fin.clone()
        print(f"   {name:26s} final latent vs fp32 {rel_l2(fin, ref_final):.3e} (cos {cosine(fin, ref_final):.5f})  vs bf16 path {rel_l2(fin, fin16):.3e}", flush=True)
    # ---- cost: one forward at the bench step-batch (10 items)
    print("== ms per forward at the bench step-batch (10 x (4096+64) rows)", flush=True)
    B = 10
    latb = torch.randn(B, S, 64, device=DEV, generator=g).to(BF16)
    txtb = torch.randn(B, T, 3584, device=DEV, generator=g).to(BF16)
    sigb = torch.full((B,), 0.6015625, device=DEV)
    kw = dict(hidden_states=latb, encoder_hidden_states=txtb, timestep=sigb, img_shapes=[[grid]] * B, txt_seq_lens=[T] * B, return_dict=False)
    for name in loop_names:
        cls = recipes[name]
        m.enable_fp8(cls) if cls else m.enable_fp8(False)
        m(**kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            m(**kw)
        torch.cuda.synchronize()
        print(f"   {name:26s} {(time.perf_counter() - t0) / 2 * 1e3:8.1f} ms", flush=True)


if __name__ == "__main__":
    main()
