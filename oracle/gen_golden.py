#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the UNMODIFIED reference DiT.

Run in the authoring container only (needs /root/reference):

    python oracle/gen_golden.py

For each case it builds the reference `QwenImageTransformer2DModel` through oracle/ref_shims.py,
loads seeded synthetic weights (oracle.make_dit_params — the same generator the tests use, so the
weights are NOT stored, only a checksum), runs the reference forward on seeded inputs and stores
inputs + outputs (+ per-block outputs captured with forward hooks).  The fixtures pin
oracle/qwen_image_oracle.py (tests/test_oracle_golden.py) and are the targets of the GPU parity
tests (tests/test_gpu_dit_forward.py, tests/test_gpu_pipeline.py, tests/test_gpu_bench_shape_parity.py).
Nothing here travels to the GPU box except the .npz files.

Round 2 adds reference-run fixtures for the rest of the path (oracle/ref_shims_pipeline.py):
  vae_decode_*          the reference's vendored AutoencoderKLQwenImage.decode (autoencoder_kl_qwenimage.py:839-887)
  pipe_helpers          calculate_shift / _pack_latents / _unpack_latents / prepare_timesteps
                        (pipeline_qwen_image.py:63-73, 436-457, 492-508)
  pipe_diffuse_cfg_256  the reference `diffuse` loop incl. the true-CFG combine (:530-586) at BASELINE config 1's
                        shape (256x256 -> 16x16 tokens, 4 steps) driving the reference DiT
The scheduler class behind `prepare_timesteps` / `scheduler.step` is diffusers' (absent): restated stub, parity unpinned.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import qwen_image_oracle as O  # noqa: E402
import ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CASES = {
    # name: model kwargs, input shape, seeds
    "dit_small_fp32": dict(layers=2, heads=2, joint=128, grid=(8, 8), T=7, B=1, bias_std=0.0, jitter=0.0,
                           dtype="float32", sigma=[0.731]),
    "dit_rect_b2_fp32": dict(layers=3, heads=4, joint=192, grid=(16, 8), T=13, B=2, bias_std=0.02, jitter=0.1,
                             dtype="float32", sigma=[0.912, 0.237]),
    "dit_small_bf16": dict(layers=2, heads=2, joint=128, grid=(8, 8), T=7, B=1, bias_std=0.02, jitter=0.1,
                           dtype="bfloat16", sigma=[0.731]),
    "dit_fullwidth_1layer_fp32": dict(layers=1, heads=24, joint=3584, grid=(16, 16), T=16, B=1, bias_std=0.02,
                                      jitter=0.1, dtype="float32", sigma=[0.5]),
}


def params_checksum(P) -> str:
    h = hashlib.sha256()
    for k in sorted(P):
        h.update(k.encode())
        h.update(P[k].float().numpy().tobytes())
    return h.hexdigest()


def make_inputs(case, seed_lat=42, seed_txt=1):
    gh, gw = case["grid"]
    g1 = torch.Generator().manual_seed(seed_lat)
    g2 = torch.Generator().manual_seed(seed_txt)
    lat = torch.randn(case["B"], gh * gw, 64, generator=g1)
    txt = torch.randn(case["B"], case["T"], case["joint"], generator=g2)
    return lat, txt


def run_case(name, case):
    dtype = getattr(torch, case["dtype"])
    P = O.make_dit_params(case["layers"], seed=1234, bias_std=case["bias_std"], norm_jitter=case["jitter"],
                          num_heads=case["heads"], joint_dim=case["joint"])
    model, cfg = ref_shims.build_reference_model(case["layers"], num_attention_heads=case["heads"],
                                                 joint_attention_dim=case["joint"], dtype=torch.float32)
    ref_names = [n for n, _ in model.named_parameters()]
    assert ref_names == list(P.keys()), "oracle.dit_param_shapes order != reference named_parameters order"
    missing = model.load_state_dict(P, strict=True)
    model = model.to(dtype)
    lat, txt = make_inputs(case)
    sig = torch.tensor(case["sigma"], dtype=torch.float32)
    gh, gw = case["grid"]
    taps = {}

    def hook(i):
        def f(_m, _inp, out):
            taps[f"block{i}_enc"], taps[f"block{i}_hidden"] = out[0].float().numpy(), out[1].float().numpy()
        return f

    if name.startswith("dit_small"):  # per-block taps only where they stay small
        for i, blk in enumerate(model.transformer_blocks):
            blk.register_forward_hook(hook(i))
    out = ref_shims.reference_forward(
        model, cfg, hidden_states=lat.to(dtype), encoder_hidden_states=txt.to(dtype),
        encoder_hidden_states_mask=torch.ones(case["B"], case["T"], dtype=torch.long),
        timestep=sig.to(dtype), img_shapes=[[(1, gh, gw)]] * case["B"], txt_seq_lens=[case["T"]] * case["B"])
    meta = dict(case=case, params_sha256=params_checksum(P), param_seed=1234, seed_lat=42, seed_txt=1,
                reference="vllm_omni/diffusion/models/qwen_image/qwen_image_transformer.py:692-802 via oracle/ref_shims.py")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), latents=lat.numpy(), prompt_embeds=txt.numpy(),
                        sigma=sig.numpy(), noise_pred=out.float().numpy(), meta=json.dumps(meta), **taps)
    print(f"{name}: out {tuple(out.shape)} std {out.float().std():.4f} sha {meta['params_sha256'][:12]}")

    # cross-check the reference's own q/k/v stacking loader (qwen_image_transformer.py:804-839)
    if name == "dit_small_fp32":
        model2, _ = ref_shims.build_reference_model(case["layers"], num_attention_heads=case["heads"],
                                                    joint_attention_dim=case["joint"])
        D = case["heads"] * 128
        split = []
        for k, v in P.items():
            if ".to_qkv." in k:
                for j, s in enumerate(("to_q", "to_k", "to_v")):
                    split.append((k.replace("to_qkv", s), v[j * D:(j + 1) * D]))
            elif ".add_kv_proj." in k:
                for j, s in enumerate(("add_q_proj", "add_k_proj", "add_v_proj")):
                    split.append((k.replace("add_kv_proj", s), v[j * D:(j + 1) * D]))
            else:
                split.append((k, v))
        loaded = model2.load_weights(split)
        assert len(loaded) == len(P)
        for (n, p) in model2.named_parameters():
            assert torch.equal(p.data, P[n]), n
        print("  load_weights(q/k/v split) == fused state dict: OK")


VAE_CASES = {"vae_decode_16x16_fp32": dict(h=16, w=16, seed=3), "vae_decode_24x40_fp32": dict(h=24, w=40, seed=5)}


def run_vae_cases():
    import ref_shims_pipeline as RP

    vae = RP.build_reference_vae()
    Pv = O.make_vae_params()
    dec_names = [n for n, _ in vae.named_parameters() if n.startswith(("decoder.", "post_quant_conv."))]
    assert sorted(dec_names) == sorted(Pv.keys()), "oracle.vae_decoder_param_shapes != reference decoder parameters"
    sd = vae.state_dict()
    for k, v in Pv.items():
        assert sd[k].shape == v.shape, k
        sd[k].copy_(v)
    for name, c in VAE_CASES.items():
        z = torch.randn(1, 16, 1, c["h"], c["w"], generator=torch.Generator().manual_seed(c["seed"]))
        with torch.no_grad():
            img = vae.decode(z, return_dict=False)[0]
        meta = dict(case=c, params_sha256=params_checksum(Pv), param_seed=4321,
                    reference="vllm_omni/diffusion/models/qwen_image/autoencoder_kl_qwenimage.py:839-887 via oracle/ref_shims_pipeline.py")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), z=z.numpy(), image=img.numpy().astype(np.float32),
                            meta=json.dumps(meta))
        print(f"{name}: image {tuple(img.shape)} std {img.std():.4f} clamp-frac {float((img.abs() >= 1).float().mean()):.3f}")


def run_vae_encode_case():
    """The reference's vendored AutoencoderKLQwenImage.encode(...).mode() (autoencoder_kl_qwenimage.py:788-835)."""
    import ref_shims_pipeline as RP

    vae = RP.build_reference_vae()
    Pe = O.make_vae_encoder_params()
    ref_names = {n for n, _ in vae.named_parameters() if n.startswith(("encoder.", "quant_conv.")) and "time_conv" not in n}
    assert ref_names == set(Pe), "oracle.vae_encoder_param_shapes != reference encoder parameters (minus time_conv)"
    sd = vae.state_dict()
    for k, v in Pe.items():
        assert sd[k].shape == v.shape, k
        sd[k].copy_(v)
    img = torch.rand(1, 3, 1, 64, 96, generator=torch.Generator().manual_seed(2)) * 2 - 1
    with torch.no_grad():
        mean = vae.encode(img, return_dict=False)[0].mode()
    meta = dict(case=dict(h=64, w=96, seed=2), params_sha256=params_checksum(Pe), param_seed=8765,
                reference="autoencoder_kl_qwenimage.py:788-835 (+ DiagonalGaussianDistribution.mode) via oracle/ref_shims_pipeline.py")
    np.savez_compressed(os.path.join(OUT, "vae_encode_64x96_fp32.npz"), image=img.numpy(), mean=mean.numpy(), meta=json.dumps(meta))
    print(f"vae_encode_64x96_fp32: mean {tuple(mean.shape)} std {mean.std():.4f}")


def run_vae_tiled_case():
    """The reference's vendored VAE with spatial tiling on (`vae.enable_tiling()`; od_config.vae_use_tiling sets `use_tiling`,
    registry.py:88-92): tiled_decode (autoencoder_kl_qwenimage.py:971-1031) of a 36 x 40 latent (2 x 2 tiles of 32 x 32 every
    24, both blends) and tiled_encode (:905-969) of a 288 x 320 image."""
    import ref_shims_pipeline as RP

    vae = RP.build_reference_vae()
    Pd, Pe = O.make_vae_params(), O.make_vae_encoder_params()
    sd = vae.state_dict()
    for k, v in list(Pd.items()) + list(Pe.items()):
        assert sd[k].shape == v.shape, k
        sd[k].copy_(v)
    vae.enable_tiling()
    z = torch.randn(1, 16, 1, 36, 40, generator=torch.Generator().manual_seed(13))
    img_in = torch.rand(1, 3, 1, 288, 320, generator=torch.Generator().manual_seed(14)) * 2 - 1
    with torch.no_grad():
        img = vae.decode(z, return_dict=False)[0]
        mean = vae.encode(img_in, return_dict=False)[0].mode()
    meta = dict(case=dict(latent=[36, 40], tile_sample_min=256, tile_sample_stride=192, seeds=[13, 14]),
                params_sha256=params_checksum({**Pd, **Pe}),
                reference="autoencoder_kl_qwenimage.py:742-770 (enable_tiling), :844-845 -> :971-1031 (tiled_decode, un-clamped), "
                          ":791-792 -> :905-969 (tiled_encode), :889-903 (blend) via oracle/ref_shims_pipeline.py")
    np.savez_compressed(os.path.join(OUT, "vae_tiled_36x40_fp32.npz"), z=z.numpy(), image=img.numpy().astype(np.float32),
                        mean=mean.numpy(), meta=json.dumps(meta))       # the encoder's input image is torch.rand(seed 14) * 2 - 1 (CPU generator)
    print(f"vae_tiled_36x40_fp32: image {tuple(img.shape)} std {img.std():.4f} |max| {float(img.abs().max()):.3f}; "
          f"mean {tuple(mean.shape)} std {mean.std():.4f}")


def run_edit_case():
    """Reference DiT forward over a TWO-image sequence (target latents + one condition image of another size), the way the
    Edit pipelines call it: img_shapes = [[(1, h, w), (1, h2, w2)]] -> per-image RoPE frame index
    (qwen_image_transformer.py:231-250; pipeline_qwen_image_edit.py:600-632,770)."""
    case = dict(layers=2, heads=2, joint=128, grids=[(1, 8, 8), (1, 6, 10)], T=9, B=1, bias_std=0.02, jitter=0.1,
                dtype="float32", sigma=[0.62])
    P = O.make_dit_params(case["layers"], seed=1234, bias_std=case["bias_std"], norm_jitter=case["jitter"],
                          num_heads=case["heads"], joint_dim=case["joint"])
    model, cfg = ref_shims.build_reference_model(case["layers"], num_attention_heads=case["heads"],
                                                 joint_attention_dim=case["joint"], dtype=torch.float32)
    model.load_state_dict(P, strict=True)
    S = sum(f * h * w for f, h, w in case["grids"])
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, S, 64, generator=g)
    txt = torch.randn(1, case["T"], case["joint"], generator=g)
    sig = torch.tensor(case["sigma"])
    out = ref_shims.reference_forward(
        model, cfg, hidden_states=lat, encoder_hidden_states=txt,
        encoder_hidden_states_mask=torch.ones(1, case["T"], dtype=torch.long), timestep=sig,
        img_shapes=[[tuple(gr) for gr in case["grids"]]], txt_seq_lens=[case["T"]])
    meta = dict(case=case, params_sha256=params_checksum(P), param_seed=1234,
                reference="qwen_image_transformer.py:692-802 with a two-entry img_shapes via oracle/ref_shims.py")
    np.savez_compressed(os.path.join(OUT, "dit_edit_two_images_fp32.npz"), latents=lat.numpy(), prompt_embeds=txt.numpy(),
                        sigma=sig.numpy(), noise_pred=out.float().numpy(), meta=json.dumps(meta))
    print(f"dit_edit_two_images_fp32: out {tuple(out.shape)} std {out.std():.4f}")


def run_pipeline_helpers():
    import ref_shims_pipeline as RP

    mod = RP.load_reference_pipeline_module()
    pipe, _ = RP.reference_pipeline_shell(None, None)
    out = {}
    seqs = np.array([64, 256, 1024, 4096, 8192, 16384], dtype=np.int64)
    out["shift_seq"] = seqs
    out["shift_default"] = np.array([mod.calculate_shift(int(s)) for s in seqs], dtype=np.float64)
    out["shift_qwen"] = np.array([mod.calculate_shift(int(s), 256, 8192, 0.5, 0.9) for s in seqs], dtype=np.float64)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(2, 16, 12, 20, generator=g)
    packed = mod.QwenImagePipeline._pack_latents(lat, 2, 16, 12, 20)
    out["pack_in"], out["pack_out"] = lat.numpy(), packed.numpy()
    out["unpack_out"] = mod.QwenImagePipeline._unpack_latents(packed, 96, 160, 8).numpy()
    combos = [(4, 256), (20, 4096), (50, 4096), (50, 16384), (2, 64), (3, 1024)]   # N=1 is 0/0 in the terminal stretch (NaN in diffusers too)
    out["ts_combos"] = np.array(combos, dtype=np.int64)
    for n, s in combos:
        ts, nn_ = mod.QwenImagePipeline.prepare_timesteps(pipe, n, None, s)
        assert nn_ == n
        out[f"timesteps_{n}_{s}"] = ts.numpy()
        out[f"sigmas_{n}_{s}"] = pipe.scheduler.sigmas.numpy()
    meta = dict(reference="pipeline_qwen_image.py:63-73,436-457,492-508 via oracle/ref_shims_pipeline.py; scheduler = restated stub",
                scheduler_config=pipe.scheduler.config)
    np.savez_compressed(os.path.join(OUT, "pipe_helpers.npz"), meta=json.dumps(meta), **out)
    print("pipe_helpers: ", {k: v.shape for k, v in out.items() if k.startswith("timesteps")})


def run_diffuse_case():
    """Reference diffuse() (true-CFG on, 4 steps) over the reference DiT at BASELINE config 1's token grid."""
    import ref_shims_pipeline as RP

    case = dict(layers=2, heads=2, joint=128, grid=(16, 16), T=9, Tneg=5, steps=4, cfg=4.0, bias_std=0.02, jitter=0.1)
    P = O.make_dit_params(case["layers"], seed=1234, bias_std=case["bias_std"], norm_jitter=case["jitter"],
                          num_heads=case["heads"], joint_dim=case["joint"])
    model, cfg = ref_shims.build_reference_model(case["layers"], num_attention_heads=case["heads"],
                                                 joint_attention_dim=case["joint"], dtype=torch.float32)
    model.load_state_dict(P, strict=True)
    pipe, mod = RP.reference_pipeline_shell(model, cfg)
    gh, gw = case["grid"]
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, gh * gw, 64, generator=g)
    pos = torch.randn(1, case["T"], case["joint"], generator=g)
    neg = torch.randn(1, case["Tneg"], case["joint"], generator=g)
    timesteps, _ = mod.QwenImagePipeline.prepare_timesteps(pipe, case["steps"], None, gh * gw)
    traj = []
    orig_step = pipe.scheduler.step

    def tap(*a, **k):
        r = orig_step(*a, **k)
        traj.append(r[0].clone())
        return r

    pipe.scheduler.step = tap
    final = RP.reference_diffuse(
        pipe, cfg, prompt_embeds=pos, prompt_embeds_mask=torch.ones(1, case["T"], dtype=torch.long),
        negative_prompt_embeds=neg, negative_prompt_embeds_mask=torch.ones(1, case["Tneg"], dtype=torch.long),
        latents=lat, img_shapes=[[(1, gh, gw)]], txt_seq_lens=[case["T"]], negative_txt_seq_lens=[case["Tneg"]],
        timesteps=timesteps, do_true_cfg=True, guidance=None, true_cfg_scale=case["cfg"])
    meta = dict(case=case, params_sha256=params_checksum(P), param_seed=1234,
                reference="pipeline_qwen_image.py:530-586 over qwen_image_transformer.py:692-802 via oracle/ref_shims*.py")
    np.savez_compressed(os.path.join(OUT, "pipe_diffuse_cfg_256.npz"), latents=lat.numpy(), pos=pos.numpy(),
                        neg=neg.numpy(), timesteps=timesteps.numpy(), sigmas=pipe.scheduler.sigmas.numpy(),
                        trajectory=torch.stack(traj).numpy(), final=final.numpy(), meta=json.dumps(meta))
    print(f"pipe_diffuse_cfg_256: final std {final.std():.4f}, steps {len(traj)}")


def run_edit_plus_case():
    """(a) Reference DiT forward over a THREE-image sequence (target + two condition images of different sizes), the way
    the Edit-Plus pipeline calls it (pipeline_qwen_image_edit_plus.py:543-556,729-738).  (b) The Edit-Plus pre-process
    arithmetic and prompt template, taken from the unmodified reference sources by executing ONLY their constant
    assignments and the `calculate_dimensions` function (the module itself imports PIL / transformers / diffusers)."""
    import ast

    case = dict(layers=2, heads=2, joint=128, grids=[(1, 8, 8), (1, 6, 10), (1, 4, 4)], T=9, B=1, bias_std=0.02, jitter=0.1,
                dtype="float32", sigma=[0.37])
    P = O.make_dit_params(case["layers"], seed=1234, bias_std=case["bias_std"], norm_jitter=case["jitter"],
                          num_heads=case["heads"], joint_dim=case["joint"])
    model, cfg = ref_shims.build_reference_model(case["layers"], num_attention_heads=case["heads"],
                                                 joint_attention_dim=case["joint"], dtype=torch.float32)
    model.load_state_dict(P, strict=True)
    S = sum(f * h * w for f, h, w in case["grids"])
    g = torch.Generator().manual_seed(43)
    lat = torch.randn(1, S, 64, generator=g)
    txt = torch.randn(1, case["T"], case["joint"], generator=g)
    sig = torch.tensor(case["sigma"])
    out = ref_shims.reference_forward(
        model, cfg, hidden_states=lat, encoder_hidden_states=txt,
        encoder_hidden_states_mask=torch.ones(1, case["T"], dtype=torch.long), timestep=sig,
        img_shapes=[[tuple(gr) for gr in case["grids"]]], txt_seq_lens=[case["T"]])
    # --- constants / helpers of the reference Edit(-Plus) modules
    rdir = os.path.join(ref_shims.REFERENCE_ROOT, "vllm_omni", "diffusion", "models", "qwen_image")
    ns: dict = {"math": __import__("math")}
    tree = ast.parse(open(os.path.join(rdir, "pipeline_qwen_image_edit.py")).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "calculate_dimensions":
            exec(compile(ast.Module([node], []), "ref_edit", "exec"), ns)
    tree = ast.parse(open(os.path.join(rdir, "pipeline_qwen_image_edit_plus.py")).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in ("CONDITION_IMAGE_SIZE", "VAE_IMAGE_SIZE"):
            exec(compile(ast.Module([node], []), "ref_edit_plus", "exec"), ns)
    template = img_template = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute) and node.targets[0].attr == "prompt_template_encode":
            template = ast.literal_eval(node.value)
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "img_prompt_template":
            img_template = ast.literal_eval(node.value)
    assert template and img_template
    sizes = [(1920, 1080), (640, 480), (512, 512), (300, 900)]
    helpers = dict(CONDITION_IMAGE_SIZE=ns["CONDITION_IMAGE_SIZE"], VAE_IMAGE_SIZE=ns["VAE_IMAGE_SIZE"], sizes=sizes,
                   condition=[list(ns["calculate_dimensions"](ns["CONDITION_IMAGE_SIZE"], w / h)) for w, h in sizes],
                   vae=[list(ns["calculate_dimensions"](ns["VAE_IMAGE_SIZE"], w / h)) for w, h in sizes],
                   prompt_2_images=template.format("".join(img_template.format(i + 1) for i in range(2)) + "make it snow"))
    meta = dict(case=case, params_sha256=params_checksum(P), param_seed=1234, helpers=helpers,
                reference="qwen_image_transformer.py:692-802 with a three-entry img_shapes via oracle/ref_shims.py; "
                          "pipeline_qwen_image_edit.py:124-132 and pipeline_qwen_image_edit_plus.py:44-45,203-209,286-299 via ast")
    np.savez_compressed(os.path.join(OUT, "dit_edit_plus_three_images_fp32.npz"), latents=lat.numpy(), prompt_embeds=txt.numpy(),
                        sigma=sig.numpy(), noise_pred=out.float().numpy(), meta=json.dumps(meta))
    print(f"dit_edit_plus_three_images_fp32: out {tuple(out.shape)} std {out.std():.4f}; helpers {helpers['vae']}")


def run_teacache_case():
    """The reference's OWN TeaCache (cache/teacache/hook.py:82-217 `TeaCacheHook.new_forward` +
    `_should_compute_full_transformer`, extractors.py:145-261 `extract_qwen_context`, state.py, config.py, hooks.py — all
    torch/numpy only) applied with `apply_teacache_hook` to the reference DiT and driven by the reference `diffuse` loop
    (true-CFG: the hook alternates positive / negative states).  Stored: the compute / skip decision of every forward per
    branch, the rel-L1 distances behind them, the trajectory and the final latent."""
    import importlib

    import ref_shims_pipeline as RP
    from ref_shims import _pkg

    case = dict(layers=2, heads=2, joint=128, grid=(16, 16), T=9, Tneg=5, steps=10, cfg=4.0, bias_std=0.02, jitter=0.1,
                rel_l1_thresh=0.15, dtype="float32")
    P = O.make_dit_params(case["layers"], seed=1234, bias_std=case["bias_std"], norm_jitter=case["jitter"],
                          num_heads=case["heads"], joint_dim=case["joint"])
    model, cfg = ref_shims.build_reference_model(case["layers"], num_attention_heads=case["heads"],
                                                 joint_attention_dim=case["joint"], dtype=torch.float32)
    model.load_state_dict(P, strict=True)
    RP.install()
    _pkg("vllm_omni.diffusion.cache.teacache", os.path.join(ref_shims.REFERENCE_ROOT, "vllm_omni", "diffusion", "cache", "teacache"))
    hookmod = importlib.import_module("vllm_omni.diffusion.cache.teacache.hook")
    cfgmod = importlib.import_module("vllm_omni.diffusion.cache.teacache.config")
    tcfg = cfgmod.TeaCacheConfig(transformer_type="QwenImageTransformer2DModel", rel_l1_thresh=case["rel_l1_thresh"])
    hookmod.apply_teacache_hook(model, tcfg)
    hook = model._hook_registry.get_hook("teacache")
    decisions = {"positive": [], "negative": []}
    rels = {"positive": [], "negative": []}
    orig = hook._should_compute_full_transformer

    def tap(state, modulated):
        branch = hook.state_manager._context.split("_")[-1]
        rel = float("nan")
        if state.cnt > 0 and state.previous_modulated_input is not None:
            rel = float(((modulated - state.previous_modulated_input).abs().mean()
                         / (state.previous_modulated_input.abs().mean() + 1e-8)).item())
        r = orig(state, modulated)
        decisions[branch].append(bool(r))
        rels[branch].append(rel)
        return r

    hook._should_compute_full_transformer = tap
    pipe, mod = RP.reference_pipeline_shell(model, cfg)
    model.do_true_cfg = True                       # what the reference pipeline sets before diffuse (:729-731)
    gh, gw = case["grid"]
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, gh * gw, 64, generator=g)
    pos = torch.randn(1, case["T"], case["joint"], generator=g)
    neg = torch.randn(1, case["Tneg"], case["joint"], generator=g)
    timesteps, _ = mod.QwenImagePipeline.prepare_timesteps(pipe, case["steps"], None, gh * gw)
    traj = []
    orig_step = pipe.scheduler.step

    def step_tap(*a, **k):
        r = orig_step(*a, **k)
        traj.append(r[0].clone())
        return r

    pipe.scheduler.step = step_tap
    final = RP.reference_diffuse(
        pipe, cfg, prompt_embeds=pos, prompt_embeds_mask=torch.ones(1, case["T"], dtype=torch.long),
        negative_prompt_embeds=neg, negative_prompt_embeds_mask=torch.ones(1, case["Tneg"], dtype=torch.long),
        latents=lat, img_shapes=[[(1, gh, gw)]], txt_seq_lens=[case["T"]], negative_txt_seq_lens=[case["Tneg"]],
        timesteps=timesteps, do_true_cfg=True, guidance=None, true_cfg_scale=case["cfg"])
    meta = dict(case=case, params_sha256=params_checksum(P), param_seed=1234, coefficients=list(tcfg.coefficients),
                reference="cache/teacache/hook.py:82-217 + extractors.py:145-261 + state.py + hooks.py over "
                          "qwen_image_transformer.py and pipeline_qwen_image.py:530-586 via oracle/ref_shims*.py")
    np.savez_compressed(os.path.join(OUT, "teacache_diffuse_cfg_256.npz"), latents=lat.numpy(), pos=pos.numpy(), neg=neg.numpy(),
                        timesteps=timesteps.numpy(), trajectory=torch.stack(traj).numpy(), final=final.numpy(),
                        compute_pos=np.array(decisions["positive"]), compute_neg=np.array(decisions["negative"]),
                        rel_pos=np.array(rels["positive"]), rel_neg=np.array(rels["negative"]), meta=json.dumps(meta))
    print("teacache_diffuse_cfg_256: compute pattern pos", "".join("C" if d else "s" for d in decisions["positive"]),
          "neg", "".join("C" if d else "s" for d in decisions["negative"]))
    print("   rel pos", np.round(rels["positive"], 4).tolist())
    print("   rel neg", np.round(rels["negative"], 4).tolist())


class _RefTokenizerStub:
    """Tokenizer stand-in for the prompt-encoding fixture: whitespace words -> ids, with the reference template's prefix
    ("<|im_start|>system ... <|im_start|>user\n") mapped to EXACTLY 34 ids, which is what the Qwen2 tokenizer produces and
    what `prompt_template_encode_start_idx = 34` (pipeline_qwen_image.py:282) relies on; the template's suffix
    ("<|im_end|>\n<|im_start|>assistant\n") is 5 ids as under Qwen2.  Deterministic, no vocabulary file."""

    PREFIX_IDS, SUFFIX_IDS, vocab_size, pad_token_id = 34, 5, 512, 0

    def __init__(self, template: str):
        self.prefix, self.suffix = template.split("{}")

    def _ids(self, text: str):
        assert text.startswith(self.prefix) and text.endswith(self.suffix), "prompt was not wrapped in the reference template"
        body = text[len(self.prefix): len(text) - len(self.suffix)]
        words = body.split()
        mid = [3 + (sum(w.encode()) * 7 + len(w)) % 400 for w in words]
        return [410 + i for i in range(self.PREFIX_IDS)] + mid + [460 + i for i in range(self.SUFFIX_IDS)]

    def __call__(self, text, max_length=None, padding=True, truncation=True, return_tensors="pt"):
        ids = [self._ids(t) for t in ([text] if isinstance(text, str) else text)]
        if truncation and max_length:
            ids = [x[:max_length] for x in ids]
        L = max(len(x) for x in ids)
        out = types.SimpleNamespace(
            input_ids=torch.tensor([x + [0] * (L - len(x)) for x in ids], dtype=torch.long),
            attention_mask=torch.tensor([[1] * len(x) + [0] * (L - len(x)) for x in ids], dtype=torch.long))
        out.to = lambda device: out
        return out


def run_prompt_encode_case():
    """The reference's `_extract_masked_hidden` / `_get_qwen_prompt_embeds` / `encode_prompt` (pipeline_qwen_image.py:
    351-433) called UNBOUND on a bare pipeline object whose `text_encoder` is a seeded random HF `Qwen2_5_VLTextModel`
    (real width 3584, 2 layers) and whose tokenizer is `_RefTokenizerStub`.  Stored: token ids, embeddings, masks."""
    import ref_shims_pipeline as RP
    from transformers import Qwen2_5_VLTextConfig, Qwen2_5_VLTextModel

    pipe, mod = RP.reference_pipeline_shell(None, None)
    torch.manual_seed(7)
    tcfg = Qwen2_5_VLTextConfig(vocab_size=_RefTokenizerStub.vocab_size, hidden_size=3584, num_hidden_layers=2,
                                num_attention_heads=28, num_key_value_heads=4, intermediate_size=512,
                                max_position_embeddings=4096, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                                rope_scaling={"type": "default", "mrope_section": [16, 24, 24], "rope_type": "default"})
    te = Qwen2_5_VLTextModel(tcfg).float().eval()
    pipe.text_encoder = te
    pipe.tokenizer_max_length = 1024
    pipe.prompt_template_encode = ("<|im_start|>system\nDescribe the image by detailing the color, shape, size, texture, quantity, "
                                   "text, spatial relationships of the objects and background:<|im_end|>\n<|im_start|>user\n{}"
                                   "<|im_end|>\n<|im_start|>assistant\n")
    # take template + start index from the reference SOURCE (its __init__, :280-282), not from this file
    import ast
    src = open(os.path.join(ref_shims.REFERENCE_ROOT, "vllm_omni", "diffusion", "models", "qwen_image", "pipeline_qwen_image.py")).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute):
            if node.targets[0].attr == "prompt_template_encode":
                pipe.prompt_template_encode = ast.literal_eval(node.value)
            if node.targets[0].attr == "prompt_template_encode_start_idx":
                pipe.prompt_template_encode_start_idx = ast.literal_eval(node.value)
            if node.targets[0].attr == "tokenizer_max_length":
                pipe.tokenizer_max_length = ast.literal_eval(node.value)
    assert pipe.prompt_template_encode_start_idx == 34
    pipe.tokenizer = _RefTokenizerStub(pipe.prompt_template_encode)
    pipe.device = torch.device("cpu")
    prompts = ["a red fox jumping over a frozen lake at dawn", "two cats", ""]
    with torch.no_grad():
        emb, msk = mod.QwenImagePipeline._get_qwen_prompt_embeds(pipe, prompts, dtype=torch.float32)
        emb2, msk2 = mod.QwenImagePipeline.encode_prompt(pipe, prompts[:2], num_images_per_prompt=2, max_sequence_length=6)
    toks = pipe.tokenizer([pipe.prompt_template_encode.format(e) for e in prompts])
    sd = {k: v.numpy() for k, v in te.state_dict().items()}
    meta = dict(prompts=prompts, drop_idx=34, text_config=tcfg.to_dict(), seed=7, template=pipe.prompt_template_encode,
                reference="pipeline_qwen_image.py:351-433 (unbound) over HF Qwen2_5_VLTextModel via oracle/ref_shims_pipeline.py")
    np.savez_compressed(os.path.join(OUT, "prompt_encode.npz"), input_ids=toks.input_ids.numpy(),
                        attention_mask=toks.attention_mask.numpy(), prompt_embeds=emb.numpy(), prompt_embeds_mask=msk.numpy(),
                        encode_prompt_embeds=emb2.numpy(), encode_prompt_mask=msk2.numpy(),
                        te_checksum=np.array([float(sum(float(np.abs(v).sum()) for v in sd.values()))]), meta=json.dumps(meta, default=str))
    print(f"prompt_encode: embeds {tuple(emb.shape)} mask rows {msk.sum(1).tolist()}; encode_prompt {tuple(emb2.shape)}")


def run_edit_prompt_encode_case():
    """The reference's Edit / Edit-Plus `_get_qwen_prompt_embeds` (pipeline_qwen_image_edit.py:352-397,
    pipeline_qwen_image_edit_plus.py:274-330) called UNBOUND on bare pipeline objects whose `text_encoder` is a seeded random HF
    `Qwen2_5_VLForConditionalGeneration` (vision tower + language model) and whose `processor` is oracle/vl_stubs.StubVLProcessor
    (HF's real Qwen2VL image processor + a word tokenizer with a 64-id template prefix).  Templates and start indices are read
    from the reference SOURCE."""
    import ast
    import importlib

    import ref_shims_pipeline as RP
    import vl_stubs as V

    RP.install()
    model, cfg = V.make_random_vl_model(seed=5)
    out = {}
    imgs = V.test_images(3)
    prompts = {"edit": "turn the sky purple and add two birds", "plus": "put the cat from picture two onto the sofa of picture one"}
    for key, modname, clsname, pics in (("edit", "pipeline_qwen_image_edit", "QwenImageEditPipeline", imgs[0]),
                                        ("plus", "pipeline_qwen_image_edit_plus", "QwenImageEditPlusPipeline", imgs[:3]),
                                        ("plus1", "pipeline_qwen_image_edit_plus", "QwenImageEditPlusPipeline", imgs[1])):
        mod = importlib.import_module(f"vllm_omni.diffusion.models.qwen_image.{modname}")
        cls = getattr(mod, clsname)
        pipe = object.__new__(cls)
        torch.nn.Module.__init__(pipe)
        src = open(os.path.join(ref_shims.REFERENCE_ROOT, "vllm_omni", "diffusion", "models", "qwen_image", modname + ".py")).read()
        for node in ast.walk(ast.parse(src)):
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute) and \
                    node.targets[0].attr in ("prompt_template_encode", "prompt_template_encode_start_idx"):
                setattr(pipe, node.targets[0].attr, ast.literal_eval(node.value))
        assert pipe.prompt_template_encode_start_idx == 64
        pipe.text_encoder, pipe.processor, pipe.device = model, V.StubVLProcessor(), torch.device("cpu")
        with torch.no_grad():
            emb, msk = cls._get_qwen_prompt_embeds(pipe, prompts[key[:4]], image=pics, dtype=torch.float32)
        out[f"{key}_embeds"], out[f"{key}_mask"] = emb.numpy(), msk.numpy()
        out[f"{key}_text"] = np.array(pipe.processor.calls[-1]["text"][0])
        out[f"{key}_template"] = np.array(pipe.prompt_template_encode)
        print(f"prompt_encode_edit[{key}]: embeds {tuple(emb.shape)} (vision tokens + text behind the 64-token prefix)")
    sd = model.state_dict()
    meta = dict(prompts=prompts, seed=5, drop_idx=64, n_images=dict(edit=1, plus=3, plus1=1),
                reference="pipeline_qwen_image_edit.py:352-397 and pipeline_qwen_image_edit_plus.py:274-330 (unbound) over HF "
                          "Qwen2_5_VLForConditionalGeneration + oracle/vl_stubs.py")
    np.savez_compressed(os.path.join(OUT, "prompt_encode_edit.npz"),
                        vl_checksum=np.array([float(sum(float(v.abs().sum()) for v in sd.values()))]),
                        meta=json.dumps(meta), **out)


def layered_params(case):
    """make_dit_params + the Layered variant's `time_text_embed.addition_t_embedding.weight` [2, D] (seeded)."""
    P = O.make_dit_params(case["layers"], seed=1234, bias_std=case["bias_std"], norm_jitter=case["jitter"],
                          num_heads=case["heads"], joint_dim=case["joint"])
    g = torch.Generator().manual_seed(4321)
    P["time_text_embed.addition_t_embedding.weight"] = torch.randn(2, case["heads"] * 128, generator=g) * 0.5
    return P


def run_layered_case():
    """The Layered variant, the buildable part (zero_cond_t = False): the reference DiT built with
    use_additional_t_cond=True, use_layer3d_rope=True (qwen_image_transformer.py:47-62, 65-176) over the Layered pipeline's
    sequence [layers + 1 generated frames ; condition image]; the pipeline's own helpers (_pack_latents / _unpack_latents /
    calculate_dimensions / retrieve_timesteps with mu = sqrt(S_cond / 256)) and its diffuse() loop
    (pipeline_qwen_image_layered.py:518-536, 538-615, 808-816) with cfg_normalize off (its default) and on."""
    import ref_shims_pipeline as RP

    case = dict(layers=2, heads=2, joint=128, img_layers=2, gen_grid=(8, 6), cond_grid=(6, 10), T=9, Tneg=5, steps=3, cfg=4.0,
                bias_std=0.02, jitter=0.1)
    P = layered_params(case)
    model, cfg = ref_shims.build_reference_model(case["layers"], num_attention_heads=case["heads"],
                                                 joint_attention_dim=case["joint"], dtype=torch.float32,
                                                 use_additional_t_cond=True, use_layer3d_rope=True, zero_cond_t=False)
    assert [n for n, _ in model.named_parameters()].count("time_text_embed.addition_t_embedding.weight") == 1
    model.load_state_dict(P, strict=True)
    gh, gw = case["gen_grid"]
    ch, cw = case["cond_grid"]
    nl = case["img_layers"]
    shapes = [(1, gh, gw)] * (nl + 1) + [(1, ch, cw)]
    S_gen, S_c = (nl + 1) * gh * gw, ch * cw
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, S_gen, 64, generator=g)
    img_lat = torch.randn(1, S_c, 64, generator=g)
    pos = torch.randn(1, case["T"], case["joint"], generator=g)
    neg = torch.randn(1, case["Tneg"], case["joint"], generator=g)
    out = {}
    # ---- one forward over a batch of two, additional_t_cond = [0, 1]
    x2 = torch.cat([torch.cat([lat, img_lat], 1), torch.cat([lat.flip(1), img_lat], 1)])
    fwd = ref_shims.reference_forward(
        model, cfg, hidden_states=x2, encoder_hidden_states=torch.cat([pos, pos.flip(1)]),
        encoder_hidden_states_mask=torch.ones(2, case["T"], dtype=torch.long), timestep=torch.tensor([0.62, 0.62]),
        img_shapes=[shapes] * 2, txt_seq_lens=[case["T"]] * 2, additional_t_cond=torch.tensor([0, 1]))
    out["fwd_in"], out["fwd_txt"], out["fwd_out"] = x2.numpy(), torch.cat([pos, pos.flip(1)]).numpy(), fwd.numpy()
    # ---- helpers
    pipe, mod = RP.reference_layered_shell(model, cfg)
    L = mod.QwenImageLayeredPipeline
    raw = torch.randn(2, nl + 1, 16, 2 * gh, 2 * gw, generator=g)
    packed = L._pack_latents(raw, 2, 16, 2 * gh, 2 * gw, nl + 1)
    out["pack_in"], out["pack_out"] = raw.numpy(), packed.numpy()
    out["unpack_out"] = L._unpack_latents(packed, 16 * gh, 16 * gw, nl, 8).numpy()
    ratios = [1.0, 0.75, 1.7777, 0.5]
    out["dims_ratio"] = np.array(ratios)
    out["dims_640"] = np.array([mod.calculate_dimensions(640 * 640, r) for r in ratios])
    out["dims_1024"] = np.array([mod.calculate_dimensions(1024 * 1024, r) for r in ratios])
    sig_in = np.linspace(1.0, 0, case["steps"] + 1)[:-1]
    mu = (S_c / (256 * 256 / 16 / 16)) ** 0.5
    timesteps, n = mod.retrieve_timesteps(pipe.scheduler, case["steps"], None, sigmas=sig_in, mu=mu)
    assert n == case["steps"]
    out["timesteps"], out["sigmas"], out["mu"] = timesteps.numpy(), pipe.scheduler.sigmas.numpy(), np.array(mu)
    # ---- diffuse, cfg_normalize off / on
    for norm in (False, True):
        traj = []
        sch = RP.FlowMatchEulerDiscreteSchedulerStub()
        pipe.scheduler = sch
        timesteps, _ = mod.retrieve_timesteps(sch, case["steps"], None, sigmas=sig_in, mu=mu)
        orig_step = sch.step

        def tap(*a, _o=orig_step, **k):
            r = _o(*a, **k)
            traj.append(r[0].clone())
            return r

        sch.step = tap
        fc = __import__("importlib").import_module("vllm_omni.diffusion.forward_context")
        with torch.no_grad(), fc.set_forward_context(omni_diffusion_config=cfg):
            final = L.diffuse(pipe, pos, torch.ones(1, case["T"], dtype=torch.long), neg, torch.ones(1, case["Tneg"], dtype=torch.long),
                              lat, img_lat, [shapes], [case["T"]], [case["Tneg"]], timesteps, True, None, case["cfg"], norm,
                              torch.tensor([0], dtype=torch.long))
        tag = "norm" if norm else "plain"
        out[f"final_{tag}"], out[f"traj_{tag}"] = final.numpy(), torch.stack(traj).numpy()
    meta = dict(case=case, params_sha256=params_checksum(P), param_seed=1234, t_embed_seed=4321,
                reference="qwen_image_transformer.py:47-62,65-176,692-802 (use_additional_t_cond, use_layer3d_rope) and "
                          "pipeline_qwen_image_layered.py:518-615,808-816 via oracle/ref_shims*.py; scheduler = restated stub")
    np.savez_compressed(os.path.join(OUT, "layered_dit_and_pipeline.npz"), latents=lat.numpy(), image_latents=img_lat.numpy(),
                        pos=pos.numpy(), neg=neg.numpy(), meta=json.dumps(meta), **out)
    print(f"layered_dit_and_pipeline: fwd std {fwd.std():.4f}, final(plain) std {out['final_plain'].std():.4f}, "
          f"final(norm) std {out['final_norm'].std():.4f}, timesteps {timesteps.tolist()}")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:] or None
    for name, case in CASES.items():
        if only is None or name in only:
            run_case(name, case)
    if only is None or "vae" in only:
        run_vae_cases()
    if only is None or "helpers" in only:
        run_pipeline_helpers()
    if only is None or "diffuse" in only:
        run_diffuse_case()
    if only is None or "encode" in only:
        run_vae_encode_case()
    if only is None or "vaetiled" in only:
        run_vae_tiled_case()
    if only is None or "edit" in only:
        run_edit_case()
    if only is None or "editplus" in only:
        run_edit_plus_case()
    if only is None or "teacache" in only:
        run_teacache_case()
    if only is None or "prompt" in only:
        run_prompt_encode_case()
    if only is None or "editprompt" in only:
        run_edit_prompt_encode_case()
    if only is None or "layered" in only:
        run_layered_case()


if __name__ == "__main__":
    main()
