#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the UNMODIFIED reference DiT.

Run in the authoring container only (needs /root/reference):

    python oracle/gen_golden.py

For each case it builds the reference `QwenImageTransformer2DModel` through oracle/ref_shims.py,
loads seeded synthetic weights (oracle.make_dit_params — the same generator the tests use, so the
weights are NOT stored, only a checksum), runs the reference forward on seeded inputs and stores
inputs + outputs (+ per-block outputs captured with forward hooks).  The fixtures pin
oracle/qwen_image_oracle.py (tests/test_oracle_golden.py) and are the targets of the GPU parity
tests (tests/test_gpu_parity.py).  Nothing here travels to the GPU box except the .npz files.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

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


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    for name, case in CASES.items():
        run_case(name, case)


if __name__ == "__main__":
    main()
