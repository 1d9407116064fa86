"""GPU: the fp8 mode of the DiT (BASELINE.json config 5: e4m3 operands for the eight block GEMMs of every layer).

The reference has no fp8 path, so there is no reference output to be identical to: the bar is (a) the fp8 GEMM equals the
product of its own dequantised operands (tests/test_gpu_ops.py, bit-level), and (b) HERE: how far the fp8 forward / denoise
loop drifts from the bf16 product path and from the fp32 oracle, with the tolerance written down."""
import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def _model(layers=3, heads=2, joint=256):
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    P = O.make_dit_params(layers, seed=1234, bias_std=0.02, norm_jitter=0.1, num_heads=heads, joint_dim=joint)
    m = QwenImageTransformer2DModel(num_layers=layers, num_attention_heads=heads, joint_attention_dim=joint, device=DEV)
    m.load_weights(P.items())
    return m, P


def test_fp8_forward_stays_close_to_the_bf16_forward_and_the_fp32_oracle():
    heads, joint, layers, grid, T = 2, 256, 3, (1, 16, 16), 19
    m, P = _model(layers, heads, joint)
    g = torch.Generator().manual_seed(4)
    lat = bf16_round(torch.randn(2, 256, 64, generator=g))
    txt = bf16_round(torch.randn(2, T, joint, generator=g))
    sig = torch.tensor([0.6015625, 0.6015625])
    kw = dict(hidden_states=lat.to(DEV, BF16), encoder_hidden_states=txt.to(DEV, BF16), timestep=sig.to(DEV),
              img_shapes=[[grid]] * 2, txt_seq_lens=[T] * 2, return_dict=False)
    out16 = m(**kw)[0].float().cpu()
    m.enable_fp8()
    out8 = m(**kw)[0].float().cpu()
    out8b = m(**kw)[0].float().cpu()
    m.enable_fp8(False)
    again16 = m(**kw)[0].float().cpu()
    torch.cuda.synchronize()
    oracle = O.dit_forward({k: bf16_round(v) for k, v in P.items()}, lat, txt, sig, grid, num_heads=heads)
    e8, e16, d = rel_l2(out8, oracle), rel_l2(out16, oracle), rel_l2(out8, out16)
    print(f"{layers}-layer forward vs fp32 oracle: bf16 path {e16:.3e}, fp8 path {e8:.3e}; fp8 vs bf16 {d:.3e}")
    assert torch.equal(out8, out8b) and torch.equal(out16, again16)          # deterministic; switching back restores bf16 bit for bit
    assert torch.isfinite(out8).all()
    assert e16 <= 1e-2
    assert e8 <= 6e-2 and d <= 6e-2       # e4m3 operands (3 mantissa bits, per-token / per-channel scales) through 4 GEMMs x 3 layers


def test_fp8_denoise_loop_runs_and_tracks_the_bf16_loop():
    import _gpu_factory
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    pipe = _gpu_factory.make_small_pipeline()            # 2 layers, joint 128 -> D = 256: K % 128 == 0 for every block GEMM
    g = torch.Generator().manual_seed(12)
    req = OmniDiffusionRequest(height=128, width=128, num_inference_steps=4, true_cfg_scale=4.0, output_type="latent",
                               latents=torch.randn(1, 64, 64, generator=g).to(BF16),
                               prompt_embeds=torch.randn(1, 9, 128, generator=g).to(BF16),
                               negative_prompt_embeds=torch.randn(1, 5, 128, generator=g).to(BF16))
    ref = pipe.generate([req], output_type="latent")[0].output
    pipe.transformer.enable_fp8()
    pipe._step_state.clear()
    out = pipe.generate([req], output_type="latent")[0].output
    pipe.transformer.enable_fp8(False)
    d = rel_l2(out, ref.float().cpu())
    print(f"4-step CFG loop: fp8 vs bf16 final latent rel_l2 {d:.3e}")
    assert torch.isfinite(out.float()).all() and d <= 8e-2


def test_fp8_at_the_benchmarked_width_and_depth():
    """Round-3 verdict: the fp8 throughput figures had no accuracy gate at the size they are quoted for (D = 3072, 60 layers).
    ONE 1024^2 item (4096 + 64 rows), 60 full-width layers, random N(0, 0.02^2) weights: one forward and the 4-step true-CFG
    loop, in bf16, in the ACCURATE fp8 recipe (attention-side projections) and in all-fp8, each against the fp32 oracle on the
    GPU and against the bf16 product path.

    What the bounds are (measured with tools/fp8_error_budget.py, profiles/r04_fp8_error_budget.log; e4m3 has 3 mantissa bits:
    an fp8 GEMM is 3.1-3.7e-2 from the unquantised product whatever the scale granularity, and through 60 random-weight layers
    the four GEMM classes add 2.1e-2 / 2.6e-2 / 5.3e-2 / 5.2e-2 (qkv / out / MLP-up / MLP-down) in quadrature on bf16's own 1.6e-2):
      * accurate recipe: the final latent of the loop no further from fp32 than 2x the bf16 path (measured 1.57x; 1.08x with the
        qkv class alone), one forward within 2.5x (quadrature of the class contributions: 2.3x);
      * all-fp8: forward <= 0.11 (measured 8.6e-2), loop <= 0.22 (measured 0.171 = 3.9x bf16) - a regression gate on a number
        that is PRINTED next to every fp8 throughput figure (bench.py secondary.fp8_*), not a claim of parity."""
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    torch.backends.cuda.matmul.allow_tf32 = False
    m = QwenImageTransformer2DModel(num_layers=60, device=DEV)
    m.init_random_(seed=1234)
    g = torch.Generator(device=DEV).manual_seed(1235)
    for n, p in m.named_parameters():
        if p.dim() == 1 and "norm" in n:
            p.data.add_(0.1 * torch.randn(p.shape, device=DEV, generator=g).to(BF16))
        elif p.dim() == 1:
            p.data.copy_((0.02 * torch.randn(p.shape, device=DEV, generator=g)).to(BF16))
    grid, S, steps = (1, 64, 64), 4096, 4
    g = torch.Generator(device=DEV).manual_seed(42)
    lat = torch.randn(1, S, 64, device=DEV, generator=g).to(BF16)
    pos = torch.randn(1, 64, 3584, device=DEV, generator=g).to(BF16)
    neg = torch.randn(1, 48, 3584, device=DEV, generator=g).to(BF16)
    sig = torch.tensor([0.6015625], device=DEV)
    P32 = {n: p.detach().float() for n, p in m.named_parameters()}
    with torch.no_grad():
        ref_f = O.dit_forward(P32, lat.float(), pos.float(), sig, grid, num_heads=24)
        ts, sg = O.flow_match_sigmas(steps, S)
        x = lat.float()
        for i, t in enumerate(ts):
            s_in = (t.bfloat16() / 1000).bfloat16().float().expand(1).to(DEV)
            p = O.dit_forward(P32, x, pos.float(), s_in, grid, num_heads=24)
            n = O.dit_forward(P32, x, neg.float(), s_in, grid, num_heads=24)
            x = O.euler_step(x, O.cfg_combine(p, n, 4.0), float(sg[i]), float(sg[i + 1])).bfloat16().float()
        ref_l = x
    del P32
    torch.cuda.empty_cache()
    pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(use_hip_graph=False), device=DEV, transformer=m)
    req = OmniDiffusionRequest(height=1024, width=1024, num_inference_steps=steps, true_cfg_scale=4.0, latents=lat,
                               prompt_embeds=pos, negative_prompt_embeds=neg, output_type="latent")
    kw = dict(hidden_states=lat, encoder_hidden_states=pos, timestep=sig, img_shapes=[[grid]], txt_seq_lens=[64], return_dict=False)
    res = {}
    for name, cls in (("bf16", None), ("accurate", m.FP8_RECIPE_ACCURATE), ("all", m.FP8_CLASSES)):
        m.enable_fp8(cls) if cls else m.enable_fp8(False)
        f = m(**kw)[0]
        lo = pipe.generate([req], output_type="latent")[0].output
        torch.cuda.synchronize()
        res[name] = (rel_l2(f, ref_f), rel_l2(lo, ref_l), f.clone(), lo.clone())
        print(f"   {name:9s} forward vs fp32 {res[name][0]:.3e}, 4-step loop vs fp32 {res[name][1]:.3e}"
              + ("" if name == "bf16" else f"; vs bf16 path: forward {rel_l2(f, res['bf16'][2]):.3e}, loop {rel_l2(lo, res['bf16'][3]):.3e}"))
    m.enable_fp8(False)
    for name in res:
        assert torch.isfinite(res[name][2].float()).all() and torch.isfinite(res[name][3].float()).all()
    assert res["accurate"][0] <= 2.5 * res["bf16"][0] and res["accurate"][1] <= 2.0 * res["bf16"][1]
    assert res["all"][0] <= 0.11 and res["all"][1] <= 0.22
    assert res["all"][0] >= res["accurate"][0] >= 0.9 * res["bf16"][0]       # the ordering the error budget predicts
