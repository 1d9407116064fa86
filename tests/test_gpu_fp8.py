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
