"""GPU parity of the denoise loop and the VAE decode against the fp32 oracle.

BASELINE.json configs[0]: Qwen-Image DiT 256x256, 4 denoise steps, batch 1.  Final-latent tolerance:
rel_l2 <= 2e-2, cosine >= 0.9995 vs the fp32 oracle on the same bf16-rounded weights / identical injected latents."""
import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, cosine, golden_params, load_golden, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def make(heads, joint, layers, seed=1234):
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    P = O.make_dit_params(layers, seed=seed, bias_std=0.02, norm_jitter=0.1, num_heads=heads, joint_dim=joint)
    m = QwenImageTransformer2DModel(num_layers=layers, num_attention_heads=heads, joint_attention_dim=joint, device=DEV)
    m.load_weights(P.items())
    pipe = QwenImagePipeline(device=DEV, transformer=m)
    return pipe, {k: bf16_round(v) for k, v in P.items()}


def oracle_denoise(Pb, lat, pos, neg, grid, steps, heads, cfg=4.0):
    ts, sig = O.flow_match_sigmas(steps, lat.shape[1])
    x = lat.float()
    for i, t in enumerate(ts):
        s_in = (t.bfloat16() / 1000).bfloat16().float().expand(1)   # the reference's bf16 timestep chain
        p = O.dit_forward(Pb, x, pos.float(), s_in, grid, num_heads=heads)
        if neg is not None:
            n = O.dit_forward(Pb, x, neg.float(), s_in, grid, num_heads=heads)
            p = O.cfg_combine(p, n, cfg)
        x = bf16_round(O.euler_step(x, p, float(sig[i]), float(sig[i + 1])))   # latents are kept in bf16 (:585)
    return x


@pytest.mark.parametrize("heads,joint,layers", [(2, 128, 2), (24, 3584, 2)])
def test_config0_256px_4steps_final_latent(heads, joint, layers):
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    pipe, Pb = make(heads, joint, layers)
    g = torch.Generator().manual_seed(42)
    lat = bf16_round(torch.randn(1, 256, 64, generator=g))                  # 256x256 image -> 16x16 tokens
    pos = bf16_round(torch.randn(1, 11, joint, generator=torch.Generator().manual_seed(1)))
    neg = bf16_round(torch.randn(1, 5, joint, generator=torch.Generator().manual_seed(2)))
    req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=4, true_cfg_scale=4.0, latents=lat.to(BF16),
                               prompt_embeds=pos.to(BF16), negative_prompt_embeds=neg.to(BF16), output_type="latent")
    out = pipe.generate([req], output_type="latent")[0].output
    torch.cuda.synchronize()
    ref = oracle_denoise(Pb, lat, pos, neg, (1, 16, 16), 4, heads)
    r, c = rel_l2(out, ref), cosine(out, ref)
    print(f"config0 heads={heads}: final latent rel_l2 {r:.3e} cos {c:.6f}")
    assert r <= 2e-2 and c >= 0.9995


def test_step_batched_requests_equal_solo_runs():
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    pipe, _ = make(2, 128, 2)
    reqs = []
    for i, (t, tn) in enumerate([(7, 3), (19, 12), (4, 4)]):
        g = torch.Generator().manual_seed(10 + i)
        reqs.append(OmniDiffusionRequest(height=128, width=128, num_inference_steps=3, true_cfg_scale=4.0,
                                         latents=torch.randn(1, 64, 64, generator=g).to(BF16),
                                         prompt_embeds=torch.randn(1, t, 128, generator=g).to(BF16),
                                         negative_prompt_embeds=torch.randn(1, tn, 128, generator=g).to(BF16),
                                         output_type="latent"))
    batched = pipe.generate(reqs, output_type="latent")
    for r, b in zip(reqs, batched):
        solo = pipe.generate([r], output_type="latent")[0].output
        assert rel_l2(b.output, solo) <= 5e-3      # same per-request semantics; only GEMM tile grouping differs


def test_no_cfg_path_and_forward_entry():
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    pipe, Pb = make(2, 128, 2)
    g = torch.Generator().manual_seed(3)
    lat, pos = bf16_round(torch.randn(1, 64, 64, generator=g)), bf16_round(torch.randn(1, 9, 128, generator=g))
    req = OmniDiffusionRequest(height=128, width=128, num_inference_steps=2, latents=lat.to(BF16),
                               prompt_embeds=pos.to(BF16), output_type="latent")
    out = pipe.forward(req).output
    ref = oracle_denoise(Pb, lat, pos, None, (1, 8, 8), 2, 2)
    assert rel_l2(out, ref) <= 2e-2


def test_vae_decode_matches_oracle():
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage

    Pv = O.make_vae_params()
    vae = AutoencoderKLQwenImage(device=DEV)
    assert vae.load_weights(Pv.items()) == {k for k in Pv if "time_conv" not in k}
    z = bf16_round(torch.randn(1, 16, 1, 16, 16, generator=torch.Generator().manual_seed(9)) * 1.5)
    img = vae.decode(z.to(DEV, BF16))[0]
    torch.cuda.synchronize()
    ref = O.vae_decode({k: bf16_round(v) for k, v in Pv.items()}, z)
    assert img.shape == ref.shape == (1, 3, 1, 128, 128)
    r = rel_l2(img, ref)
    print(f"vae decode rel_l2 {r:.3e}  max_abs {(img.float().cpu() - ref).abs().max():.3e}")
    assert r <= 3e-2 and (img.float().cpu() - ref).abs().mean() <= 2e-2     # the reference's own pixel bar: mean <= 2e-2


@pytest.mark.parametrize("hw", [(16, 16), (40, 24)])
def test_vae_decode_of_a_batch_equals_the_images_decoded_one_by_one(hw):
    """The pipeline decodes all finished images of one size in ONE VAE call (pipeline_qwen_image.py generate): every kernel of the
    decoder treats the images independently (conv tiles carry the image in blockIdx.z, attention runs per image), so at sizes where
    the batch does not change the conv launcher's tile choice the batch reproduces the single-image results bit for bit.  (At the
    production raster the choice DOES depend on the batch: next test.)"""
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage

    vae = AutoencoderKLQwenImage(device=DEV)
    vae.init_random_(seed=7)
    z = (torch.randn(3, 16, 1, *hw, generator=torch.Generator().manual_seed(10)) * 1.5).to(DEV, BF16)
    batch = vae.decode(z)[0]
    single = torch.cat([vae.decode(z[i:i + 1])[0] for i in range(3)])
    torch.cuda.synchronize()
    assert batch.shape == (3, 3, 1, 8 * hw[0], 8 * hw[1]) and torch.isfinite(batch.float()).all()
    assert torch.equal(batch, single)


def test_vae_decode_batch_vs_single_at_the_production_raster():
    """at 1024^2 the conv launcher switches the 130^2 / Cout = 384 layers to the 8-wave 192-channel tile (and the fused
    norm at Cout = 192 with it) once a call carries >= 4 images (csrc/vae.hip conv_uses_big_tile: it looks at the call's TOTAL tile
    count) — another K-tile grouping, i.e. another fp32 summation order.  So an image decoded in the bench's batch of five and the
    same image decoded alone (the serving path's finish_request) are NOT bit-identical there; what is asserted is that they agree
    like two bf16 evaluations of the same decoder (rel_l2 <= 1e-2, mean |diff| <= 5e-3 on [-1, 1] pixels — well inside the
    reference's pixel bar of 2e-2) and that the batch itself is deterministic."""
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage

    vae = AutoencoderKLQwenImage(device=DEV)
    vae.init_random_(seed=7)
    z = (torch.randn(5, 16, 1, 128, 128, generator=torch.Generator().manual_seed(11)) * 1.5).to(DEV, BF16)
    batch = vae.decode(z)[0]
    again = vae.decode(z)[0]
    single = torch.cat([vae.decode(z[i:i + 1])[0] for i in (0, 4)])
    torch.cuda.synchronize()
    assert torch.equal(batch, again)
    got = torch.stack([batch[0], batch[4]])
    r = rel_l2(got, single.float().cpu())
    d = float((got.float() - single.float()).abs().mean())
    print(f"1024^2 decode, image in a batch of five vs alone: rel_l2 {r:.3e}, mean |diff| {d:.3e}, identical: {bool(torch.equal(got, single))}")
    assert r <= 1e-2 and d <= 5e-3


def test_pipeline_decode_end_to_end_shapes():
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    pipe, _ = make(2, 128, 1)
    pipe.vae.init_random_()
    g = torch.Generator().manual_seed(3)
    req = OmniDiffusionRequest(height=128, width=128, num_inference_steps=2, seed=7,
                               prompt_embeds=torch.randn(1, 9, 128, generator=g).to(BF16))
    out = pipe.forward(req)
    assert out.error is None and out.output.shape == (1, 3, 128, 128)
    assert torch.isfinite(out.output.float()).all() and float(out.output.float().abs().max()) <= 1.0


# ---------------------------------------------------------------- against REFERENCE-RUN fixtures (not via the oracle)
@pytest.mark.parametrize("name", ["vae_decode_16x16_fp32", "vae_decode_24x40_fp32"])
def test_vae_decode_matches_reference_golden(name):
    """Product VAE vs what the reference's vendored AutoencoderKLQwenImage.decode produced in fp32
    (tests/golden, oracle/gen_golden.py; autoencoder_kl_qwenimage.py:839-887)."""
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage

    z, meta, c = load_golden(name)
    vae = AutoencoderKLQwenImage(device=DEV)
    vae.load_weights(O.make_vae_params().items())
    img = vae.decode(torch.from_numpy(z["z"]).to(DEV, BF16))[0]
    torch.cuda.synchronize()
    ref = torch.from_numpy(z["image"])
    r = rel_l2(img, ref)
    d = (img.float().cpu() - ref).abs()
    print(f"{name}: product vs reference golden rel_l2 {r:.3e} mean|err| {float(d.mean()):.3e} max {float(d.max()):.3e}")
    assert img.shape == ref.shape and r <= 3e-2 and float(d.mean()) <= 2e-2


def test_diffuse_matches_reference_golden():
    """Product denoise loop (bf16, fused CFG+Euler kernel, CFG pair as one ragged forward) vs the trajectory the
    reference's own diffuse() produced in fp32 over the reference DiT (pipe_diffuse_cfg_256: 16x16 tokens, 4 steps)."""
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    z, meta, c = load_golden("pipe_diffuse_cfg_256")
    P = golden_params(c)
    m = QwenImageTransformer2DModel(num_layers=c["layers"], num_attention_heads=c["heads"],
                                    joint_attention_dim=c["joint"], device=DEV)
    m.load_weights(P.items())
    pipe = QwenImagePipeline(device=DEV, transformer=m)
    req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=c["steps"], true_cfg_scale=c["cfg"],
                               latents=torch.from_numpy(z["latents"]).to(BF16), prompt_embeds=torch.from_numpy(z["pos"]).to(BF16),
                               negative_prompt_embeds=torch.from_numpy(z["neg"]).to(BF16), output_type="latent")
    out = pipe.generate([req], output_type="latent")[0].output
    torch.cuda.synchronize()
    ref = torch.from_numpy(z["final"])
    r, cs = rel_l2(out, ref), cosine(out, ref)
    print(f"diffuse vs reference-run golden: final latent rel_l2 {r:.3e} cos {cs:.6f}")
    assert r <= 2e-2 and cs >= 0.9995


def test_hip_graph_is_recaptured_after_the_weights_change():
    """A captured denoise-step graph bakes in the device addresses and the K32-blocked layout flag of the weights.
    `state_dict()` / `load_weights()` undo the blocked re-layout (new storages, old ones freed): the next generate with the
    same batch key must NOT replay the stale graph (round-2 advisor finding: silently wrong images + a device use-after-free)."""
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    pipe0, _ = make(2, 128, 2)
    tr = pipe0.transformer
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(1, 256, 64, generator=g).to(BF16)
    pos, neg = torch.randn(1, 9, 128, generator=g).to(BF16), torch.randn(1, 4, 128, generator=g).to(BF16)
    req = lambda: OmniDiffusionRequest(height=256, width=256, num_inference_steps=3, true_cfg_scale=4.0, latents=lat,  # noqa: E731
                                       prompt_embeds=pos, negative_prompt_embeds=neg, output_type="latent")
    eager = QwenImagePipeline(od_config=OmniDiffusionConfig(use_hip_graph=False), device=DEV, transformer=tr)
    graph = QwenImagePipeline(od_config=OmniDiffusionConfig(use_hip_graph=True), device=DEV, transformer=tr)
    want = eager.generate([req()], output_type="latent")[0].output
    a = graph.generate([req()], output_type="latent")[0].output                # captures
    gen0 = tr._native_gen
    sd = {k: v.clone() for k, v in tr.state_dict().items()}                      # un-blocks the weights: storages replaced
    assert tr._native_gen != gen0 and tr._native is None
    junk = [torch.full((1 << 20,), 7.0, device=DEV) for _ in range(8)]           # recycle the freed blocks with garbage
    b = graph.generate([req()], output_type="latent")[0].output                # must re-capture, not replay
    tr.load_state_dict(sd)
    c = graph.generate([req()], output_type="latent")[0].output
    torch.cuda.synchronize()
    del junk
    assert torch.equal(a, want) and torch.equal(b, want) and torch.equal(c, want)
