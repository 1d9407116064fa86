"""GPU: the Layered variant, the buildable part — `use_layer3d_rope` + `use_additional_t_cond` in the DiT and
`QwenImageLayeredPipeline` — against tests/golden/layered_dit_and_pipeline.npz, which the reference's own
QwenImageTransformer2DModel(use_additional_t_cond=True, use_layer3d_rope=True, zero_cond_t=False) and its Layered pipeline's
diffuse() produced (oracle/gen_golden.py `layered`).  zero_cond_t raises with the citation of the reference defect."""
import json
import os

import numpy as np
import pytest
import torch

import qwen_image_oracle as O
from _util import GOLDEN_DIR, cosine, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def _fixture():
    z = np.load(os.path.join(GOLDEN_DIR, "layered_dit_and_pipeline.npz"))
    meta = json.loads(str(z["meta"]))
    c = meta["case"]
    P = O.make_dit_params(c["layers"], seed=1234, bias_std=c["bias_std"], norm_jitter=c["jitter"], num_heads=c["heads"],
                          joint_dim=c["joint"])
    P["time_text_embed.addition_t_embedding.weight"] = torch.randn(
        2, c["heads"] * 128, generator=torch.Generator().manual_seed(meta["t_embed_seed"])) * 0.5
    return z, c, P


def _model(c, P):
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    m = QwenImageTransformer2DModel(num_layers=c["layers"], num_attention_heads=c["heads"], joint_attention_dim=c["joint"],
                                    use_additional_t_cond=True, use_layer3d_rope=True, device=DEV)
    assert set(m.load_weights(P.items())) == set(P)
    return m


def test_layered_forward_matches_the_reference_run():
    z, c, P = _fixture()
    m = _model(c, P)
    gh, gw = c["gen_grid"]
    ch, cw = c["cond_grid"]
    shapes = [(1, gh, gw)] * (c["img_layers"] + 1) + [(1, ch, cw)]
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    kw = dict(hidden_states=t("fwd_in").to(DEV, BF16), encoder_hidden_states=t("fwd_txt").to(DEV, BF16),
              timestep=torch.tensor([0.62, 0.62], device=DEV), img_shapes=[shapes] * 2, txt_seq_lens=[c["T"]] * 2, return_dict=False)
    out = m(**kw, additional_t_cond=torch.tensor([0, 1]))[0]
    torch.cuda.synchronize()
    r = rel_l2(out, t("fwd_out"))
    print(f"layered forward (layer-3D RoPE, additional_t_cond [0, 1]) vs the reference run: {r:.3e}")
    assert r <= 1.5e-2 and cosine(out, t("fwd_out")) >= 0.9995
    same = m(**kw, additional_t_cond=torch.tensor([0, 0]))[0]
    assert rel_l2(same[0:1], t("fwd_out")[0:1]) <= 1.5e-2 and rel_l2(same[1:2], t("fwd_out")[1:2]) > 3e-2   # the t-cond row matters
    with pytest.raises(ValueError):
        m(**kw)                                              # reference :56-57: the flag makes the argument mandatory


def test_zero_cond_t_is_refused_with_the_reference_defect_cited():
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    with pytest.raises(NotImplementedError, match="552-564"):
        QwenImageTransformer2DModel(num_layers=1, num_attention_heads=2, joint_attention_dim=128, zero_cond_t=True, device=DEV)


@pytest.mark.parametrize("norm", [False, True])
def test_layered_pipeline_loop_matches_the_reference_run(norm):
    """QwenImageLayeredPipeline: (layers + 1) generated frames + the condition image on one sequence axis, mu from the condition
    rows, is_rgb = 0, un-normalised true-CFG by default — static loop and continuous step batcher vs the reference's diffuse()."""
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image_layered import QwenImageLayeredPipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest
    from vllm_omni_amd.diffusion.step_batcher import ContinuousStepBatcher

    z, c, P = _fixture()
    pipe = QwenImageLayeredPipeline(device=DEV, transformer=_model(c, P))
    pipe.vae.init_random_(seed=3)
    gh, gw = c["gen_grid"]
    ch, cw = c["cond_grid"]
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    tag = "norm" if norm else "plain"

    def req(out="latent"):
        return OmniDiffusionRequest(height=16 * gh, width=16 * gw, num_inference_steps=c["steps"], true_cfg_scale=c["cfg"],
                                    latents=t("latents").to(BF16), prompt_embeds=t("pos").to(BF16),
                                    negative_prompt_embeds=t("neg").to(BF16), output_type=out,
                                    extra={"image_latents": t("image_latents").to(BF16), "image_latent_grid": (ch, cw),
                                           "layers": c["img_layers"], "cfg_normalize": norm})

    sm = pipe.resolve_request(req())[0]
    sch_ts = pipe.scheduler.set_timesteps(c["steps"], sm["lat"].shape[0], sm["sigmas"], mu=sm["mu"])
    assert torch.equal(sch_ts, t("timesteps")) and torch.equal(pipe.scheduler.sigmas, t("sigmas"))        # bit-exact schedule
    out = pipe.generate([req()], output_type="latent")[0].output
    torch.cuda.synchronize()
    r = rel_l2(out, t(f"final_{tag}"))
    b = ContinuousStepBatcher(pipe, max_items=2)
    b.add(req(), "a")
    served = dict(b.drain())["a"].output
    r2 = rel_l2(served, t(f"final_{tag}"))
    print(f"layered loop ({tag}): static {r:.3e}, step batcher {r2:.3e} vs the reference run")
    assert r <= 2e-2 and r2 <= 2e-2
    other = "plain" if norm else "norm"
    assert rel_l2(out, t(f"final_{other}")) > rel_l2(out, t(f"final_{tag}"))                              # the flag is honoured
    if not norm:                                             # decode: frame 0 dropped, one image per layer
        imgs = pipe.generate([req("pt")])[0].output
        assert imgs.shape == (c["img_layers"], 3, 16 * gh, 16 * gw) and torch.isfinite(imgs.float()).all()
