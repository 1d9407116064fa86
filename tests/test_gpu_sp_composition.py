"""GPU (one device): Ulysses sequence parallelism COMPOSED with the rest of the path — TeaCache, the Edit pipelines' condition
images, the Layered variant (round-4 verdict: these three raised NotImplementedError under SP, while the reference's strategy
wraps any DiT forward: attention/parallel/ulysses.py:59-135, qwen_image_transformer.py:735-742,776-801).

The boxes have one MI355X, so the P ranks are VIRTUAL: `pipe._sp_emulate_ranks = P` makes the sequence-parallel loop build the
forward generators of all P ranks (each with its own row slice, head slice and TeaCache state) and perform their collectives in
process (distributed/sp_driver.drive_in_process: all-to-all recv[r][j] = send[j][r], all-gather = stack, all-reduce = sum) —
every kernel, reshard and index permutation of the real path runs, only the wire is replaced, and every virtual rank must end
with the same prediction (checked inside the driver).  The same generators run over gloo in tests/test_sequence_parallel_host.py
and over RCCL in tests/test_gpu_sequence_parallel.py wherever two devices are visible.

Contract (the reference's own SP contract, tests/e2e/offline_inference/test_sequence_parallel.py:128-147, and
tests/diffusion/attention/test_ulysses_sequence_parallel.py:332-343: max_rel < 1e-2): SP == non-SP within tolerance (here final
latent rel_l2 <= 1e-2 after a true-CFG loop — one forward differs by <= 4e-3: same kernels, but attention sees another head /
tile grouping and the row-sharded GEMMs other tile shapes; CFG scale 4 amplifies that over the steps), plus, for the pinned pipelines, the oracle /
reference-run fixture at the tolerance of their single-device tests."""
import json
import os

import numpy as np
import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEADS, JOINT, LAYERS = 4, 128, 2


def _dit(seed=1234, **kw):
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    P = O.make_dit_params(LAYERS, seed=seed, bias_std=0.02, norm_jitter=0.1, num_heads=HEADS, joint_dim=JOINT)
    m = QwenImageTransformer2DModel(num_layers=LAYERS, num_attention_heads=HEADS, joint_attention_dim=JOINT, device=DEV, **kw)
    m.load_weights(P.items())
    return m, P


def _with_virtual_ranks(pipe, P, fn):
    pipe._sp_emulate_ranks = P
    try:
        return fn()
    finally:
        pipe._sp_emulate_ranks = 0


@pytest.mark.parametrize("P", [2, 4])
def test_text_to_image_loop_over_virtual_ranks_equals_the_plain_loop(P):
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    m, _ = _dit()
    pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(max_step_batch=4), device=DEV, transformer=m)
    g = torch.Generator().manual_seed(21)
    reqs = [OmniDiffusionRequest(height=128, width=128, num_inference_steps=4, true_cfg_scale=4.0, output_type="latent",
                                 latents=torch.randn(1, 64, 64, generator=g).to(BF16),
                                 prompt_embeds=torch.randn(1, T, JOINT, generator=g).to(BF16),
                                 negative_prompt_embeds=torch.randn(1, Tn, JOINT, generator=g).to(BF16))
            for T, Tn in ((7, 3), (19, 12))]
    plain = [o.output for o in pipe.generate(reqs, output_type="latent")]
    sp = _with_virtual_ranks(pipe, P, lambda: [o.output for o in pipe.generate(reqs, output_type="latent")])
    torch.cuda.synchronize()
    for a, b in zip(plain, sp):
        e = rel_l2(b, a)
        print(f"P={P}: SP loop vs ragged loop rel_l2 {e:.3e}")
        assert e <= 1e-2


@pytest.mark.parametrize("P", [2, 4])
def test_teacache_under_sequence_parallelism_takes_the_single_device_decisions(P):
    """Per-rank residual slices + ONE all-reduced pair of sums per forward: every virtual rank decides what the device-side
    single-GPU TeaCache decides (never-skip == the uncached SP loop bit for bit, always-skip: only the first forward of each
    branch computes, an intermediate threshold: the same skip counts per CFG branch), and the latents follow."""
    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    steps = 8
    m, _ = _dit()
    pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(use_hip_graph=False), device=DEV, transformer=m)
    g = torch.Generator().manual_seed(3)
    req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=steps, true_cfg_scale=4.0, output_type="latent",
                               latents=bf16_round(torch.randn(1, 256, 64, generator=g)).to(BF16),
                               prompt_embeds=bf16_round(torch.randn(1, 11, JOINT, generator=g)).to(BF16),
                               negative_prompt_embeds=bf16_round(torch.randn(1, 6, JOINT, generator=g)).to(BF16))

    def run(thresh, ranks):
        m.teacache = None if thresh is None else TeaCacheConfig(rel_l1_thresh=thresh)
        pipe._step_state.clear()
        try:
            fn = lambda: pipe.generate([req], output_type="latent")[0].output.clone()  # noqa: E731
            out = _with_virtual_ranks(pipe, ranks, fn) if ranks else fn()
            torch.cuda.synchronize()
            st = pipe.last_teacache_state
            return out, (None if thresh is None else st.skipped_forwards()), st
        finally:
            m.teacache = None

    sp_plain, _, _ = run(None, P)
    sp_never, skips, _ = run(1e-12, P)
    assert skips == [0, 0] and torch.equal(sp_never, sp_plain)
    dev_always, dskips, _ = run(1e12, 0)
    sp_always, skips, _ = run(1e12, P)
    assert skips == dskips == [steps - 1, steps - 1]
    assert rel_l2(sp_always, dev_always) <= 1e-2
    # an intermediate threshold between the observed rescaled distances of the all-compute run
    from vllm_omni_amd.diffusion.cache.teacache.hook import apply_teacache_hook
    from vllm_omni_amd.diffusion.hooks import HookRegistry
    from test_gpu_teacache import _host_hook_loop

    hook = apply_teacache_hook(m, TeaCacheConfig(rel_l1_thresh=1e-12))
    try:
        _host_hook_loop(m, hook, req.latents.float(), req.prompt_embeds.float(), req.negative_prompt_embeds.float(), steps)
        observed = sorted(hook.rescaled_history)
    finally:
        HookRegistry.get_or_create(m).remove_hook("teacache")
    thresh = 1.6 * observed[len(observed) // 2]
    dev_mid, dskips, _ = run(thresh, 0)
    sp_mid, skips, st = run(thresh, P)
    print(f"P={P}: skipped forwards per CFG branch at thresh {thresh:.3e}: device single-GPU {dskips}, sequence-parallel {skips}; "
          f"decisions (positive branch) {''.join('c' if d else 's' for d in st.states[0].decisions)}")
    assert 0 < sum(skips) < 2 * (steps - 1)                                  # the threshold really splits the forwards
    assert all(abs(a - b) <= 1 for a, b in zip(skips, dskips))               # (a near-tie may fall either way: fp32 summation order)
    if skips == dskips:
        e = rel_l2(sp_mid, dev_mid)
        print(f"   same pattern: SP vs single-GPU TeaCache loop rel_l2 {e:.3e}")
        assert e <= 1e-2


@pytest.mark.parametrize("P", [2, 4])
def test_edit_condition_images_under_sequence_parallelism(P):
    """[latents ; condition latents] sharded as ONE sequence (reference pipeline_qwen_image_edit.py:600-632 + the chunk at
    qwen_image_transformer.py:735-738): 64 + 24 = 88 rows over P ranks, two RoPE frames; vs the plain Edit loop and the oracle."""
    from vllm_omni_amd.diffusion.models.qwen_image.autoencoder_kl_qwenimage import AutoencoderKLQwenImage
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image_edit import QwenImageEditPipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    m, Pm = _dit()
    vae = AutoencoderKLQwenImage(device=DEV, with_encoder=True)
    Pe, Pd = O.make_vae_encoder_params(), O.make_vae_params()
    vae.load_weights(list(Pe.items()) + list(Pd.items()))
    pipe = QwenImageEditPipeline(device=DEV, transformer=m, vae=vae)
    g = torch.Generator().manual_seed(4)
    image = bf16_round(torch.rand(1, 3, 64, 96, generator=g) * 2 - 1)            # condition image -> 4 x 6 tokens
    lat = bf16_round(torch.randn(1, 64, 64, generator=g))                         # target 128 x 128 -> 8 x 8 tokens
    pos, neg = bf16_round(torch.randn(1, 9, JOINT, generator=g)), bf16_round(torch.randn(1, 5, JOINT, generator=g))
    req = OmniDiffusionRequest(height=128, width=128, num_inference_steps=3, true_cfg_scale=4.0, latents=lat.to(BF16),
                               prompt_embeds=pos.to(BF16), negative_prompt_embeds=neg.to(BF16), output_type="latent",
                               extra={"image": image})
    plain = pipe.generate([req], output_type="latent")[0].output[0]
    sp = _with_virtual_ranks(pipe, P, lambda: pipe.generate([req], output_type="latent")[0].output[0])
    torch.cuda.synchronize()
    Pb = {k: bf16_round(v) for k, v in Pm.items()}
    cond = bf16_round(O.image_to_latents({k: bf16_round(v) for k, v in Pe.items()}, image.unsqueeze(2)))
    grids = [(1, 8, 8), (1, 4, 6)]
    ts, sig = O.flow_match_sigmas(3, 64)
    x = lat.float()
    for i, t in enumerate(ts):
        s_in = (t.bfloat16() / 1000).bfloat16().float().expand(1)
        inp = torch.cat([x, cond], dim=1)
        p = O.dit_forward(Pb, inp, pos.float(), s_in, grids, num_heads=HEADS)[:, :64]
        n = O.dit_forward(Pb, inp, neg.float(), s_in, grids, num_heads=HEADS)[:, :64]
        x = bf16_round(O.euler_step(x, O.cfg_combine(p, n, 4.0), float(sig[i]), float(sig[i + 1])))
    e, r = rel_l2(sp, plain), rel_l2(sp, x[0])
    print(f"P={P}: Edit loop SP vs plain {e:.3e}; SP vs fp32 oracle {r:.3e} (plain vs oracle {rel_l2(plain, x[0]):.3e})")
    assert e <= 1e-2 and r <= 2e-2


@pytest.mark.parametrize("norm", [False, True])
def test_layered_pipeline_under_sequence_parallelism(norm):
    """Layered: 3 generated frames + the condition image = 204 rows, layer-3D RoPE frame indices, additional_t_cond rows on
    every rank, un-normalised true-CFG — two virtual ranks (the fixture's model has two heads) vs the plain loop and the
    REFERENCE-RUN fixture."""
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image_layered import QwenImageLayeredPipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    z = np.load(os.path.join(GOLDEN_DIR, "layered_dit_and_pipeline.npz"))
    meta = json.loads(str(z["meta"]))
    c = meta["case"]
    Pm = O.make_dit_params(c["layers"], seed=1234, bias_std=c["bias_std"], norm_jitter=c["jitter"], num_heads=c["heads"],
                           joint_dim=c["joint"])
    Pm["time_text_embed.addition_t_embedding.weight"] = torch.randn(
        2, c["heads"] * 128, generator=torch.Generator().manual_seed(meta["t_embed_seed"])) * 0.5
    m = QwenImageTransformer2DModel(num_layers=c["layers"], num_attention_heads=c["heads"], joint_attention_dim=c["joint"],
                                    use_additional_t_cond=True, use_layer3d_rope=True, device=DEV)
    m.load_weights(Pm.items())
    pipe = QwenImageLayeredPipeline(device=DEV, transformer=m)
    pipe.vae.init_random_(seed=3)
    gh, gw = c["gen_grid"]
    ch, cw = c["cond_grid"]
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    req = OmniDiffusionRequest(height=16 * gh, width=16 * gw, num_inference_steps=c["steps"], true_cfg_scale=c["cfg"],
                               latents=t("latents").to(BF16), prompt_embeds=t("pos").to(BF16),
                               negative_prompt_embeds=t("neg").to(BF16), output_type="latent",
                               extra={"image_latents": t("image_latents").to(BF16), "image_latent_grid": (ch, cw),
                                      "layers": c["img_layers"], "cfg_normalize": norm})
    plain = pipe.generate([req], output_type="latent")[0].output
    sp = _with_virtual_ranks(pipe, 2, lambda: pipe.generate([req], output_type="latent")[0].output)
    torch.cuda.synchronize()
    tag = "norm" if norm else "plain"
    e, r = rel_l2(sp, plain), rel_l2(sp, t(f"final_{tag}"))
    print(f"layered loop ({tag}) over 2 virtual ranks: vs plain {e:.3e}, vs the reference run {r:.3e}")
    assert e <= 1e-2 and r <= 2e-2
