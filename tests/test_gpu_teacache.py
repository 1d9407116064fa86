"""GPU: the module-level plug-in surface that cache hooks walk, and TeaCache on it.

  * the extractor contract (cache/teacache/extractors.py:145-261: img_in, txt_norm, txt_in, time_text_embed, pos_embed,
    blocks[0].img_mod / img_norm1, block(...) -> (enc, hid), norm_out, proj_out) against the fp32 oracle's taps, and against
    the one-call native forward (same kernels -> same bits);
  * the device-side TeaCache of omni_dit_forward (no host sync) against the host-driven hook (the reference's algorithm,
    hook.py:82-217) on the same model: never-skip == uncached bit-exactly, always-skip == the hook's result, and equal
    skip counts per CFG branch at an intermediate threshold."""
import numpy as np
import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"
HEADS, JOINT, LAYERS = 2, 128, 3


def _model(seed=1234):
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    P = O.make_dit_params(LAYERS, seed=seed, bias_std=0.02, norm_jitter=0.1, num_heads=HEADS, joint_dim=JOINT)
    m = QwenImageTransformer2DModel(num_layers=LAYERS, num_attention_heads=HEADS, joint_attention_dim=JOINT, device=DEV)
    m.load_weights(P.items())
    return m, {k: bf16_round(v) for k, v in P.items()}


def test_extractor_contract_against_oracle_and_native_forward():
    from vllm_omni_amd.diffusion.cache.teacache.extractors import extract_qwen_context

    m, Pb = _model()
    B, grid, T = 2, (1, 8, 8), 9
    g = torch.Generator().manual_seed(0)
    lat = bf16_round(torch.randn(B, 64, 64, generator=g))
    txt = bf16_round(torch.randn(B, T, JOINT, generator=g))
    sig = torch.tensor([0.75, 0.3125])
    kw = dict(hidden_states=lat.to(DEV, BF16), encoder_hidden_states=txt.to(DEV, BF16), encoder_hidden_states_mask=None,
              timestep=sig.to(DEV), img_shapes=[[grid]] * B, txt_seq_lens=[T] * B)
    ctx = extract_qwen_context(m, **kw, return_dict=False)
    ctx.validate()
    taps = {}
    ref = O.dit_forward(Pb, lat, txt, sig, grid, num_heads=HEADS, taps=taps)
    assert rel_l2(ctx.hidden_states, taps["hidden_in"]) <= 4e-3 and rel_l2(ctx.encoder_hidden_states, taps["enc_in"]) <= 4e-3
    assert rel_l2(ctx.temb, taps["temb"]) <= 6e-3
    assert rel_l2(ctx.modulated_input, taps["block0"]["img_n1"]) <= 6e-3          # the TeaCache decision signal
    h, e = ctx.run_transformer_blocks()
    assert rel_l2(h, taps[f"block{LAYERS - 1}"]["hidden"]) <= 1e-2 and rel_l2(e, taps[f"block{LAYERS - 1}"]["enc"]) <= 1e-2
    walked = ctx.postprocess(h)[0]
    native = m(**kw, return_dict=False)[0]
    torch.cuda.synchronize()
    assert rel_l2(walked, ref) <= 1e-2
    assert torch.equal(walked, native)            # module walk and the one-call runner launch the same kernels


def _loop_inputs(steps, seed=3):
    g = torch.Generator().manual_seed(seed)
    lat = bf16_round(torch.randn(1, 256, 64, generator=g))
    pos = bf16_round(torch.randn(1, 11, JOINT, generator=g))
    neg = bf16_round(torch.randn(1, 6, JOINT, generator=g))
    return lat, pos, neg


def _host_hook_loop(m, hook, lat, pos, neg, steps, cfg=4.0):
    """The reference's diffuse() over the HOOKED transformer.forward (two forwards per step: positive, negative)."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.diffusion.models.qwen_image.scheduling_flow_match import FlowMatchEulerSchedule

    sch = FlowMatchEulerSchedule()
    ts = sch.set_timesteps(steps, 256)
    sig_in, dt = sch.model_timestep(ts).to(DEV), sch.dt().to(DEV)
    x = lat.to(DEV, BF16).reshape(256, 64).clone()
    m.do_true_cfg = True
    hook.reset_state(m)
    for i in range(steps):
        kw = dict(encoder_hidden_states_mask=None, timestep=sig_in[i:i + 1], img_shapes=[[(1, 16, 16)]], return_dict=False)
        p = m(hidden_states=x.view(1, 256, 64), encoder_hidden_states=pos.to(DEV, BF16), txt_seq_lens=[11], **kw)[0]
        n = m(hidden_states=x.view(1, 256, 64), encoder_hidden_states=neg.to(DEV, BF16), txt_seq_lens=[6], **kw)[0]
        ops.cfg_euler_step_(x, p.reshape(256, 64).contiguous(), n.reshape(256, 64).contiguous(), cfg, dt[i:i + 1])
    return x.view(1, 256, 64), list(hook.decisions)


def _native_loop(m, thresh, lat, pos, neg, steps, graph):
    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    m.teacache = None if thresh is None else TeaCacheConfig(rel_l1_thresh=thresh)
    pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(use_hip_graph=graph), device=DEV, transformer=m)
    req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=steps, true_cfg_scale=4.0, latents=lat.to(BF16),
                               prompt_embeds=pos.to(BF16), negative_prompt_embeds=neg.to(BF16), output_type="latent")
    out = pipe.generate([req], output_type="latent")[0].output
    torch.cuda.synchronize()
    skips = pipe.last_teacache_state.skipped_forwards() if thresh is not None else None
    m.teacache = None
    return out, skips


@pytest.mark.parametrize("graph", [False, True])
def test_device_teacache_matches_host_hook(graph):
    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.cache.teacache.hook import apply_teacache_hook
    from vllm_omni_amd.diffusion.hooks import HookRegistry

    steps = 8
    m, _ = _model()
    lat, pos, neg = _loop_inputs(steps)
    plain, _ = _native_loop(m, None, lat, pos, neg, steps, graph)
    # (a) a threshold nothing stays under: every forward computes -> identical to the uncached loop, bit for bit
    never, skips = _native_loop(m, 1e-12, lat, pos, neg, steps, graph)
    assert skips == [0, 0] and torch.equal(never, plain)
    # (b) a threshold everything stays under: only the first forward of each branch computes
    always, skips = _native_loop(m, 1e12, lat, pos, neg, steps, graph)
    assert skips == [steps - 1, steps - 1]
    hook = apply_teacache_hook(m, TeaCacheConfig(rel_l1_thresh=1e12))
    try:
        host_always, dec = _host_hook_loop(m, hook, lat, pos, neg, steps)
        assert dec == [True, True] + [False] * (2 * steps - 2)
        assert rel_l2(always, host_always) <= 2e-3
        # (c) an intermediate threshold placed between the observed rescaled distances: same skip counts per branch
        hook.config.rel_l1_thresh = 1e-12
        _host_hook_loop(m, hook, lat, pos, neg, steps)                     # all-compute run to observe the distances
        observed = sorted(hook.rescaled_history)
    finally:
        HookRegistry.get_or_create(m).remove_hook("teacache")
    thresh = 1.6 * observed[len(observed) // 2]                           # ~every other forward stays under it
    hook = apply_teacache_hook(m, TeaCacheConfig(rel_l1_thresh=thresh))
    try:
        host_mid, dec = _host_hook_loop(m, hook, lat, pos, neg, steps)
    finally:
        HookRegistry.get_or_create(m).remove_hook("teacache")
    mid, skips = _native_loop(m, thresh, lat, pos, neg, steps, graph)
    host_skips = [sum(1 for c in dec[0::2] if not c), sum(1 for c in dec[1::2] if not c)]
    print(f"graph={graph} thresh={thresh:.4f}: device skips {skips}, host-hook skips {host_skips} of {steps} forwards per branch; "
          f"final latent device vs host {rel_l2(mid, host_mid):.3e}, vs uncached {rel_l2(mid, plain):.3e}")
    assert 0 < skips[0] < steps - 1 or 0 < host_skips[0] < steps - 1, "threshold did not produce a mixed pattern"
    assert abs(skips[0] - host_skips[0]) <= 1 and abs(skips[1] - host_skips[1]) <= 1
    if skips == host_skips:
        assert rel_l2(mid, host_mid) <= 5e-3


def test_step_batched_teacache_keeps_per_request_decisions():
    """Two requests with different prompts in one step-batch: each item decides for itself (B=1 semantics) — the batched
    run equals the two solo runs."""
    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    m, _ = _model()
    m.teacache = TeaCacheConfig(rel_l1_thresh=0.6)
    pipe = QwenImagePipeline(device=DEV, transformer=m)
    reqs = []
    for i in range(2):
        lat, pos, neg = _loop_inputs(6, seed=10 + i)
        reqs.append(OmniDiffusionRequest(height=256, width=256, num_inference_steps=6, true_cfg_scale=4.0, latents=lat.to(BF16),
                                         prompt_embeds=pos.to(BF16), negative_prompt_embeds=neg.to(BF16), output_type="latent"))
    both = pipe.generate(reqs, output_type="latent")
    skips_b = pipe.last_teacache_state.skipped_forwards()              # items: pos0, pos1, neg0, neg1
    for i, r in enumerate(reqs):
        solo = pipe.generate([r], output_type="latent")[0].output
        s = pipe.last_teacache_state.skipped_forwards()                # items: pos, neg
        assert s == [skips_b[i], skips_b[2 + i]]
        assert rel_l2(both[i].output, solo) <= 5e-3
    m.teacache = None


def test_device_teacache_reproduces_the_reference_run_pattern():
    """tests/golden/teacache_diffuse_cfg_256.npz = the reference's OWN TeaCacheHook + extract_qwen_context over the reference
    DiT, driven by the reference diffuse loop (fp32, 10 steps, true-CFG, rel_l1_thresh 0.15: a mixed compute / skip pattern).
    The device-side TeaCache inside omni_dit_forward must take the SAME decision at every forward of both branches, and land
    on the same final latent within the 4-step bf16 tolerance."""
    from _util import golden_params, load_golden
    from vllm_omni_amd import ops
    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    z, meta, c = load_golden("teacache_diffuse_cfg_256")
    P = golden_params(c)
    m = QwenImageTransformer2DModel(num_layers=c["layers"], num_attention_heads=c["heads"], joint_attention_dim=c["joint"], device=DEV)
    m.load_weights(P.items())
    m.teacache = TeaCacheConfig(rel_l1_thresh=c["rel_l1_thresh"])
    assert list(m.teacache.coefficients) == list(meta["coefficients"])
    pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(use_hip_graph=False), device=DEV, transformer=m)
    gh, gw = c["grid"]
    req = OmniDiffusionRequest(height=16 * gh, width=16 * gw, num_inference_steps=c["steps"], true_cfg_scale=c["cfg"],
                               latents=torch.from_numpy(z["latents"]).to(BF16), prompt_embeds=torch.from_numpy(z["pos"]).to(BF16),
                               negative_prompt_embeds=torch.from_numpy(z["neg"]).to(BF16), output_type="latent")
    pattern, orig = [], ops.cfg_euler_step_

    def tap(*a, **k):                                   # test-only host read of this forward's per-item decisions
        pattern.append(pipe.last_teacache_state.skip.tolist())
        return orig(*a, **k)

    ops.cfg_euler_step_ = tap
    try:
        out = pipe.generate([req], output_type="latent")[0].output
    finally:
        ops.cfg_euler_step_ = orig
        m.teacache = None
    torch.cuda.synchronize()
    got_pos, got_neg = [not bool(s[0]) for s in pattern], [not bool(s[1]) for s in pattern]
    show = lambda d: "".join("C" if x else "s" for x in d)  # noqa: E731
    print(f"device pattern pos {show(got_pos)} neg {show(got_neg)}; reference run pos {show(z['compute_pos'])} neg {show(z['compute_neg'])}; "
          f"final latent vs reference run {rel_l2(out, torch.from_numpy(z['final'])):.3e}")
    assert got_pos == z["compute_pos"].tolist() and got_neg == z["compute_neg"].tolist()
    assert rel_l2(out, torch.from_numpy(z["final"])) <= 1e-2
