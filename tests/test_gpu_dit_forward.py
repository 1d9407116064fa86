"""GPU parity of the whole native DiT forward against (a) the golden outputs the UNMODIFIED reference produced
(tests/golden, generated through oracle/ref_shims.py) and (b) the fp32 oracle on bf16-rounded weights.

Tolerance: single forward, <= 4 layers, bf16 kernels vs fp32 reference: rel_l2(noise_pred) <= 1e-2,
cosine >= 0.9995.  For calibration, the reference itself run in bf16 on CPU differs from its own fp32 run by
~4e-3..1e-2 on these cases (tests/test_oracle_golden.py)."""
import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, cosine, golden_params, load_golden, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def build_model(case, P):
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    m = QwenImageTransformer2DModel(num_layers=case["layers"], num_attention_heads=case["heads"],
                                    joint_attention_dim=case["joint"], device="cuda:0")
    loaded = m.load_weights((k, v) for k, v in P.items())
    assert loaded == set(P.keys())
    return m


@pytest.mark.parametrize("name", ["dit_small_fp32", "dit_rect_b2_fp32", "dit_fullwidth_1layer_fp32", "dit_small_bf16"])
def test_dit_forward_matches_reference_golden(name):
    z, meta, c = load_golden(name)
    P = golden_params(c)
    m = build_model(c, P)
    lat = torch.from_numpy(z["latents"]).to("cuda:0", BF16)
    txt = torch.from_numpy(z["prompt_embeds"]).to("cuda:0", BF16)
    sig = torch.from_numpy(z["sigma"])
    gh, gw = c["grid"]
    out = m(hidden_states=lat, encoder_hidden_states=txt, encoder_hidden_states_mask=None,
            timestep=sig.to("cuda:0"), img_shapes=[[(1, gh, gw)]] * c["B"], txt_seq_lens=[c["T"]] * c["B"],
            return_dict=False)[0]
    torch.cuda.synchronize()
    ref = torch.from_numpy(z["noise_pred"])
    # inputs/weights are rounded to bf16 on our side: compare against the oracle on the SAME rounded bits (tight),
    # and against the reference's own output (looser: includes the input-rounding effect)
    Pb = {k: bf16_round(v) for k, v in P.items()}
    # the reference casts the timestep to the activation dtype (bf16) before the sinusoid (:746)
    oracle = O.dit_forward(Pb, bf16_round(torch.from_numpy(z["latents"])), bf16_round(torch.from_numpy(z["prompt_embeds"])),
                           bf16_round(sig), (1, gh, gw), num_heads=c["heads"])
    r_or, r_ref = rel_l2(out, oracle), rel_l2(out, ref)
    print(f"{name}: rel_l2 vs oracle(bf16-rounded inputs) {r_or:.3e}  vs reference golden {r_ref:.3e}  cos {cosine(out, oracle):.6f}")
    assert r_or <= 1e-2 and cosine(out, oracle) >= 0.9995
    assert r_ref <= 1.5e-2     # measured 3.4e-3 .. 1.0e-2 (round 1): includes rounding the fp32 inputs/weights to bf16


def test_ragged_batch_equals_per_request():
    """Two requests with DIFFERENT text lengths in one forward == each request alone (B=1 semantics, no padding)."""
    from vllm_omni_amd.diffusion.batch import build_ragged_batch

    z, meta, c = load_golden("dit_small_fp32")
    P = golden_params(c)
    m = build_model(c, P)
    grid = (1, 8, 8)
    g = torch.Generator().manual_seed(5)
    lens = [7, 19]
    lat = [torch.randn(64, 64, generator=g).to("cuda:0", BF16) for _ in lens]
    txt = [torch.randn(t, c["joint"], generator=g).to("cuda:0", BF16) for t in lens]
    sig = torch.tensor([0.81, 0.33], dtype=torch.float32, device="cuda:0")
    both = m.forward_ragged(m.prepare_batch(build_ragged_batch(lens, grid)), torch.cat(lat), torch.cat(txt), sig)
    for i, t in enumerate(lens):
        solo = m.forward_ragged(m.prepare_batch(build_ragged_batch([t], grid)), lat[i], txt[i], sig[i:i + 1].contiguous())
        torch.cuda.synchronize()
        assert rel_l2(both[i * 64:(i + 1) * 64], solo) <= 2e-3
    # CFG pair sharing one temb row: same as two separate temb rows with equal sigma
    sh = m.forward_ragged(m.prepare_batch(build_ragged_batch(lens, grid, temb_rows=[0, 0])), torch.cat(lat), torch.cat(txt),
                          sig[:1].contiguous())
    sp = m.forward_ragged(m.prepare_batch(build_ragged_batch(lens, grid)), torch.cat(lat), torch.cat(txt),
                          sig[:1].repeat(2).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(sh, sp)


def test_modulation_table_equals_the_per_forward_gemvs():
    """omni_dit_modulation_table: every block's modulation vectors for all the steps of a schedule in one pass over the
    modulation weights.  Rows vs the per-forward GEMV (fp32 SiLU inside; the table rounds SiLU's output to bf16 like the
    reference's eager nn.Sequential(SiLU, Linear)), a forward fed from the table vs one that streams the weights, and the whole
    denoise loop with the switch on / off."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.batch import build_ragged_batch
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    heads, joint, layers = 2, 128, 3
    P = O.make_dit_params(layers, seed=1234, bias_std=0.02, norm_jitter=0.1, num_heads=heads, joint_dim=joint)
    m = QwenImageTransformer2DModel(num_layers=layers, num_attention_heads=heads, joint_attention_dim=joint, device=DEV)
    m.load_weights(P.items())
    D = heads * 128
    sig = torch.tensor([0.91, 0.6015625, 0.33, 0.12, 0.02], device=DEV)
    tab = m.modulation_table(sig)
    assert tab.shape == (layers, 2, 5, 6 * D)
    # up to 8 rows the table is filled by the GEMV kernel itself, beyond that by the GEMM kernel (one pass over the weights)
    sig12 = torch.linspace(0.97, 0.03, 12, device=DEV)
    for sg, tb in ((sig, tab), (sig12, m.modulation_table(sig12))):
        temb = m.time_text_embed(sg, None)
        for l in (0, layers - 1):
            blk = m.transformer_blocks[l]
            for s, mod in ((0, blk.img_mod), (1, blk.txt_mod)):
                ref = torch.cat([ops.linear_smallbatch(temb[i:i + 8], mod[1].weight, mod[1].bias, act_in=1) for i in range(0, len(sg), 8)])
                assert rel_l2(tb[l, s], ref) <= 3e-3, (l, s, len(sg))
    g = torch.Generator().manual_seed(3)
    T = [7, 12]
    lat = torch.randn(2 * 64, 64, generator=g).to(DEV, BF16)
    txt = torch.randn(sum(T), joint, generator=g).to(DEV, BF16)
    prepared = m.prepare_batch(build_ragged_batch(T, (1, 8, 8), temb_rows=[0, 1]))
    two = sig[1:3].contiguous()
    plain = m.forward_ragged(prepared, lat, txt, two).clone()
    fed = m.forward_ragged(prepared, lat, txt, two, mod_table=tab[:, :, 1:3].contiguous())
    torch.cuda.synchronize()
    assert rel_l2(fed, plain) <= 4e-3
    with pytest.raises(ValueError):
        m.forward_ragged(prepared, lat, txt, two, mod_table=tab)                  # rows of ANOTHER number of conditioning rows
    # whole loop, both switches, eager and as a replayed hipGraph
    req = OmniDiffusionRequest(height=128, width=128, num_inference_steps=10, true_cfg_scale=4.0, output_type="latent",
                               latents=torch.randn(1, 64, 64, generator=g).to(BF16),
                               prompt_embeds=torch.randn(1, 9, joint, generator=g).to(BF16),
                               negative_prompt_embeds=torch.randn(1, 5, joint, generator=g).to(BF16))
    outs = {}
    for pre in (False, True):
        for graph in (False, True):
            pipe = QwenImagePipeline(od_config=OmniDiffusionConfig(precompute_modulation=pre, use_hip_graph=graph), device=DEV, transformer=m)
            outs[pre, graph] = pipe.generate([req], output_type="latent")[0].output.float().cpu()
            again = pipe.generate([req], output_type="latent")[0].output.float().cpu()      # second generation: refreshed table
            assert torch.equal(again, outs[pre, graph])
    assert torch.equal(outs[True, False], outs[True, True]) and torch.equal(outs[False, False], outs[False, True])
    assert rel_l2(outs[True, False], outs[False, False]) <= 2e-2      # 10 steps of true-CFG 4.0: SiLU rounded to bf16 in the table's GEMM path


def test_modulation_table_is_kept_per_schedule_and_dropped_with_the_weights():
    """od_config.cache_modulation_tables (QwenImageTransformer2DModel.modulation_table_for_schedule): the table is a function of the
    weights and the schedule's sigmas only, so a later request with the same (resolution, step count) reuses it — no second pass
    over the modulation weights — and gets the same bits as with the cache off; another step count is another entry; new weights
    drop every entry; the byte cap evicts the oldest."""
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    heads, joint, layers = 2, 128, 2
    P = O.make_dit_params(layers, seed=77, bias_std=0.02, norm_jitter=0.1, num_heads=heads, joint_dim=joint)
    m = QwenImageTransformer2DModel(num_layers=layers, num_attention_heads=heads, joint_attention_dim=joint, device=DEV)
    m.load_weights(P.items())
    calls = []
    plain = m.modulation_table
    m.modulation_table = lambda *a, **k: (calls.append(1), plain(*a, **k))[1]

    def req(steps, seed):
        g = torch.Generator().manual_seed(seed)
        return OmniDiffusionRequest(height=128, width=128, num_inference_steps=steps, true_cfg_scale=4.0, output_type="latent",
                                    latents=torch.randn(1, 64, 64, generator=g).to(BF16),
                                    prompt_embeds=torch.randn(1, 9, joint, generator=g).to(BF16),
                                    negative_prompt_embeds=torch.randn(1, 5, joint, generator=g).to(BF16))

    def run(pipe, r):
        return pipe.generate([r], output_type="latent")[0].output.float().cpu()

    cold = QwenImagePipeline(od_config=OmniDiffusionConfig(cache_modulation_tables=False), device=DEV, transformer=m)
    warm = QwenImagePipeline(od_config=OmniDiffusionConfig(), device=DEV, transformer=m)
    ref_a, ref_b = run(cold, req(6, 1)), run(cold, req(6, 2))
    assert len(calls) == 2 and not m._mod_tables                     # cache off: one pass per generation, nothing kept
    got_a = run(warm, req(6, 1))
    assert len(calls) == 3 and len(m._mod_tables) == 1
    got_b = run(warm, req(6, 2))                                       # another prompt / seed, the same schedule: a hit
    assert len(calls) == 3
    assert torch.equal(got_a, ref_a) and torch.equal(got_b, ref_b)
    run(warm, req(4, 1))                                               # another step count: another schedule
    assert len(calls) == 4 and len(m._mod_tables) == 2
    assert torch.equal(run(warm, req(6, 1)), ref_a) and len(calls) == 4
    # the byte cap: room for one table only -> the older entry goes
    one_table = next(iter(m._mod_tables.values())).numel() * 2
    m.MOD_TABLE_CACHE_BYTES = one_table
    run(warm, req(4, 1))
    assert len(m._mod_tables) == 1 and len(calls) == 4
    del m.MOD_TABLE_CACHE_BYTES
    # new weights: the generation moves, every entry is stale
    P2 = O.make_dit_params(layers, seed=78, bias_std=0.02, norm_jitter=0.1, num_heads=heads, joint_dim=joint)
    m.load_weights(P2.items())
    fresh = run(warm, req(4, 1))
    assert len(calls) == 5 and len(m._mod_tables) == 1
    assert torch.equal(fresh, run(cold, req(4, 1)))
