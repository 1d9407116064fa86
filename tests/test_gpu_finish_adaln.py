"""GPU: the split-K finish folded into the AdaLN behind it (ABI v13; csrc/elementwise.hip `splitk_finish_adaln_kernel`,
`omni_gemm_bf16` with OMNI_GEMM_KERNEL_SPLITK_DEFER_FINISH, `omni_gemm_splitk_factor`).

A forward over one or two small images runs its out-projection and MLP down-projection K-split; the three launches that follow
the main kernel — `gemm_splitk_finish_kernel` (partials -> bias -> gate -> residual), then the AdaLN of the image and text
stream — become ONE pass in which a wave sums a row's partials, writes the new residual row and normalises it.  The arithmetic
is the finish kernel's followed by `rownorm_kernel`'s, operation for operation, so the contract is BIT EQUALITY with the
three-kernel sequence: of the residual stream written in place and of the AdaLN output (row-major and K32-blocked), for every
split factor split-K produces (2, 3, 4, 6, 8), widths whose chunks are partly masked, ragged row counts, with and without bias.
The forward takes this path for both GEMMs (the MLP-down finish moves into the NEXT block's norm1) whenever the modulation
vectors come from the request's table: that is checked against the same loop with a never-skipping TeaCache attached, which
defers nothing.
"""
import pytest
import torch

from _util import bf16_round

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16_round(torch.randn(shape, generator=g) * scale)


def g_(t):
    return t.to(DEV, BF16).contiguous()


def expected_factor(tiles: int, K: int, cus: int = 256) -> int:
    nkt = K // 64
    for s in (8, 6, 4, 3, 2):
        if nkt % s == 0 and nkt // s >= 4 and tiles * s <= cus:
            return s
    return 1


# (Mi, Mt, D, K, expected split factor): one 256^2 CFG pair, its MLP-down, narrow widths (partly masked chunks: D = 768 is 1.5
# chunks of 512), and taller batches that split 4 / 3 / 2 ways
CASES = [(512, 128, 3072, 3072, 6), (512, 128, 3072, 12288, 6), (300, 70, 1024, 3072, 8), (512, 128, 768, 3072, 8),
         (1024, 128, 3072, 3072, 4), (1536, 100, 3072, 3072, 3), (2304, 128, 3072, 3072, 2), (400, 75, 2048, 1024, 4)]


@pytest.mark.parametrize("Mi,Mt,D,K,want_s", CASES)
@pytest.mark.parametrize("blocked", [True, False])
def test_fused_finish_adaln_is_bit_identical_to_finish_kernel_then_adaln(Mi, Mt, D, K, want_s, blocked):
    from vllm_omni_amd import ops

    n_items = 3
    a_i, a_t = g_(rnd((Mi, K), 1)), g_(rnd((Mt, K), 2))
    w_i, w_t = g_(rnd((D, K), 3, 0.03)), g_(rnd((D, K), 4, 0.03))
    bias_i, bias_t = g_(rnd((D,), 5, 0.5)), g_(rnd((D,), 6, 0.5))
    mod = g_(rnd((n_items, 6 * D), 7, 0.5))                    # [shift1 | scale1 | gate1 | shift2 | scale2 | gate2] per item
    item_i = (torch.arange(Mi) * 7 % n_items).to(torch.int32).to(DEV)
    item_t = (torch.arange(Mt) % n_items).to(torch.int32).to(DEV)
    res_i, res_t = g_(rnd((Mi, D), 8, 3.0)), g_(rnd((Mt, D), 9, 3.0))
    tiles = (-(-Mi // 256) + -(-Mt // 256)) * (-(-D // 256))
    assert expected_factor(tiles, K) == want_s
    ws = torch.empty(8 * (Mi + Mt) * D, dtype=torch.float32, device=DEV)

    def groups(oi, ot, with_bias=True):
        return [ops.GemmGroupArgs(a_i, w_i, bias_i if with_bias else None, oi, res=oi, gate=mod[:, 2 * D:], gate_item_stride=6 * D,
                                  row_item_map=item_i),
                ops.GemmGroupArgs(a_t, w_t, bias_t if with_bias else None, ot, res=ot, gate=mod[:, 2 * D:], gate_item_stride=6 * D,
                                  row_item_map=item_t)]

    for with_bias in (True, False):
        # reference: the two-kernel split-K (main + finish kernel), then the paired AdaLN (norm2's vectors: scale2 / shift2)
        oi, ot = res_i.clone(), res_t.clone()
        assert ops.gemm_splitk_factor(groups(oi, ot, with_bias), ops.EPI_BIAS_GATE_RES, splitk_ws=ws) == want_s
        ws.fill_(float("nan"))
        ops.gemm(groups(oi, ot, with_bias), ops.EPI_BIAS_GATE_RES, splitk_ws=ws)
        yi, yt = ops.adaln_modulate_pair([(oi, mod[:, 4 * D:], mod[:, 3 * D:], item_i), (ot, mod[:, 4 * D:], mod[:, 3 * D:], item_t)],
                                         mod_item_stride=6 * D, out_k32_blocked=blocked)
        # fused: the main kernel only, then ONE pass
        hi, ht = res_i.clone(), res_t.clone()
        ws.fill_(float("nan"))
        ops.gemm(groups(hi, ht, with_bias), ops.EPI_BIAS_GATE_RES, splitk_ws=ws, kernel_hint=ops.GEMM_KERNEL_SPLITK_DEFER_FINISH)
        torch.cuda.synchronize()
        assert torch.equal(hi, res_i) and torch.equal(ht, res_t), "a deferred launch must not touch the residual stream"
        fi, ft = ops.splitk_finish_adaln_pair(
            ws, want_s, Mi + Mt,
            [(0, bias_i if with_bias else None, hi, mod[:, 2 * D:], mod[:, 4 * D:], mod[:, 3 * D:], item_i),
             (Mi, bias_t if with_bias else None, ht, mod[:, 2 * D:], mod[:, 4 * D:], mod[:, 3 * D:], item_t)],
            mod_item_stride=6 * D, out_k32_blocked=blocked)
        torch.cuda.synchronize()
        assert torch.isfinite(hi.float()).all() and torch.isfinite(fi.float()).all()
        assert torch.equal(hi, oi) and torch.equal(ht, ot), "residual stream differs from the finish kernel's"
        assert torch.equal(fi, yi) and torch.equal(ft, yt), "AdaLN output differs from omni_adaln_modulate_pair's"
        # and the numbers are the right ones (fp32 host arithmetic on the image stream)
        c = bf16_round(a_i.float().cpu() @ w_i.float().cpu().t() + (bias_i.float().cpu() if with_bias else 0.0))
        gate = mod.float().cpu()[item_i.long().cpu(), 2 * D:3 * D]
        h_ref = res_i.float().cpu() + gate * c
        assert float((hi.float().cpu() - h_ref).norm() / h_ref.norm()) <= 4e-3


def test_deferred_finish_is_refused_where_nothing_splits():
    """The hint on a launch that does not split (no workspace; a grid that fills the chip) is an error, never another path."""
    from vllm_omni_amd import _native as N
    from vllm_omni_amd import ops

    D, K = 1024, 1024
    a, w = g_(rnd((8192, K), 1)), g_(rnd((D, K), 2, 0.03))
    item = torch.zeros(8192, dtype=torch.int32, device=DEV)
    gate = g_(rnd((1, D), 3))
    out = g_(rnd((8192, D), 4))
    grp = [ops.GemmGroupArgs(a, w, None, out, res=out, gate=gate, gate_item_stride=D, row_item_map=item)]
    ws = torch.empty(8 * 8192 * D, dtype=torch.float32, device=DEV)
    assert ops.gemm_splitk_factor(grp, ops.EPI_BIAS_GATE_RES, splitk_ws=ws) == 1          # 32 x 4 = 128 tiles, but 32 row tiles
    assert ops.gemm_splitk_factor(grp[:1], ops.EPI_BIAS_GATE_RES) == 1                    # no workspace
    before = out.clone()
    for kw in (dict(splitk_ws=ws), dict()):
        with pytest.raises(N.OmniNativeError):
            ops.gemm(grp, ops.EPI_BIAS_GATE_RES, kernel_hint=ops.GEMM_KERNEL_SPLITK_DEFER_FINISH, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out, before)


def test_forward_with_deferred_finishes_equals_the_forward_without_them():
    """The denoise loop at one 256x256 CFG pair, 3 full-width layers.  Plain, both K-split GEMMs of a block defer their finish
    (the MLP-down's into the NEXT block's norm1: the pipeline feeds the modulation vectors from the request's table).  With
    TeaCache attached and a threshold that never skips, every GEMM carries tile predicates and nothing is deferred — the finish
    kernels and the AdaLN launches run — while the arithmetic is the same (never-skip == uncached bit for bit,
    tests/test_gpu_teacache.py).  Equal bits pin the wiring of the fused path (bias, gate, the norm1 / norm2 vectors of the
    right block, the row offsets of the two streams) to the three-kernel sequence at the production width."""
    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig, TransformerConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    dev = torch.device(DEV)
    cfg = OmniDiffusionConfig(model="x", tf_model_config=TransformerConfig.from_dict({"num_layers": 3}))
    pipe = QwenImagePipeline(od_config=cfg, device=dev)
    pipe.transformer.init_random_(seed=1234)

    def run():
        gg = torch.Generator().manual_seed(5)
        req = OmniDiffusionRequest(height=256, width=256, num_inference_steps=3, true_cfg_scale=4.0,
                                   latents=torch.randn(1, 256, 64, generator=gg).to(dev, BF16),
                                   prompt_embeds=torch.randn(1, 64, 3584, generator=gg).to(dev, BF16),
                                   negative_prompt_embeds=torch.randn(1, 64, 3584, generator=gg).to(dev, BF16), output_type="latent")
        out = pipe.generate([req], output_type="latent")[0].output.clone()
        torch.cuda.synchronize()
        return out

    fused = run()
    pipe.transformer.teacache = TeaCacheConfig(rel_l1_thresh=1e-12)
    try:
        unfused = run()
        assert pipe.last_teacache_state.skipped_forwards() == [0, 0]
    finally:
        pipe.transformer.teacache = None
    assert torch.isfinite(fused.float()).all()
    assert torch.equal(fused, unfused)
