"""GPU, BASELINE full sizes (1024x1024 -> 4096 image tokens + 64 text tokens per item, D=3072, 24 heads):
size-independent properties that need no CPU oracle at that size (the oracle cannot finish these in seconds)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def rnd(*shape, seed=0, s=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * s).to(BF16)


def test_attention_fullsize_convexity_and_ones():
    """softmax rows are convex weights: V = const -> O = const exactly; O within [min V, max V] per head/dim."""
    from vllm_omni_amd import ops

    B, H, S = 2, 24, 4096 + 64
    q, k = rnd(B * S, H * 128, seed=1), rnd(B * S, H * 128, seed=2)
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(DEV)
    ones = torch.full((B * S, H * 128), 0.75, dtype=BF16, device=DEV)
    o = ops.flash_attn_varlen(q, k, ones, cu, H, S, 1 / math.sqrt(128))
    assert torch.equal(o, ones)                     # sum(p)/l == 1 to within bf16 rounding of 0.75
    v = rnd(B * S, H * 128, seed=3)
    o = ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128)).float().view(B, S, H, 128)
    vv = v.float().view(B, S, H, 128)
    assert (o <= vv.amax(1, keepdim=True) + 1e-2).all() and (o >= vv.amin(1, keepdim=True) - 1e-2).all()


def test_attention_fullsize_key_permutation_invariance():
    """attention is invariant to a permutation of the (key, value) rows of an item."""
    from vllm_omni_amd import ops

    H, S = 24, 4096 + 64
    q, k, v = rnd(S, H * 128, seed=4), rnd(S, H * 128, seed=5), rnd(S, H * 128, seed=6)
    cu = torch.tensor([0, S], dtype=torch.int32, device=DEV)
    perm = torch.randperm(S, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))
    a = ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128)).float()
    b = ops.flash_attn_varlen(q, k[perm].contiguous(), v[perm].contiguous(), cu, H, S, 1 / math.sqrt(128)).float()
    # each ordering rounds P to bf16 against a different running max and rounds O to bf16 once: two independent
    # ~2e-3 errors -> bound their difference by 6e-3 (per-op attention tolerance is 4e-3 vs the exact result)
    assert ((a - b).norm() / a.norm()) < 6e-3


def test_gemm_fullsize_linearity_and_row_gather():
    """Y(a1 + a2) == Y(a1) + Y(a2) (no bias) at the MLP-up shape; gathering rows == permuting outputs (bit-exact)."""
    from vllm_omni_amd import ops

    M, N, K = 8192 + 128, 12288, 3072
    a1, a2, w = rnd(M, K, seed=1, s=0.5), rnd(M, K, seed=2, s=0.5), rnd(N, K, seed=3, s=0.02)
    asum = (a1.float() + a2.float()).to(BF16)
    exact = (asum.float() == a1.float() + a2.float()).all(dim=1)        # rows where the bf16 sum is exact
    y1, y2, ys = ops.linear(a1, w).float(), ops.linear(a2, w).float(), ops.linear(asum, w).float()
    err = ((ys - (y1 + y2))[exact].norm() / ys[exact].norm()) if exact.any() else torch.tensor(0.0)
    assert err < 6e-3                                                    # 3 bf16 output roundings
    perm = torch.randperm(M, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9)).int()
    yg = torch.empty(M, N, dtype=BF16, device=DEV)
    ops.gemm([ops.GemmGroupArgs(a1, w, None, yg, a_row_map=perm)])
    assert torch.equal(yg, ops.linear(a1, w)[perm.long()])


def test_forward_fullwidth_item_order_equivariance():
    """Full-width DiT (24 heads, joint 3584), 2 layers, 1024^2 tokens: swapping the two items of a ragged batch swaps
    the outputs (per-request semantics at full size), and a CFG pair sharing one temb row == two equal rows."""
    from vllm_omni_amd.diffusion.batch import build_ragged_batch
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    m = QwenImageTransformer2DModel(num_layers=2, device=DEV).init_random_(seed=3)
    grid, S = (1, 64, 64), 4096
    lat = [rnd(S, 64, seed=10 + i) for i in range(2)]
    txt = [rnd(t, 3584, seed=20 + i) for i, t in enumerate((64, 37))]
    sig = torch.tensor([0.61], dtype=torch.float32, device=DEV)
    ab = m.forward_ragged(m.prepare_batch(build_ragged_batch([64, 37], grid, temb_rows=[0, 0])),
                          torch.cat(lat), torch.cat(txt), sig).clone()
    ba = m.forward_ragged(m.prepare_batch(build_ragged_batch([37, 64], grid, temb_rows=[0, 0])),
                          torch.cat(lat[::-1]), torch.cat(txt[::-1]), sig).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(ab.float()).all()
    d0 = (ab[:S].float() - ba[S:].float()).norm() / ab[:S].float().norm()
    d1 = (ab[S:].float() - ba[:S].float()).norm() / ab[S:].float().norm()
    assert d0 < 5e-3 and d1 < 5e-3          # same math per request; only GEMM tile membership differs


def test_cfg_euler_fullsize_scale_one_is_positive_branch():
    """true_cfg_scale == 1: comb = pos and the norm ratio is 1 -> identical to the no-CFG update."""
    from vllm_omni_amd import ops

    rows = 3 * 4096
    pos, neg, lat = rnd(rows, 64, seed=1), rnd(rows, 64, seed=2), rnd(rows, 64, seed=3)
    dt = torch.tensor([-0.04], dtype=torch.float32, device=DEV)
    a, b = lat.clone(), lat.clone()
    ops.cfg_euler_step_(a, pos, neg, 1.0, dt)
    ops.cfg_euler_step_(b, pos, None, 1.0, dt)
    assert ((a.float() - b.float()).abs().max()) <= 2 ** -7 * lat.float().abs().max()
