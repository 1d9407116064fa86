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


def test_attention_8wave_workgroups_equal_4wave_workgroups_bitwise():
    """B*H*ceil(S/256) >= 1536 selects 256-query workgroups (8 waves), below that 128-query ones (4 waves): the per-wave
    arithmetic is the same, so running the batch in one call or in two halves must give identical bits."""
    from vllm_omni_amd import ops

    B, H, S = 8, 24, 2048
    q, k, v = rnd(B * S, H * 128, seed=11), rnd(B * S, H * 128, seed=12), rnd(B * S, H * 128, seed=13)
    cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(DEV)
    full = ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128))
    h = B // 2 * S
    cu_h = (torch.arange(B // 2 + 1, dtype=torch.int32) * S).to(DEV)
    lo = ops.flash_attn_varlen(q[:h], k[:h], v[:h], cu_h, H, S, 1 / math.sqrt(128))
    hi = ops.flash_attn_varlen(q[h:], k[h:], v[h:], cu_h, H, S, 1 / math.sqrt(128))
    assert torch.equal(full, torch.cat([lo, hi]))
    ref = torch.nn.functional.scaled_dot_product_attention(
        *(t[:S].view(1, S, H, 128).permute(0, 2, 1, 3).float() for t in (q, k, v))).permute(0, 2, 1, 3).reshape(S, H * 128)
    assert ((full[:S].float() - ref).norm() / ref.norm()) < 4e-3


def test_fullwidth_layer_gemms_blocked_layouts_and_fused_qk_epilogue_are_bit_identical():
    """At D=3072, 24 heads, one CFG pair (8192 + 128 rows): K32-blocked A / W / out and the fused q/k norm+RoPE epilogue
    against the row-major two-kernel path — identical bits (the layouts only move bytes; the fused math is the same)."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.diffusion.batch import build_ragged_batch
    from vllm_omni_amd.diffusion.models.qwen_image.rope import rope_table

    D, H, T, grid = 3072, 24, 64, (1, 64, 64)
    rb = build_ragged_batch([T, T], grid, [0, 1], T)
    maps = rb.device_maps(torch.device(DEV))
    Ri, Rt, Rj = rb.n_img_rows, rb.n_txt_rows, rb.n_joint_rows
    cos, sin = rope_table(grid, T)
    cosb, sinb = cos.to(DEV, BF16), sin.to(DEV, BF16)
    xi, xt = rnd(Ri, D, seed=21), rnd(Rt, D, seed=22)
    wi, wt = rnd(3 * D, D, seed=23, s=0.02), rnd(3 * D, D, seed=24, s=0.02)
    bi, bt = rnd(3 * D, seed=25, s=0.1), rnd(3 * D, seed=26, s=0.1)
    nw = [(1 + rnd(128, seed=30 + i, s=0.1).float()).to(BF16) for i in range(4)]
    jp = maps["joint_pos"]
    outs = []
    for fancy in (False, True):
        q = torch.zeros(Rj, D, dtype=BF16, device=DEV)
        k, v = torch.zeros_like(q), torch.zeros_like(q)
        a_i, a_t = (ops.w_to_k32_blocked(xi), ops.w_to_k32_blocked(xt)) if fancy else (xi, xt)
        w_i, w_t = (ops.w_to_k32_blocked(wi), ops.w_to_k32_blocked(wt)) if fancy else (wi, wt)
        kw_i = dict(qk_norm_q_w=nw[0], qk_norm_k_w=nw[1], qk_rope_cos=cosb, qk_rope_sin=sinb,
                    qk_row_pos=jp[maps["img_joint_row"].long()].contiguous()) if fancy else {}
        kw_t = dict(qk_norm_q_w=nw[2], qk_norm_k_w=nw[3], qk_rope_cos=cosb, qk_rope_sin=sinb,
                    qk_row_pos=jp[maps["txt_joint_row"].long()].contiguous()) if fancy else {}
        ops.gemm([ops.GemmGroupArgs(a_i, w_i, bi, q, out1=k, out2=v, out_row_map=maps["img_joint_row"],
                                    a_k32_blocked=fancy, **kw_i),
                  ops.GemmGroupArgs(a_t, w_t, bt, q, out1=k, out2=v, out_row_map=maps["txt_joint_row"],
                                    a_k32_blocked=fancy, **kw_t)],
                 ops.EPI_BIAS_SPLIT3_QKNORM_ROPE if fancy else ops.EPI_BIAS_SPLIT3, split_n=D, w_k32_blocked=fancy)
        if not fancy:
            ops.qk_norm_rope_(q, H, nw[0], nw[2], cosb, sinb, jp, rb.txt_pos_end)
            ops.qk_norm_rope_(k, H, nw[1], nw[3], cosb, sinb, jp, rb.txt_pos_end)
        outs.append((q, k, v))
    torch.cuda.synchronize()
    for name, a, b in zip("qkv", *outs):
        ndiff = int((a != b).sum())
        assert ndiff == 0, f"{name}: {ndiff} of {a.numel()} elements differ, max |d| {float((a.float() - b.float()).abs().max())}"
    assert torch.isfinite(outs[1][0].float()).all() and outs[1][0].float().abs().mean() > 0.05


def test_forward_2048px_token_count_item_order_equivariance():
    """BASELINE config 5 geometry (2048x2048 -> 16384 image tokens per item, joint 16448): one full-width layer, two items
    of different text length in one ragged forward; swapping the items swaps the outputs.  Exercises the largest sequence
    the path names: 258 key tiles per attention row block, 130 m-tiles per GEMM, RoPE table of 16384 + text rows."""
    from vllm_omni_amd.diffusion.batch import build_ragged_batch
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    m = QwenImageTransformer2DModel(num_layers=1, device=DEV).init_random_(seed=5)
    grid, S = (1, 128, 128), 16384
    lat = [rnd(S, 64, seed=30 + i) for i in range(2)]
    txt = [rnd(t, 3584, seed=40 + i) for i, t in enumerate((64, 19))]
    sig = torch.tensor([0.37, 0.37], dtype=torch.float32, device=DEV)
    ab = m.forward_ragged(m.prepare_batch(build_ragged_batch([64, 19], grid)), torch.cat(lat), torch.cat(txt), sig).clone()
    ba = m.forward_ragged(m.prepare_batch(build_ragged_batch([19, 64], grid)), torch.cat(lat[::-1]), torch.cat(txt[::-1]),
                          sig).clone()
    torch.cuda.synchronize()
    assert ab.shape == (2 * S, 64) and torch.isfinite(ab.float()).all() and ab.float().abs().mean() > 1e-3
    d0 = (ab[:S].float() - ba[S:].float()).norm() / ab[:S].float().norm()
    d1 = (ab[S:].float() - ba[:S].float()).norm() / ab[S:].float().norm()
    assert d0 < 5e-3 and d1 < 5e-3
