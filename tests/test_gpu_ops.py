"""GPU parity: every C-ABI kernel against the fp32 oracle ops on the same seeded, bf16-rounded inputs.

Tolerances (bf16 storage, fp32 accumulate, one rounding):
  elementwise / norm ops : |err| <= 2^-7 * (|ref| + rms(ref)) per element   (+ rel_l2 <= 4e-3)
  GEMM / attention       : rel_l2 <= 4e-3
All inputs are rounded to bf16 first so that both sides see identical bits.
"""
import math

import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16_round(torch.randn(shape, generator=g) * scale)


def g_(t):
    return t.to(dev(), BF16).contiguous()


def elem_close(got, ref, what=""):
    """bf16 output rounding is relative (2^-8 of the value): bound the error by 2^-7 * (|ref| + rms(ref))."""
    ref = ref.float()
    got = got.float().cpu()
    rms = ref.pow(2).mean().sqrt()
    err = ((got - ref).abs() / (ref.abs() + rms)).max()
    assert err <= 2 ** -7, f"{what}: max scaled err {err}"
    assert rel_l2(got, ref) <= 4e-3, f"{what}: rel_l2 {rel_l2(got, ref)}"


def test_library_loads_on_gpu_box():
    from vllm_omni_amd import _native as N

    assert N.lib().omni_abi_version() == N.ABI_VERSION
    assert N.lib().omni_build_arch() == b"gfx950"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 264, 128), (1024, 768, 256), (77, 64, 3072),
                                   (4096 + 64, 1024, 512), (513, 3072, 64)])
@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_bias_and_gelu(M, N, K, gelu):
    from vllm_omni_amd import ops

    a, w, b = rnd((M, K), 1), rnd((N, K), 2, 0.05), rnd((N,), 3, 0.5)
    ref = a @ w.t() + b
    if gelu:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    got = ops.linear(g_(a), g_(w), g_(b), gelu=gelu)
    torch.cuda.synchronize()
    assert rel_l2(got, ref) <= 4e-3


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 264, 128), (77, 72, 3072), (4096 + 64, 1024, 512)])
def test_gemm_k32_blocked_weight_layout_is_bit_identical(M, N, K):
    """w_k32_blocked=1 only changes WHERE the weight bytes are fetched from: same MFMA order -> identical bits."""
    from vllm_omni_amd import ops

    a, w, b = rnd((M, K), 11), rnd((N, K), 12, 0.05), rnd((N,), 13, 0.5)
    wb = ops.w_to_k32_blocked(g_(w))
    assert torch.equal(wb.view(K // 32, N, 32)[3 % (K // 32), 5, 7].cpu(), g_(w)[5, (3 % (K // 32)) * 32 + 7].cpu())
    y0 = ops.linear(g_(a), g_(w), g_(b), gelu=True)
    y1 = ops.linear(g_(a), wb, g_(b), gelu=True, w_k32_blocked=True)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert rel_l2(y1, torch.nn.functional.gelu(a @ w.t() + b, approximate="tanh")) <= 4e-3


def test_gemm_k32_blocked_activations_are_bit_identical():
    """a_k32_rows / out_k32_rows (grouped, ragged M, gathered A rows): same bits as the row-major run, re-laid-out."""
    from vllm_omni_amd import ops

    Mi, Mt, N, K, R = 700, 130, 512, 256, 1000          # A buffers have more rows (R) than the groups use
    a = rnd((R, K), 21)
    wi, wt, b = rnd((N, K), 22, 0.05), rnd((N, K), 23, 0.05), rnd((N,), 24, 0.5)
    gi = torch.Generator().manual_seed(5)
    map_i = torch.randperm(R, generator=gi)[:Mi].to(torch.int32)
    map_t = torch.randperm(R, generator=gi)[:Mt].to(torch.int32)
    a_rm, a_blk = g_(a), ops.w_to_k32_blocked(g_(a))
    outs = []
    for blocked in (False, True):
        oi = torch.zeros(Mi, N, dtype=BF16, device=dev())
        ot = torch.zeros(Mt, N, dtype=BF16, device=dev())
        A = a_blk if blocked else a_rm
        ops.gemm([ops.GemmGroupArgs(A, g_(wi), g_(b), oi, a_row_map=map_i.to(dev()), a_k32_blocked=blocked,
                                    out_k32_blocked=blocked),
                  ops.GemmGroupArgs(A, g_(wt), g_(b), ot, a_row_map=map_t.to(dev()), a_k32_blocked=blocked,
                                    out_k32_blocked=blocked)], ops.EPI_BIAS_GELU_TANH)
        torch.cuda.synchronize()
        outs.append((ops.k32_blocked_to_rows(oi), ops.k32_blocked_to_rows(ot)) if blocked else (oi, ot))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.nn.functional.gelu(a[map_i.long()] @ wi.t() + b, approximate="tanh")
    assert rel_l2(outs[1][0], ref) <= 4e-3


def test_adaln_and_attention_k32_blocked_outputs_are_bit_identical():
    from vllm_omni_amd import ops

    rows, D = 333, 3072
    x, mod = rnd((rows, D), 31), rnd((2, 6 * D), 32, 0.3)
    item = (torch.arange(rows) % 2).to(torch.int32).to(dev())
    kw = dict(mod_item_stride=6 * D, row_item_map=item)
    y0 = ops.adaln_modulate(g_(x), g_(mod)[:, D:], g_(mod), **kw)
    y1 = ops.adaln_modulate(g_(x), g_(mod)[:, D:], g_(mod), out_k32_blocked=True, **kw)
    H, lens = 3, [200, 77]
    tot = sum(lens)
    q, k, v = (g_(rnd((tot, H * 128), 40 + i)) for i in range(3))
    cu = torch.tensor([0, lens[0], tot], dtype=torch.int32, device=dev())
    o0 = ops.flash_attn_varlen(q, k, v, cu, H, max(lens), 1 / math.sqrt(128))
    o1 = ops.flash_attn_varlen(q, k, v, cu, H, max(lens), 1 / math.sqrt(128), out_k32_blocked=True)
    torch.cuda.synchronize()
    assert torch.equal(y0, ops.k32_blocked_to_rows(y1))
    assert torch.equal(o0, ops.k32_blocked_to_rows(o1))


@pytest.mark.parametrize("K", [64, 128, 192, 3072])
@pytest.mark.parametrize("blocked", [False, True])
def test_gemm_pingpong_kernel_is_bit_identical_to_ring_kernel(K, blocked):
    """The ping-pong kernel (default; v_mfma_f32_16x16x32_bf16) and the ring kernel (its fallback, forced here through
    omni_gemm_params.kernel_hint = OMNI_GEMM_KERNEL_RING; 32x32x16) accumulate
    every output element in the same k order; on gfx950 both MFMA shapes add their products to the fp32 accumulator in
    8-k groups in ascending k (measured: identical bits, here and at the bench shapes in tools/bench_ab.py), so the two
    kernels agree bit for bit — for 1 / 2 / 3 / many K-tiles (prologue, steady state and drain of the 6-phase DMA lead),
    ragged M in both groups, gathered A rows, both operand layouts, all epilogues that have a coalesced form."""
    from vllm_omni_amd import ops

    Mi, Mt, N, R = 700, 130, 768, 1000
    a = rnd((R, K), 21)
    wi, wt, b = rnd((N, K), 22, 0.05), rnd((N, K), 23, 0.05), rnd((N,), 24, 0.5)
    res_i, res_t = g_(rnd((Mi, N), 25)), g_(rnd((Mt, N), 26))
    gate = g_(rnd((3, N), 27))
    gi = torch.Generator().manual_seed(5)
    map_i = torch.randperm(R, generator=gi)[:Mi].to(torch.int32).to(dev())
    map_t = torch.randperm(R, generator=gi)[:Mt].to(torch.int32).to(dev())
    A = ops.w_to_k32_blocked(g_(a)) if blocked else g_(a)
    Wi = ops.w_to_k32_blocked(g_(wi)) if blocked else g_(wi)
    Wt = ops.w_to_k32_blocked(g_(wt)) if blocked else g_(wt)
    item_i = (torch.arange(Mi) % 3).to(torch.int32).to(dev())
    item_t = (torch.arange(Mt) % 3).to(torch.int32).to(dev())
    outs = {}
    if True:
        for variant in (1, 3):
            hint = ops.GEMM_KERNEL_RING if variant == 1 else ops.GEMM_KERNEL_AUTO
            got = []
            for epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU_TANH, ops.EPI_BIAS_GATE_RES):
                oi = res_i.clone() if epi == ops.EPI_BIAS_GATE_RES else torch.zeros(Mi, N, dtype=BF16, device=dev())
                ot = res_t.clone() if epi == ops.EPI_BIAS_GATE_RES else torch.zeros(Mt, N, dtype=BF16, device=dev())
                kw = dict(a_k32_blocked=blocked)
                # the SADDR (32-bit offset) DMA form does not take gathered row-major rows: gather only in the blocked layout
                mi, mt = (map_i, map_t) if blocked else (None, None)
                gkw_i = dict(res=oi, gate=gate, gate_item_stride=N, row_item_map=item_i) if epi == ops.EPI_BIAS_GATE_RES else {}
                gkw_t = dict(res=ot, gate=gate, gate_item_stride=N, row_item_map=item_t) if epi == ops.EPI_BIAS_GATE_RES else {}
                Ai = A if blocked else A[:Mi]
                At = A if blocked else A[Mi:Mi + Mt]
                ops.gemm([ops.GemmGroupArgs(Ai, Wi, g_(b), oi, a_row_map=mi, **kw, **gkw_i),
                          ops.GemmGroupArgs(At, Wt, g_(b), ot, a_row_map=mt, **kw, **gkw_t)], epi, w_k32_blocked=blocked,
                         kernel_hint=hint)
                torch.cuda.synchronize()
                got += [oi, ot]
            outs[variant] = got
    for x, y in zip(outs[1], outs[3]):
        assert torch.equal(x, y), "the ping-pong kernel differs from the ring kernel"
    rows = map_i.long().cpu() if blocked else torch.arange(Mi)
    assert rel_l2(outs[3][0], a[rows] @ wi.t() + b) <= 4e-3


def test_gemm_transpose_detecting_identity():
    # A = I (asymmetric W): catches swapped row/col in the MFMA accumulator write (cdna guide rule 16)
    from vllm_omni_amd import ops

    K = 256
    a = torch.eye(K)
    w = bf16_round(torch.arange(320 * K, dtype=torch.float32).view(320, K) % 251 / 16.0)
    got = ops.linear(g_(a), g_(w))
    torch.cuda.synchronize()
    assert torch.equal(got.float().cpu(), w.t().contiguous())


def test_gemm_grouped_gate_residual_with_row_maps():
    from vllm_omni_amd import ops

    D, K = 512, 256
    M0, M1, items = 600, 90, 3
    a_src0, a_src1 = rnd((700, K), 1), rnd((128, K), 2)
    map0 = torch.randperm(700, generator=torch.Generator().manual_seed(5))[:M0].int()
    map1 = torch.randperm(128, generator=torch.Generator().manual_seed(6))[:M1].int()
    w0, w1, b0, b1 = rnd((D, K), 3, 0.05), rnd((D, K), 4, 0.05), rnd((D,), 7, 0.3), rnd((D,), 8, 0.3)
    res0, res1 = rnd((M0, D), 9), rnd((M1, D), 10)
    gate = rnd((items, 6 * D), 11)          # gate vector lives inside a [items, 6D] modulation tensor
    item0 = (torch.arange(M0) * items // M0).int()
    item1 = (torch.arange(M1) % items).int()
    ref0 = res0 + gate[item0.long(), 2 * D:3 * D] * (a_src0[map0.long()] @ w0.t() + b0)
    ref1 = res1 + gate[item1.long(), 2 * D:3 * D] * (a_src1[map1.long()] @ w1.t() + b1)
    r0, r1, gd = g_(res0), g_(res1), g_(gate)
    ops.gemm([ops.GemmGroupArgs(g_(a_src0), g_(w0), g_(b0), r0, a_row_map=map0.to(dev()), res=r0,
                                gate=gd[:, 2 * D:], gate_item_stride=6 * D, row_item_map=item0.to(dev())),
              ops.GemmGroupArgs(g_(a_src1), g_(w1), g_(b1), r1, a_row_map=map1.to(dev()), res=r1,
                                gate=gd[:, 2 * D:], gate_item_stride=6 * D, row_item_map=item1.to(dev()))],
             ops.EPI_BIAS_GATE_RES)
    torch.cuda.synchronize()
    assert rel_l2(r0, ref0) <= 4e-3 and rel_l2(r1, ref1) <= 4e-3


def test_gemm_split3_scatter_to_joint():
    from vllm_omni_amd import ops

    D, K, Mi, Mt = 256, 256, 300, 20
    xi, xt = rnd((Mi, K), 1), rnd((Mt, K), 2)
    wi, wt, bi, bt = rnd((3 * D, K), 3, 0.05), rnd((3 * D, K), 4, 0.05), rnd((3 * D,), 5), rnd((3 * D,), 6)
    rows = Mi + Mt
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(7)).int()
    mi, mt = perm[:Mi], perm[Mi:]
    q = torch.zeros(rows, D, dtype=BF16, device=dev())
    k, v = torch.zeros_like(q), torch.zeros_like(q)
    ops.gemm([ops.GemmGroupArgs(g_(xi), g_(wi), g_(bi), q, out1=k, out2=v, out_row_map=mi.to(dev())),
              ops.GemmGroupArgs(g_(xt), g_(wt), g_(bt), q, out1=k, out2=v, out_row_map=mt.to(dev()))],
             ops.EPI_BIAS_SPLIT3, split_n=D)
    torch.cuda.synchronize()
    ref = torch.zeros(rows, 3 * D)
    ref[mi.long()] = xi @ wi.t() + bi
    ref[mt.long()] = xt @ wt.t() + bt
    for got, j in ((q, 0), (k, 1), (v, 2)):
        assert rel_l2(got, ref[:, j * D:(j + 1) * D]) <= 4e-3


def test_gemm_split3_fused_qk_norm_rope_is_bit_identical_to_the_two_kernel_path():
    """OMNI_EPI_BIAS_SPLIT3_QKNORM_ROPE == SPLIT3 followed by omni_qk_norm_rope on q and k (and v untouched)."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.diffusion.models.qwen_image.rope import rope_table

    H, D, K = 2, 256, 128
    T, grid = 20, (1, 18, 16)                      # 20 text rows, 288 image rows -> two m-tiles for the image group
    Mi, Mt = grid[1] * grid[2], T
    rows = Mi + Mt
    xi, xt = rnd((Mi, K), 1), rnd((Mt, K), 2)
    wi, wt, bi, bt = rnd((3 * D, K), 3, 0.05), rnd((3 * D, K), 4, 0.05), rnd((3 * D,), 5), rnd((3 * D,), 6)
    nw = [bf16_round(rnd((128,), 10 + i, 0.2) + 1) for i in range(4)]      # q_img, k_img, q_txt, k_txt
    cos, sin = rope_table(grid, T)
    cosb, sinb = g_(bf16_round(cos)), g_(bf16_round(sin))
    joint_pos = torch.arange(rows, dtype=torch.int32)                        # joint order: text first, then image
    mt, mi = joint_pos[:T].clone(), joint_pos[T:].clone()                     # out_row_map of each stream
    outs = []
    for fused in (False, True):
        q = torch.zeros(rows, D, dtype=BF16, device=dev())
        k, v = torch.zeros_like(q), torch.zeros_like(q)
        kw_i = dict(qk_norm_q_w=g_(nw[0]), qk_norm_k_w=g_(nw[1]), qk_rope_cos=cosb, qk_rope_sin=sinb,
                    qk_row_pos=joint_pos[mi.long()].to(dev())) if fused else {}
        kw_t = dict(qk_norm_q_w=g_(nw[2]), qk_norm_k_w=g_(nw[3]), qk_rope_cos=cosb, qk_rope_sin=sinb,
                    qk_row_pos=joint_pos[mt.long()].to(dev())) if fused else {}
        ops.gemm([ops.GemmGroupArgs(g_(xi), g_(wi), g_(bi), q, out1=k, out2=v, out_row_map=mi.to(dev()), **kw_i),
                  ops.GemmGroupArgs(g_(xt), g_(wt), g_(bt), q, out1=k, out2=v, out_row_map=mt.to(dev()), **kw_t)],
                 ops.EPI_BIAS_SPLIT3_QKNORM_ROPE if fused else ops.EPI_BIAS_SPLIT3, split_n=D)
        if not fused:
            ops.qk_norm_rope_(q, H, g_(nw[0]), g_(nw[2]), cosb, sinb, joint_pos.to(dev()), T)
            ops.qk_norm_rope_(k, H, g_(nw[1]), g_(nw[3]), cosb, sinb, joint_pos.to(dev()), T)
        torch.cuda.synchronize()
        outs.append((q, k, v))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_gemm_rejects_bad_k():
    from vllm_omni_amd import ops
    from vllm_omni_amd._native import OmniNativeError

    with pytest.raises(OmniNativeError):
        ops.linear(g_(rnd((64, 40), 1)), g_(rnd((64, 40), 2)))


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,D,items", [(64, 256, 1), (300, 512, 3), (130, 3072, 2)])
def test_adaln_modulate(rows, D, items):
    from vllm_omni_amd import ops

    x, mod = rnd((rows, D), 1, 3.0) + 0.5, rnd((items, 6 * D), 2, 0.5)
    x = bf16_round(x)
    item = (torch.arange(rows) % items).int()
    m = mod[item.long()]
    ref = torch.nn.functional.layer_norm(x, (D,), eps=1e-6) * (1 + m[:, D:2 * D]) + m[:, :D]
    md = g_(mod)
    got = ops.adaln_modulate(g_(x), md[:, D:], md, mod_item_stride=6 * D, row_item_map=item.to(dev()))
    torch.cuda.synchronize()
    elem_close(got, ref, "adaln")
    # uniform rows_per_item path
    if rows % items == 0:
        m2 = mod.repeat_interleave(rows // items, 0)
        ref2 = torch.nn.functional.layer_norm(x, (D,), eps=1e-6) * (1 + m2[:, D:2 * D]) + m2[:, :D]
        got2 = ops.adaln_modulate(g_(x), md[:, D:], md, mod_item_stride=6 * D, rows_per_item=rows // items)
        elem_close(got2, ref2, "adaln-uniform")


def test_rmsnorm_txt():
    from vllm_omni_amd import ops

    x, w = rnd((77, 3584), 1, 2.0), rnd((3584,), 2, 0.1) + 1.0
    w = bf16_round(w)
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
    elem_close(ops.rmsnorm(g_(x), g_(w)), ref, "rmsnorm")


def test_qk_norm_rope_joint():
    from vllm_omni_amd import ops
    from vllm_omni_amd.diffusion.models.qwen_image.rope import rope_table

    H, T, grid = 4, 9, (1, 6, 4)
    S = grid[1] * grid[2]
    rows = T + S
    x = rnd((rows, H * 128), 1, 2.0)
    wi, wt = bf16_round(rnd((128,), 2, 0.2) + 1), bf16_round(rnd((128,), 3, 0.2) + 1)
    cos, sin = rope_table(grid, T)
    cosb, sinb = bf16_round(cos), bf16_round(sin)
    pos = torch.arange(rows, dtype=torch.int32)   # rows 0..T-1 text, then image
    xh = x.view(1, rows, H, 128)
    xt = O.rope_interleaved(O.rms_norm(xh[:, :T], wt), cosb[:T], sinb[:T])
    xi = O.rope_interleaved(O.rms_norm(xh[:, T:], wi), cosb[T:], sinb[T:])
    ref = torch.cat([xt, xi], 1).reshape(rows, H * 128)
    xd = g_(x)
    ops.qk_norm_rope_(xd, H, g_(wi), g_(wt), g_(cosb), g_(sinb), pos.to(dev()), T)
    torch.cuda.synchronize()
    elem_close(xd, ref, "qk_norm_rope")


def test_rope_table_matches_oracle():
    from vllm_omni_amd.diffusion.models.qwen_image.rope import rope_table

    for grid, T in (((1, 8, 8), 7), ((1, 16, 8), 13), ((1, 64, 64), 64), ((1, 5, 9), 3)):
        cos, sin = rope_table(grid, T)
        (vc, vs), (tc, ts) = O.rope_tables(*grid, T)
        assert torch.equal(cos[:T], tc) and torch.equal(sin[:T], ts)
        assert torch.equal(cos[T:], vc) and torch.equal(sin[T:], vs)


def test_rope_interleaved_plugin():
    from vllm_omni_amd import ops

    B, S, H, dh = 2, 33, 3, 128
    x = rnd((B, S, H, dh), 1)
    cos, sin = bf16_round(torch.cos(rnd((S, 64), 2))), bf16_round(torch.sin(rnd((S, 64), 2)))
    ref = O.rope_interleaved(x, cos, sin)
    elem_close(ops.rope_interleaved(g_(x), g_(cos), g_(sin)), ref, "rope")


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, lens, H):
    outs, s0 = [], 0
    for L in lens:
        qq, kk, vv = (t[s0:s0 + L].view(1, L, H, 128) for t in (q, k, v))
        outs.append(O.sdpa_nhd(qq, kk, vv, 1 / math.sqrt(128)).reshape(L, H * 128))
        s0 += L
    return torch.cat(outs)


@pytest.mark.parametrize("lens,H", [([64], 1), ([128], 2), ([71], 2), ([4160 // 8 + 7, 300], 3), ([1, 65, 129], 2),
                                    ([1024 + 64], 4)])
def test_flash_attention_varlen(lens, H):
    from vllm_omni_amd import ops

    rows = sum(lens)
    q, k, v = rnd((rows, H * 128), 1), rnd((rows, H * 128), 2), rnd((rows, H * 128), 3)
    ref = attn_ref(q, k, v, lens, H)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev())
    got = ops.flash_attn_varlen(g_(q), g_(k), g_(v), cu, H, max(lens), 1 / math.sqrt(128))
    torch.cuda.synchronize()
    assert rel_l2(got, ref) <= 4e-3
    assert (got.float().cpu() - ref).abs().max() <= 2e-2


def test_flash_attention_forced_rescale_spike():
    # one key row spiked against one query so the running max jumps late in the sequence (guide rule 26)
    from vllm_omni_amd import ops

    H, L = 1, 512
    q, k, v = rnd((L, 128), 1, 0.3), rnd((L, 128), 2, 0.3), rnd((L, 128), 3)
    k[400] = bf16_round(q[17] * 40.0)
    k[70] = bf16_round(q[300] * 25.0)
    ref = attn_ref(q, k, v, [L], H)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev())
    got = ops.flash_attn_varlen(g_(q), g_(k), g_(v), cu, H, L, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    assert rel_l2(got, ref) <= 4e-3


def test_flash_attention_first_tile_far_below_zero_stays_finite():
    """Round-3 advisor finding (small-grid pipelined kernel, max baked into the accumulator): tile 0 'rescales' o = l = 0 by
    exp2(-rowmax); a row whose every first-tile score sits below about -128 in the exp2 domain made that factor +inf and the whole
    output row NaN.  All keys share a large component that one query opposes: its scores are ~ -150 in the exp2 domain."""
    from vllm_omni_amd import ops

    for L in (256, 272, 1024):                    # 1 - 4 q-blocks, one item, one head: the small-grid kernel
        q, k, v = rnd((L, 128), 1, 0.3), rnd((L, 128), 2, 0.3), rnd((L, 128), 3)
        k[:, 0] = 4.0
        q[5, 0] = -300.0
        q[L - 1, 0] = -400.0
        ref = attn_ref(q, k, v, [L], 1)
        cu = torch.tensor([0, L], dtype=torch.int32, device=dev())
        got = ops.flash_attn_varlen(g_(q), g_(k), g_(v), cu, 1, L, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all(), L
        assert rel_l2(got, ref) <= 4e-3, L
        assert rel_l2(got[5], ref[5]) <= 8e-3 and rel_l2(got[L - 1], ref[L - 1]) <= 8e-3, L


@pytest.mark.parametrize("lens,H", [([1100, 300, 257, 65, 1, 647], 24), ([2100], 64), ([256, 512, 1000, 64, 63], 32),
                                    ([1056, 1057, 1088, 1089, 1055, 1072], 24)])   # last q-blocks of 32 / 33 / 64 / 65 / 31 / 48 rows
def test_flash_attention_w64_kernel_varlen(lens, H):
    """Grids of >= 512 workgroups run the 64-queries-per-wave kernel (csrc/attention_w64.hip): ragged items, tails that are
    not a multiple of 64 keys / 256 queries, items shorter than one tile; waves without query rows only feed the rings."""
    from vllm_omni_amd import ops

    assert len(lens) * H * ((max(lens) + 255) // 256) >= 512
    rows = sum(lens)
    q, k, v = rnd((rows, H * 128), 1), rnd((rows, H * 128), 2), rnd((rows, H * 128), 3)
    ref = attn_ref(q, k, v, lens, H)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev())
    got = ops.flash_attn_varlen(g_(q), g_(k), g_(v), cu, H, max(lens), 1 / math.sqrt(128))
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) <= 4e-3
    assert (got.float().cpu() - ref).abs().max() <= 2e-2


def test_flash_attention_w64_forced_rescale_spike():
    """The rare O-rescale path of the 64-queries-per-wave kernel (O lives in AGPRs there): spiked keys late in the sequence
    for queries of both 32-query blocks of a wave, and a row whose every score is far below zero (first-tile reference)."""
    from vllm_omni_amd import ops

    H, L = 64, 2100
    q, k, v = rnd((L, H * 128), 1, 0.3), rnd((L, H * 128), 2, 0.3), rnd((L, H * 128), 3)
    for hh, (kr, qr, f) in enumerate([(1400, 17, 40.0), (70, 300, 25.0), (2050, 40, 30.0), (900, 2099, 35.0)]):
        k[kr, hh * 128:(hh + 1) * 128] = bf16_round(q[qr, hh * 128:(hh + 1) * 128] * f)
    q[5, 4 * 128:5 * 128] = bf16_round(-k[:, 4 * 128:5 * 128].mean(0) * 200.0)      # head 4, query 5: all scores << 0
    ref = attn_ref(q, k, v, [L], H)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev())
    got = ops.flash_attn_varlen(g_(q), g_(k), g_(v), cu, H, L, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) <= 4e-3
    for hh in range(5):
        sl = slice(hh * 128, (hh + 1) * 128)
        assert rel_l2(got[:, sl], ref[:, sl]) <= 6e-3, hh


def test_attention_backend_plugin_surface():
    from vllm_omni_amd.diffusion.attention.selector import get_attn_backend

    be = get_attn_backend(128)
    assert be.get_name() == "CDNA4_FLASH" and 128 in be.get_supported_head_sizes()
    impl = be.get_impl_cls()(num_heads=2, head_size=128, softmax_scale=1 / math.sqrt(128), causal=False)
    B, S, H = 2, 200, 2
    q, k, v = rnd((B, S, H, 128), 1), rnd((B, S, H, 128), 2), rnd((B, S, H, 128), 3)
    ref = O.sdpa_nhd(q, k, v, 1 / math.sqrt(128))
    got = impl.forward(g_(q), g_(k), g_(v), None)
    assert got.shape == (B, S, H, 128) and rel_l2(got, ref) <= 4e-3


# ------------------------------------------------------------------------------------------------ small ops
@pytest.mark.parametrize("B,N,K", [(1, 3072, 256), (2, 18432, 3072), (3, 1000, 512), (1, 6 * 256, 256)])
def test_linear_smallbatch_silu_in(B, N, K):
    from vllm_omni_amd import ops

    x, w, b = rnd((B, K), 1), rnd((N, K), 2, 0.03), rnd((N,), 3, 0.2)
    ref = torch.nn.functional.silu(x) @ w.t() + b
    got = ops.linear_smallbatch(g_(x), g_(w), g_(b), act_in=1)
    torch.cuda.synchronize()
    assert rel_l2(got, ref) <= 4e-3
    ref2 = torch.nn.functional.silu(x @ w.t() + b)
    assert rel_l2(ops.linear_smallbatch(g_(x), g_(w), g_(b), act_out=1), ref2) <= 4e-3


def test_timestep_sinusoid():
    from vllm_omni_amd import ops

    t = torch.tensor([0.73046875, 0.02, 1.0], dtype=torch.float32)
    ref = O.timestep_sinusoid(t)
    got = ops.timestep_sinusoid(t.to(dev())).float().cpu()
    assert (got - ref).abs().max() <= 1.2e-2   # bf16 output rounding (2^-8) + fp32 range reduction at |arg| <= 1000


@pytest.mark.parametrize("with_neg", [True, False])
def test_cfg_euler_step(with_neg):
    from vllm_omni_amd import ops

    rows = 1000
    pos, neg, lat = rnd((rows, 64), 1), rnd((rows, 64), 2), rnd((rows, 64), 3)
    dt = torch.tensor([-0.037, -0.052], dtype=torch.float32)
    pred = O.cfg_combine(pos, neg, 4.0) if with_neg else pos
    ref = lat + dt.repeat_interleave(rows // 2)[:, None] * pred
    ld = g_(lat)
    ops.cfg_euler_step_(ld, g_(pos), g_(neg) if with_neg else None, 4.0, dt.to(dev()), dt_rows_per_item=rows // 2)
    torch.cuda.synchronize()
    elem_close(ld, ref, "cfg_euler")


@pytest.mark.parametrize("K,N", [(3072, 3072), (12288, 3072), (1024, 256)])
def test_gemm_split_k_matches_the_unsplit_kernel(K, N):
    """ABI v4 split-K (small grids: <= 64 tiles in <= 4 row tiles): fp32 partial tiles + a finishing kernel that sums them in
    split order and runs the same row-coalesced epilogue.  Same products, different fp32 summation order than the single
    pass: equal up to isolated one-ulp bf16 flips; deterministic from run to run.  Two groups, ragged M, gathered blocked A,
    bias / GELU / in-place gated residual."""
    from vllm_omni_amd import ops

    Mi, Mt, R = 512, 75, 700
    a = rnd((R, K), 31)
    wi, wt, b = rnd((N, K), 32, 0.03), rnd((N, K), 33, 0.03), rnd((N,), 34, 0.5)
    gate = g_(rnd((3, N), 35))
    gi = torch.Generator().manual_seed(6)
    map_i = torch.randperm(R, generator=gi)[:Mi].to(torch.int32).to(dev())
    map_t = torch.randperm(R, generator=gi)[:Mt].to(torch.int32).to(dev())
    item_i = (torch.arange(Mi) % 3).to(torch.int32).to(dev())
    item_t = (torch.arange(Mt) % 3).to(torch.int32).to(dev())
    A, Wi, Wt = ops.w_to_k32_blocked(g_(a)), ops.w_to_k32_blocked(g_(wi)), ops.w_to_k32_blocked(g_(wt))
    ws = torch.empty(8 * (Mi + Mt) * N, dtype=torch.float32, device=dev())
    res_i, res_t = g_(rnd((Mi, N), 36)), g_(rnd((Mt, N), 37))

    def run(epi, splitk):
        oi = res_i.clone() if epi == ops.EPI_BIAS_GATE_RES else torch.zeros(Mi, N, dtype=BF16, device=dev())
        ot = res_t.clone() if epi == ops.EPI_BIAS_GATE_RES else torch.zeros(Mt, N, dtype=BF16, device=dev())
        kw_i = dict(res=oi, gate=gate, gate_item_stride=N, row_item_map=item_i) if epi == ops.EPI_BIAS_GATE_RES else {}
        kw_t = dict(res=ot, gate=gate, gate_item_stride=N, row_item_map=item_t) if epi == ops.EPI_BIAS_GATE_RES else {}
        ops.gemm([ops.GemmGroupArgs(A, Wi, g_(b), oi, a_row_map=map_i, a_k32_blocked=True, **kw_i),
                  ops.GemmGroupArgs(A, Wt, g_(b), ot, a_row_map=map_t, a_k32_blocked=True, **kw_t)], epi, w_k32_blocked=True,
                 splitk_ws=ws if splitk else None)
        torch.cuda.synchronize()
        return oi, ot

    for epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU_TANH, ops.EPI_BIAS_GATE_RES):
        ws.fill_(float("nan"))                                  # every partial the finish reads must have been written
        plain, split, again = run(epi, False), run(epi, True), run(epi, True)
        for x, y, z in zip(plain, split, again):
            assert torch.equal(y, z), "split-K is not deterministic"
            d = (x.float() - y.float()).abs()
            assert torch.isfinite(y.float()).all()
            assert float((d > 0).float().mean()) <= 0.15 and float(d.norm() / x.float().norm()) <= 2e-3
    ref = a[map_i.long().cpu()] @ wi.t() + b
    assert rel_l2(run(ops.EPI_BIAS, True)[0], ref) <= 4e-3
    assert not bool(torch.isnan(ws[: 2 * (Mi + Mt) * N]).any())   # at least two splits were written: the split path did run


# ---------------------------------------------------------------------------------------------------------------------
# fp8 (BASELINE.json config 5): dynamic per-row e4m3 quantisation + the scaled-MFMA GEMM (ABI v7)
def _e4m3_ref(x: torch.Tensor):
    """Reference of omni_quantize_fp8_rows in torch: per-row scale = amax / 448, round-to-nearest-even e4m3fn."""
    amax = x.float().abs().amax(dim=1).clamp_min(1e-12)
    sc = amax * (1.0 / 448.0)
    q = (x.float() * (1.0 / sc)[:, None]).to(torch.float8_e4m3fn)
    return q, sc


@pytest.mark.parametrize("blocked_in", [False, True])
@pytest.mark.parametrize("rows,K", [(700, 3072), (130, 12288), (5, 64)])
def test_quantize_fp8_rows_matches_torch_e4m3(rows, K, blocked_in):
    from vllm_omni_amd import ops

    x = rnd((rows, K), 41, 3.0)
    x[3 % rows, 17 % K] = 250.0                                           # an outlier sets its row's scale
    xin = ops.w_to_k32_blocked(g_(x)) if blocked_in else g_(x)
    y8, sc = ops.quantize_fp8_rows(xin, x_k32_blocked=blocked_in)
    torch.cuda.synchronize()
    q_ref, sc_ref = _e4m3_ref(bf16_round(x))
    assert torch.allclose(sc.cpu(), sc_ref, rtol=1e-6, atol=0)
    got = ops.k64_blocked_fp8_to_rows(y8).cpu()
    same = (got.view(torch.uint8) == q_ref.view(torch.uint8)).float().mean()
    assert same >= 0.999, same                                            # (1/scale as a multiply: rare 1-ulp ties)
    assert rel_l2(got.float() * sc.cpu()[:, None], bf16_round(x)) <= 4e-2  # e4m3: 3 mantissa bits


@pytest.mark.parametrize("epi", ["bias", "gelu", "gate_res", "split3_qknorm_rope"])
def test_gemm_fp8_scaled_mfma_equals_the_dequantised_product(epi):
    """omni_gemm_params.fp8: e4m3 operands through v_mfma_scale_f32_16x16x128_f8f6f4.  The checker multiplies the DEQUANTISED
    operands in fp32 (products of fp8 values are exact in fp32, so the kernel must agree to fp32 summation order + the one bf16
    rounding of the output), which pins the operand layout, the scales and every epilogue; the distance to the UNQUANTISED
    product (what fp8 costs) is printed and bounded separately."""
    from vllm_omni_amd import ops

    Mi, Mt, N, K = 700, 130, 768, 1024
    a_i, a_t = rnd((Mi, K), 51), rnd((Mt, K), 52)
    wi, wt, b = rnd((N, K), 53, 0.05), rnd((N, K), 54, 0.05), rnd((N,), 55, 0.5)
    Ai8, sai = ops.quantize_fp8_rows(g_(a_i))
    At8, sat = ops.quantize_fp8_rows(g_(a_t))
    Wi8, swi = ops.quantize_fp8_rows(g_(wi))
    Wt8, swt = ops.quantize_fp8_rows(g_(wt))
    deq = lambda y8, sc: ops.k64_blocked_fp8_to_rows(y8).float().cpu() * sc.cpu()[:, None]       # noqa: E731
    ref_i = deq(Ai8, sai) @ deq(Wi8, swi).t() + b
    ref_t = deq(At8, sat) @ deq(Wt8, swt).t() + b
    full_i = a_i @ wi.t() + b
    kw_i, kw_t, e = {}, {}, ops.EPI_BIAS
    oi = torch.zeros(Mi, N, dtype=BF16, device=dev())
    ot = torch.zeros(Mt, N, dtype=BF16, device=dev())
    if epi == "gelu":
        e = ops.EPI_BIAS_GELU_TANH
        gelu = lambda t: torch.nn.functional.gelu(t, approximate="tanh")                          # noqa: E731
        ref_i, ref_t, full_i = gelu(ref_i), gelu(ref_t), gelu(full_i)
    elif epi == "gate_res":
        e = ops.EPI_BIAS_GATE_RES
        res_i, res_t, gate = rnd((Mi, N), 56), rnd((Mt, N), 57), rnd((3, N), 58)
        oi, ot = g_(res_i), g_(res_t)
        item_i, item_t = torch.arange(Mi) % 3, torch.arange(Mt) % 3
        kw_i = dict(res=oi, gate=g_(gate), gate_item_stride=N, row_item_map=item_i.to(torch.int32).to(dev()))
        kw_t = dict(res=ot, gate=g_(gate), gate_item_stride=N, row_item_map=item_t.to(torch.int32).to(dev()))
        ref_i = res_i + gate[item_i] * bf16_round(ref_i)
        ref_t = res_t + gate[item_t] * bf16_round(ref_t)
        full_i = res_i + gate[item_i] * full_i
    if epi == "split3_qknorm_rope":
        # the fused QKV epilogue needs its tables: compare the fp8 launch with the bf16 launch of the SAME epilogue on the
        # dequantised operands instead (bit-for-bit the same epilogue code behind different main loops)
        from vllm_omni_amd.diffusion.models.qwen_image.rope import rope_table

        D = 256
        N = 3 * D
        wq, bq = rnd((N, K), 59, 0.05), rnd((N,), 60, 0.5)
        Wq8, swq = ops.quantize_fp8_rows(g_(wq))
        cos, sin = rope_table((1, 26, 32), 0)
        nw = [(1 + rnd((128,), 61 + i, 0.1)).to(BF16).to(dev()) for i in range(2)]
        pos = (torch.arange(Mi) % cos.shape[0]).to(torch.int32).to(dev())
        outs = []
        for fp8 in (True, False):
            q, k, v = (torch.zeros(Mi, D, dtype=BF16, device=dev()) for _ in range(3))
            A = Ai8 if fp8 else ops.w_to_k32_blocked(deq(Ai8, sai).to(BF16).to(dev()))
            W = Wq8 if fp8 else ops.w_to_k32_blocked(deq(Wq8, swq).to(BF16).to(dev()))
            ops.gemm([ops.GemmGroupArgs(A, W, g_(bq), q, out1=k, out2=v, a_k32_blocked=True, qk_norm_q_w=nw[0], qk_norm_k_w=nw[1],
                                        qk_rope_cos=cos.to(dev(), BF16), qk_rope_sin=sin.to(dev(), BF16), qk_row_pos=pos,
                                        a_scale=sai if fp8 else None, w_scale=swq if fp8 else None)],
                     ops.EPI_BIAS_SPLIT3_QKNORM_ROPE, split_n=D, w_k32_blocked=True, fp8=fp8)
            torch.cuda.synchronize()
            outs.append((q, k, v))
        for x8, x16 in zip(*outs):
            assert rel_l2(x8, x16.float().cpu()) <= 6e-3          # bf16-rounded dequantised operands on the other side
        return
    ops.gemm([ops.GemmGroupArgs(Ai8, Wi8, g_(b), oi, a_k32_blocked=True, a_scale=sai, w_scale=swi, **kw_i),
              ops.GemmGroupArgs(At8, Wt8, g_(b), ot, a_k32_blocked=True, a_scale=sat, w_scale=swt, **kw_t)], e,
             w_k32_blocked=True, fp8=True)
    torch.cuda.synchronize()
    r_i, r_t, r_full = rel_l2(oi, ref_i), rel_l2(ot, ref_t), rel_l2(oi, full_i)
    print(f"fp8 gemm [{epi}]: vs dequantised fp32 product {r_i:.2e} / {r_t:.2e}; vs the unquantised product {r_full:.2e}")
    assert r_i <= 4e-3 and r_t <= 4e-3
    assert r_full <= 6e-2                                         # two e4m3 operands: ~2^-4 relative per element, averaged over K


def test_adaln_modulate_fp8_fused_equals_adaln_then_quantize():
    """omni_adaln_modulate_fp8 (the fp8 mode's AdaLN: quantisation fused into the pass that has the row in registers) returns
    the same bytes and scales as omni_adaln_modulate_ex followed by omni_quantize_fp8_rows, and the same bf16 copy."""
    from vllm_omni_amd import ops

    rows, D, items = 300, 3072, 3
    x = g_(rnd((rows, D), 71, 2.0))
    mod = g_(rnd((items, 2 * D), 72, 0.3))
    item = (torch.arange(rows) % items).to(torch.int32).to(dev())
    y = ops.adaln_modulate(x, mod[:, D:], mod[:, :D], mod_item_stride=2 * D, row_item_map=item, out_k32_blocked=True)
    y8_ref, sc_ref = ops.quantize_fp8_rows(y, x_k32_blocked=True)
    y8, sc, yb = ops.adaln_modulate_fp8(x, mod[:, D:], mod[:, :D], mod_item_stride=2 * D, row_item_map=item, want_bf16=True)
    y8b, scb, none = ops.adaln_modulate_fp8(x, mod[:, D:], mod[:, :D], mod_item_stride=2 * D, row_item_map=item)
    torch.cuda.synchronize()
    assert none is None and torch.equal(yb, y)
    assert torch.equal(sc, sc_ref) and torch.equal(y8, y8_ref) and torch.equal(y8b, y8_ref) and torch.equal(scb, sc_ref)


def test_torch_sdpa_backend_refuses_device_tensors():
    """Round-4 verdict (hygiene): DIFFUSION_ATTENTION_BACKEND=TORCH_SDPA must not swap the HIP attention kernel for a torch op on
    a GPU box — the host backend serves CPU tensors (plug-in surface tests) and raises for device tensors."""
    from vllm_omni_amd import _native
    from vllm_omni_amd.diffusion.attention.backends.sdpa import SDPAImpl

    impl = SDPAImpl(num_heads=2, head_size=128, softmax_scale=128 ** -0.5)
    q = torch.randn(1, 16, 2, 128, device=dev(), dtype=torch.bfloat16)
    with pytest.raises(_native.OmniNativeError, match="CDNA4_FLASH"):
        impl.forward(q, q, q)
    out = impl.forward(q.cpu().float(), q.cpu().float(), q.cpu().float())      # host tensors: served
    assert out.shape == (1, 16, 2, 128)
