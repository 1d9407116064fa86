"""GPU: the two small-batch launch savers of round 6 (ABI v12).

1. In-launch split-K reduce (csrc/gemm.hip `gemm_bf16_pp_kernel<EPI, 3>`, opt-in: kernel_hint OMNI_GEMM_KERNEL_SPLITK_IN_LAUNCH —
   measured, it pays only for 2-way splits on nearly full grids): a whole-launch split-K whose nsplit workgroups per tile
   wait for each other inside the launch (write-through partial stores, arrival counter, one agent-scope acquire) and then run
   the epilogue from the partials over their share of the tile's rows — no `gemm_splitk_finish_kernel` launch.  The sum over the
   splits is formed in split order by the same code as the finish kernel's, so the result must be BIT-IDENTICAL to the
   two-kernel path (the default; also taken when the workspace has no room for the counters behind the partials), for every epilogue; the
   counters must be zero again after every launch; thousands of launches with a second stream contending for the CUs (the
   workgroups of a tile then become resident at different times) must neither hang nor change a bit.
2. `omni_adaln_modulate_pair`: the image stream's and the text stream's AdaLN in one launch == the two single launches.
"""
import math

import pytest
import torch

from _util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"
CNT = 512                                      # csrc/gemm.hip SPLITK_CNT_INTS: counter words behind the partials


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16_round(torch.randn(shape, generator=g) * scale)


def g_(t):
    return t.to(DEV, BF16).contiguous()


def split_factor(tiles: int, K: int, cus: int = 256) -> int:
    """csrc/gemm.hip splitk_factor(): the largest s in {8, 6, 4, 3, 2} dividing the K-tile count with >= 4 K-tiles per piece and
    tiles * s within one round of the CUs."""
    nkt = K // 64
    for s in (8, 6, 4, 3, 2):
        if nkt % s == 0 and nkt // s >= 4 and tiles * s <= cus:
            return s
    return 1


class Case:
    """Two groups (ragged M: a full + a partial image row tile, a half-empty text row tile), gathered K32-blocked A."""

    def __init__(self, K, N, Mi=400, Mt=75):
        from vllm_omni_amd import ops

        self.K, self.N, self.Mi, self.Mt = K, N, Mi, Mt
        R = Mi + Mt + 60
        self.a = rnd((R, K), 31)
        self.wi, self.wt, self.b = rnd((N, K), 32, 0.03), rnd((N, K), 33, 0.03), rnd((N,), 34, 0.5)
        self.gate = g_(rnd((3, N), 35))
        gi = torch.Generator().manual_seed(6)
        self.map_i = torch.randperm(R, generator=gi)[:Mi].to(torch.int32).to(DEV)
        self.map_t = torch.randperm(R, generator=gi)[:Mt].to(torch.int32).to(DEV)
        self.item_i = (torch.arange(Mi) % 3).to(torch.int32).to(DEV)
        self.item_t = (torch.arange(Mt) % 3).to(torch.int32).to(DEV)
        self.A, self.Wi, self.Wt = ops.w_to_k32_blocked(g_(self.a)), ops.w_to_k32_blocked(g_(self.wi)), ops.w_to_k32_blocked(g_(self.wt))
        self.bd = g_(self.b)
        self.res_i, self.res_t = g_(rnd((Mi, N), 36)), g_(rnd((Mt, N), 37))
        tiles = (-(-Mi // 256) + -(-Mt // 256)) * (N // 256)
        self.tiles, self.s = tiles, split_factor(tiles, K)
        assert self.s > 1, "pick a shape that splits"
        self.partials = self.s * (Mi + Mt) * N
        self.ws_two_kernel = torch.empty(self.partials, dtype=torch.float32, device=DEV)            # no room for the counters
        self.ws_in_launch = torch.empty(self.partials + CNT, dtype=torch.float32, device=DEV)

    def run(self, epi, ws, out=None, in_launch=None):
        from vllm_omni_amd import ops

        if in_launch is None:
            in_launch = ws is self.ws_in_launch
        gate_res = epi == ops.EPI_BIAS_GATE_RES
        if out is None:
            oi = self.res_i.clone() if gate_res else torch.zeros(self.Mi, self.N, dtype=BF16, device=DEV)
            ot = self.res_t.clone() if gate_res else torch.zeros(self.Mt, self.N, dtype=BF16, device=DEV)
        else:
            oi, ot = out
            if gate_res:
                oi.copy_(self.res_i)
                ot.copy_(self.res_t)
        kw_i = dict(res=oi, gate=self.gate, gate_item_stride=self.N, row_item_map=self.item_i) if gate_res else {}
        kw_t = dict(res=ot, gate=self.gate, gate_item_stride=self.N, row_item_map=self.item_t) if gate_res else {}
        ops.gemm([ops.GemmGroupArgs(self.A, self.Wi, self.bd, oi, a_row_map=self.map_i, a_k32_blocked=True, **kw_i),
                  ops.GemmGroupArgs(self.A, self.Wt, self.bd, ot, a_row_map=self.map_t, a_k32_blocked=True, **kw_t)], epi,
                 w_k32_blocked=True, splitk_ws=ws, kernel_hint=ops.GEMM_KERNEL_SPLITK_IN_LAUNCH if in_launch else 0)
        return oi, ot


@pytest.mark.parametrize("K,N", [(3072, 3072), (12288, 3072), (3072, 9216), (1024, 256)])
def test_in_launch_reduce_is_bit_identical_to_the_two_kernel_split_k(K, N):
    from vllm_omni_amd import ops

    c = Case(K, N)
    for epi in (ops.EPI_BIAS, ops.EPI_BIAS_GELU_TANH, ops.EPI_BIAS_GATE_RES):
        c.ws_two_kernel.fill_(float("nan"))
        c.ws_in_launch.fill_(float("nan"))                   # poisoned counters too: a standalone call zeroes them itself
        two = c.run(epi, c.ws_two_kernel)
        one = c.run(epi, c.ws_in_launch)
        again = c.run(epi, c.ws_in_launch)
        plain = c.run(epi, None)
        dflt = c.run(epi, c.ws_in_launch, in_launch=False)   # room for the counters, no hint: the default stays the two-kernel path
        short = c.run(epi, c.ws_two_kernel, in_launch=True)  # the hint without room for the counters: the two-kernel path
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) and torch.equal(a, c_) for a, b, c_ in zip(two, dflt, short))
        for x, y, z, p in zip(two, one, again, plain):
            assert torch.isfinite(y.float()).all()
            assert torch.equal(x, y), f"in-launch reduce differs from the finish kernel (epilogue {epi})"
            assert torch.equal(y, z)
            assert float((p.float() - y.float()).norm() / p.float().norm()) <= 2e-3      # vs the unsplit kernel: summation order
        cnt = c.ws_in_launch[c.partials:].view(torch.int32)
        assert int(cnt.abs().sum()) == 0, "arrival / departure counters (or the time-out flag) are not zero after the launch"
        assert not bool(torch.isnan(c.ws_in_launch[: c.partials]).any())   # every split wrote its partial: the split path did run
    ref = c.a[c.map_i.long().cpu()] @ c.wi.t() + c.b
    assert rel_l2(c.run(ops.EPI_BIAS, c.ws_in_launch)[0], ref) <= 4e-3


def test_in_launch_reduce_with_the_fused_qkv_epilogue():
    """The QKV launch of a 256x256 CFG pair: split-3 scatter into the joint q / k / v + per-head RMSNorm + RoPE from partials."""
    from vllm_omni_amd import ops
    from vllm_omni_amd.diffusion.models.qwen_image.rope import rope_table

    H, D, K = 24, 3072, 3072
    T, grid = 64, (1, 16, 16)
    Mi, Mt = 2 * grid[1] * grid[2], 2 * T                    # two items (a CFG pair): 512 image rows, 128 text rows
    rows = Mi + Mt
    xi, xt = rnd((Mi, K), 1), rnd((Mt, K), 2)
    wi, wt, bi, bt = rnd((3 * D, K), 3, 0.02), rnd((3 * D, K), 4, 0.02), rnd((3 * D,), 5), rnd((3 * D,), 6)
    nw = [g_(bf16_round(rnd((128,), 10 + i, 0.2) + 1)) for i in range(4)]
    cos, sin = rope_table(grid, T)
    cosb, sinb = g_(bf16_round(cos)), g_(bf16_round(sin))
    pos_item = torch.arange(T + Mi // 2, dtype=torch.int32)            # table rows of one item: text first, then image
    # joint order [text_0 ; image_0 ; text_1 ; image_1]
    per = T + Mi // 2
    mt = torch.cat([torch.arange(T), per + torch.arange(T)]).to(torch.int32)
    mi = torch.cat([T + torch.arange(Mi // 2), per + T + torch.arange(Mi // 2)]).to(torch.int32)
    pos_t = torch.cat([pos_item[:T], pos_item[:T]])
    pos_i = torch.cat([pos_item[T:], pos_item[T:]])
    blk = ops.w_to_k32_blocked
    Ai, At, Wi, Wt = blk(g_(xi)), blk(g_(xt)), blk(g_(wi)), blk(g_(wt))
    tiles = (2 + 1) * (3 * D // 256)
    s = split_factor(tiles, K)
    assert s == 2
    partials = s * rows * 3 * D
    outs = []
    for room in (0, CNT):
        ws = torch.full((partials + room,), float("nan"), dtype=torch.float32, device=DEV)
        q = torch.zeros(rows, D, dtype=BF16, device=DEV)
        k, v = torch.zeros_like(q), torch.zeros_like(q)
        kw_i = dict(qk_norm_q_w=nw[0], qk_norm_k_w=nw[1], qk_rope_cos=cosb, qk_rope_sin=sinb, qk_row_pos=pos_i.to(DEV))
        kw_t = dict(qk_norm_q_w=nw[2], qk_norm_k_w=nw[3], qk_rope_cos=cosb, qk_rope_sin=sinb, qk_row_pos=pos_t.to(DEV))
        ops.gemm([ops.GemmGroupArgs(Ai, Wi, g_(bi), q, out1=k, out2=v, out_row_map=mi.to(DEV), a_k32_blocked=True, **kw_i),
                  ops.GemmGroupArgs(At, Wt, g_(bt), q, out1=k, out2=v, out_row_map=mt.to(DEV), a_k32_blocked=True, **kw_t)],
                 ops.EPI_BIAS_SPLIT3_QKNORM_ROPE, split_n=D, w_k32_blocked=True, splitk_ws=ws,
                 kernel_hint=ops.GEMM_KERNEL_SPLITK_IN_LAUNCH if room else 0)
        torch.cuda.synchronize()
        if room:
            assert int(ws[partials:].view(torch.int32).abs().sum()) == 0
        outs.append((q, k, v))
    for a, b in zip(*outs):
        assert torch.isfinite(a.float()).all() and float(a.float().abs().sum()) > 0
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape", ["out_proj_gate_res", "mlp_down_gate_res", "qkv_bias"])
def test_in_launch_reduce_is_bit_stable_over_2000_contended_launches(shape):
    """A second stream keeps taking CUs away (attention + copies + idle gaps): the workgroups of a tile are then dispatched at
    different times and the early ones poll while the late ones have not started — the launch must finish every time, with the
    bits of the two-kernel path, and leave its counters zero."""
    from vllm_omni_amd import ops

    D = 3072
    N, K, epi = {"out_proj_gate_res": (D, D, ops.EPI_BIAS_GATE_RES), "mlp_down_gate_res": (D, 4 * D, ops.EPI_BIAS_GATE_RES),
                 "qkv_bias": (3 * D, D, ops.EPI_BIAS)}[shape]
    c = Case(K, N, Mi=512, Mt=128)
    want_i, want_t = c.run(epi, c.ws_two_kernel)
    torch.cuda.synchronize()
    want = int(torch.sum(want_i.view(torch.int32), dtype=torch.int64)) + int(torch.sum(want_t.view(torch.int32), dtype=torch.int64))
    LAUNCHES = 2000
    dig = torch.zeros(LAUNCHES, dtype=torch.int64, device=DEV)
    H, S = 24, 2112
    g = torch.Generator(device=DEV).manual_seed(99)
    q, k, v = (torch.randn(2 * S, H * 128, device=DEV, generator=g).to(BF16) for _ in range(3))
    cu = torch.tensor([0, S, 2 * S], dtype=torch.int32, device=DEV)
    o = torch.empty_like(q)
    src = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    dst = torch.empty_like(src)
    side = torch.cuda.Stream()
    out = (torch.empty_like(want_i), torch.empty_like(want_t))
    for i in range(LAUNCHES):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                ops.flash_attn_varlen(q, k, v, cu, H, S, 1 / math.sqrt(128), out=o)
                dst.copy_(src)
                if i % 21 == 0:
                    torch.cuda._sleep(100_000)
        oi, ot = c.run(epi, c.ws_in_launch, out)
        dig[i] = torch.sum(oi.view(torch.int32), dtype=torch.int64) + torch.sum(ot.view(torch.int32), dtype=torch.int64)
    torch.cuda.synchronize()
    bad = (dig.cpu() != want).nonzero().flatten().tolist()
    assert not bad, f"{len(bad)} of {LAUNCHES} contended launches differ from the two-kernel path (first: launch {bad[0]})"
    assert torch.equal(out[0], want_i) and torch.equal(out[1], want_t)
    assert int(c.ws_in_launch[c.partials:].view(torch.int32).abs().sum()) == 0


def test_in_launch_reduce_replays_inside_a_hip_graph():
    """The forward of a small image is captured into a hipGraph (pipeline): the memset node + the launch replay correctly."""
    from vllm_omni_amd import ops

    c = Case(3072, 3072, Mi=512, Mt=128)
    want = c.run(ops.EPI_BIAS_GATE_RES, c.ws_two_kernel)
    out = (torch.empty_like(want[0]), torch.empty_like(want[1]))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        c.run(ops.EPI_BIAS_GATE_RES, c.ws_in_launch, out)        # warm-up outside the capture
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            c.run(ops.EPI_BIAS_GATE_RES, c.ws_in_launch, out)
        for _ in range(20):
            out[0].zero_()
            gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])


@pytest.mark.parametrize("mode", ["row_major", "k32_blocked", "fp8", "fp8_with_bf16_copy"])
def test_adaln_pair_equals_the_two_single_launches(mode):
    from vllm_omni_amd import ops

    D, Ri, Rt, items = 3072, 515, 131, 3                    # (not multiples of the four rows a workgroup holds)
    xi, xt = g_(rnd((Ri, D), 1, 3.0) + 0.5), g_(rnd((Rt, D), 2, 2.0))
    mod_i, mod_t = g_(rnd((items, 6 * D), 3, 0.5)), g_(rnd((items, 6 * D), 4, 0.5))
    it_i = (torch.arange(Ri) % items).to(torch.int32).to(DEV)
    it_t = (torch.arange(Rt) % items).to(torch.int32).to(DEV)
    streams = [(xi, mod_i[:, 4 * D:], mod_i[:, 3 * D:], it_i), (xt, mod_t[:, 4 * D:], mod_t[:, 3 * D:], it_t)]
    kw = dict(mod_item_stride=6 * D)
    if mode in ("row_major", "k32_blocked"):
        blocked = mode == "k32_blocked"
        got = ops.adaln_modulate_pair(streams, out_k32_blocked=blocked, **kw)
        want = [ops.adaln_modulate(x, sc, sh, row_item_map=m, out_k32_blocked=blocked, **kw) for x, sc, sh, m in streams]
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    else:
        copy = mode == "fp8_with_bf16_copy"
        got = ops.adaln_modulate_pair(streams, fp8=True, want_bf16=copy, **kw)
        want = [ops.adaln_modulate_fp8(x, sc, sh, row_item_map=m, want_bf16=copy, **kw) for x, sc, sh, m in streams]
        torch.cuda.synchronize()
        for (y8, s8, y), (w8, ws8, wy) in zip(got, want):
            assert torch.equal(y8, w8) and torch.equal(s8, ws8)
            assert (y is None and wy is None) or torch.equal(y, wy)
