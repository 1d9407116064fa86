"""GPU: the two "thin last round" splits of round 6 (ABI v11) against the unsplit kernels and the fp32 oracle.

  * GEMM TAIL SPLIT (csrc/gemm.hip, gemm_bf16_pp_kernel<EPI, 2> + gemm_tail_finish_kernel): a launch of more than one round of
    256x256 tiles whose last round is thin runs the full rounds' tiles unsplit — BIT-IDENTICAL to a launch without workspace —
    and the tail tiles split along K (fp32 partials, summed in split order: deterministic; equal to the unsplit kernel up to
    isolated one-ulp bf16 flips).  Shapes: the N = 3072 GEMMs of one 2048^2 request scaled down in M only.
  * ATTENTION split of the short last q-block (csrc/attention_w64.hip + attn_split_combine_kernel): full 256-query blocks
    bit-identical, the <= 64 remainder rows merged from key-range partials.
Tolerances: vs the unsplit kernel rel_l2 <= 2e-3 on the affected GEMM tiles (4e-3 on the attention rows: both sides round P to
bf16 against their own running max) and exact elsewhere; vs the fp32 oracle the usual
GEMM / attention bar rel_l2 <= 4e-3."""
import math

import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return bf16_round(torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale)


def g_(t):
    return t.to(DEV, BF16).contiguous()


def _cus():
    return torch.cuda.get_device_properties(0).multi_processor_count


@pytest.mark.parametrize("K,epi", [(3072, "gate_res"), (12288, "gate_res"), (3072, "gelu_blocked"), (3072, "bias")])
def test_gemm_tail_split_full_rounds_bit_identical_tail_tiles_close(K, epi):
    from vllm_omni_amd import ops

    cus = _cus()
    N = 3072                                              # 12 column tiles
    mt_img = (cus + 11) // 12 + 1                         # just over one round of tiles, like 6 rounds + 12 at 2048^2
    Mi, Mt = mt_img * 256 - 100, 128                      # ragged image rows + one half-empty text row tile (its own weights)
    tiles = (mt_img + 1) * 12
    assert cus < tiles < 2 * cus and (tiles % cus) * 2 <= cus
    a_i, a_t = rnd((Mi, K), 1), rnd((Mt, K), 2)
    w_i, w_t, b = rnd((N, K), 3, 0.03), rnd((N, K), 4, 0.03), rnd((N,), 5, 0.5)
    gate = g_(rnd((2, N), 6))
    item_i = (torch.arange(Mi) % 2).to(torch.int32).to(DEV)
    item_t = (torch.arange(Mt) % 2).to(torch.int32).to(DEV)
    Ai, At, Wi, Wt = (ops.w_to_k32_blocked(g_(x)) for x in (a_i, a_t, w_i, w_t))
    res_i, res_t = g_(rnd((Mi, N), 7)), g_(rnd((Mt, N), 8))
    ws = torch.empty(512 * 256 * 256, dtype=torch.float32, device=DEV)

    def run(workspace, hint=0):
        if epi == "gate_res":
            oi, ot = res_i.clone(), res_t.clone()
            kw_i = dict(res=oi, gate=gate, gate_item_stride=N, row_item_map=item_i)
            kw_t = dict(res=ot, gate=gate, gate_item_stride=N, row_item_map=item_t)
            e = ops.EPI_BIAS_GATE_RES
        else:
            oi, ot = torch.zeros(Mi, N, dtype=BF16, device=DEV), torch.zeros(Mt, N, dtype=BF16, device=DEV)
            kw_i = kw_t = dict(out_k32_blocked=True) if epi == "gelu_blocked" else {}
            e = ops.EPI_BIAS_GELU_TANH if epi == "gelu_blocked" else ops.EPI_BIAS
        ops.gemm([ops.GemmGroupArgs(Ai, Wi, g_(b), oi, a_k32_blocked=True, **kw_i),
                  ops.GemmGroupArgs(At, Wt, g_(b), ot, a_k32_blocked=True, **kw_t)], e, w_k32_blocked=True, splitk_ws=workspace,
                 kernel_hint=hint)
        torch.cuda.synchronize()
        return torch.cat([oi, ot])

    ws.fill_(float("nan"))
    plain, split, again = run(None), run(ws), run(ws)
    off = run(ws, ops.GEMM_KERNEL_NO_TAIL_SPLIT)
    assert torch.equal(off, plain), "OMNI_GEMM_KERNEL_NO_TAIL_SPLIT must give the unsplit launch"
    assert torch.equal(split, again), "the tail split is not deterministic"
    assert torch.isfinite(split.float()).all()
    assert not bool(torch.isnan(ws[:2 * 256 * 256]).any()), "no partial was written: the tail split did not run"
    d = (plain.float() - split.float()).abs()
    if epi == "gelu_blocked":                              # K32-blocked output [N/32][rows][32]: compare element sets, not tiles
        changed = float((d > 0).float().mean())
        assert changed <= 0.15 * (tiles % cus) / tiles + 1e-4
    else:
        # a 256x256 tile is either untouched (bit-identical: it ran in a full round) or a tail tile (few one-ulp flips)
        rows = plain.shape[0]
        touched = 0
        for r0 in list(range(0, Mi, 256)) + [Mi]:
            r1 = min(r0 + 256, Mi) if r0 < Mi else rows
            for c0 in range(0, N, 256):
                blk = d[r0:r1, c0:c0 + 256]
                if float(blk.max()) > 0:
                    touched += 1
                    assert float((blk > 0).float().mean()) <= 0.15
        assert 0 < touched <= tiles % cus, (touched, tiles % cus)
    assert float(d.norm() / plain.float().norm()) <= 2e-3
    if epi == "bias":
        ref = torch.cat([a_i @ w_i.t() + b, a_t @ w_t.t() + b])
        assert rel_l2(split, ref) <= 4e-3


def _attn_ref(q, k, v, lens, H):
    outs, o = [], 0
    for n in lens:
        sl = slice(o, o + n)
        outs.append(O.sdpa_nhd(q[sl].reshape(1, n, H, 128), k[sl].reshape(1, n, H, 128), v[sl].reshape(1, n, H, 128),
                               1 / math.sqrt(128)).reshape(n, H * 128))
        o += n
    return torch.cat(outs)


@pytest.mark.parametrize("lens,H,blocked", [([4160, 4160], 24, False), ([4160, 4115], 24, True), ([2112, 2065, 2100, 2111], 16, False)])
def test_attention_short_last_block_split_matches_unsplit_and_oracle(lens, H, blocked):
    """One 1024^2 true-CFG request (2 x 24 heads x (16 full q-blocks + 64 rows): 768 full workgroups = 3 rounds + 48 short ones),
    a ragged pair (text 64 / 19), and four shorter items whose remainders differ."""
    from vllm_omni_amd import ops

    rows = sum(lens)
    q, k, v = rnd((rows, H * 128), 11, 0.3), rnd((rows, H * 128), 12, 0.3), rnd((rows, H * 128), 13)
    k[lens[0] - 30, :128] = bf16_round(q[lens[0] - 5, :128] * 30.0)     # a late spike for a query of the short block (head 0), sized
    #                                                                       like tests/test_gpu_ops.py's rescale-spike cases
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    B = len(lens)
    ws = torch.full((ops.flash_attn_workspace_floats(B, H),), float("nan"), dtype=torch.float32, device=DEV)
    Q, K, V = g_(q), g_(k), g_(v)
    plain = ops.flash_attn_varlen(Q, K, V, cu, H, max(lens), 1 / math.sqrt(128), out_k32_blocked=blocked)
    split = ops.flash_attn_varlen(Q, K, V, cu, H, max(lens), 1 / math.sqrt(128), out_k32_blocked=blocked, workspace=ws)
    again = ops.flash_attn_varlen(Q, K, V, cu, H, max(lens), 1 / math.sqrt(128), out_k32_blocked=blocked, workspace=ws)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(ws[:64 * 128]).any()), "no partial was written: the split path did not run"
    assert torch.equal(split, again) and torch.isfinite(split.float()).all()
    if blocked:                                                            # back to row-major for the row-wise comparison
        unblk = lambda t: t.view(H * 4, rows, 32).permute(1, 0, 2).reshape(rows, H * 128)   # noqa: E731
        plain, split = unblk(plain), unblk(split)
    qfull = max(lens) // 256
    o = 0
    for n in lens:
        full_end = min(n, qfull * 256)
        assert torch.equal(split[o:o + full_end], plain[o:o + full_end]), "full q-blocks must be bit-identical"
        if n > full_end:
            assert rel_l2(split[o + full_end:o + n], plain[o + full_end:o + n]) <= 4e-3
        o += n
    ref = _attn_ref(q, k, v, lens, H)
    assert rel_l2(split, ref) <= 4e-3 and (split.float().cpu() - ref).abs().max() <= 2e-2


def test_forward_at_one_round_plus_thin_tail_mlp_up():
    """omni_dit_forward hands MLP-up (N = 4 D, 48 column tiles) the split-K workspace when a thin tail follows one to three full
    rounds (csrc/dit_forward.hip mlp_up_tail_split): one 384x384 CFG pair = 1152 + 128 rows = 6 row tiles x 48 = 288 tiles = one
    round + 32, whose 32 tail tiles then run 4-way K-split.  Two full-width layers against the fp32 oracle at the usual forward
    bar, determinism, and — the rule looks at tile counts — a 256x256 pair (240 tiles: no tail) for contrast at the same bar."""
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    torch.backends.cuda.matmul.allow_tf32 = False
    if _cus() != 256:
        pytest.skip("the shapes below are chosen for 256 CUs")
    m = QwenImageTransformer2DModel(num_layers=2, device=DEV)
    m.init_random_(seed=91)
    P = {n: p.detach().float() for n, p in m.named_parameters()}          # before the first forward (row-major layout)
    for hw in (24, 16):                                                    # 384^2 (the tail-split shape), 256^2
        B, S, T = 2, hw * hw, 64
        g = torch.Generator(device=DEV).manual_seed(hw)
        lat = torch.randn(B, S, 64, device=DEV, generator=g).to(BF16)
        txt = torch.randn(B, T, 3584, device=DEV, generator=g).to(BF16)
        sig = torch.full((B,), 0.6015625, device=DEV)
        kw = dict(hidden_states=lat, encoder_hidden_states=txt, timestep=sig, img_shapes=[[(1, hw, hw)]] * B, txt_seq_lens=[T] * B,
                  return_dict=False)
        out = m(**kw)[0].clone()
        again = m(**kw)[0]
        torch.cuda.synchronize()
        assert torch.equal(out, again)
        with torch.no_grad():
            for i in range(B):
                ref = O.dit_forward(P, lat[i:i + 1].float(), txt[i:i + 1].float(), sig[i:i + 1], (1, hw, hw), num_heads=24)
                r = rel_l2(out[i:i + 1], ref)
                assert r <= 1e-2, (hw, i, r)
