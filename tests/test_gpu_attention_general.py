"""GPU parity of `omni_flash_attn_general` (csrc/attention_general.hip, ABI v11) and of the CDNA4_FLASH backend's dispatch:
everything the reference's `SDPAImpl.forward` hands to F.scaled_dot_product_attention (vllm_omni/diffusion/attention/backends/
sdpa.py:46-66) beyond Qwen-Image's joint self-attention — cross-attention (S_q != S_kv: wan2_2_transformer.py:243,340), head size
64 (sd3_transformer.py:108), `attn_metadata.attn_mask` (bool / additive, broadcast over batch / heads / queries), `causal=True`,
grouped K / V heads.  Checker: `O.sdpa_nhd_general` (fp32, spelled-out softmax; pinned to the torch op the reference calls in
tests/test_attention_general_host.py).  Tolerance: rel_l2 <= 4e-3, max |err| <= 2e-2 (GEMM / attention)."""
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


def impl_for(H, dh, causal=False, Hkv=None):
    from vllm_omni_amd.diffusion.attention.backends.cdna4_flash import CDNA4FlashBackend

    return CDNA4FlashBackend.get_impl_cls()(num_heads=H, head_size=dh, softmax_scale=1 / math.sqrt(dh), causal=causal, num_kv_heads=Hkv)


def check(got, ref, tol=4e-3):
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) <= tol, rel_l2(got, ref)
    assert (got.float().cpu() - ref).abs().max() <= 2e-2


@pytest.mark.parametrize("dh", [64, 128])
@pytest.mark.parametrize("B,Sq,Sk,H", [(2, 300, 77, 4), (1, 1560, 512, 12), (3, 64, 1000, 2), (2, 129, 1, 3), (1, 1, 130, 2)])
def test_cross_attention_matches_oracle(dh, B, Sq, Sk, H):
    """S_q != S_kv without a mask: key tails that are not a multiple of 64, query tails that are not a multiple of 128, one key,
    one query."""
    from vllm_omni_amd.diffusion.attention.backends.abstract import AttentionMetadata

    q, k, v = rnd((B, Sq, H, dh), 1), rnd((B, Sk, H, dh), 2), rnd((B, Sk, H, dh), 3)
    ref = O.sdpa_nhd_general(q, k, v, 1 / math.sqrt(dh))
    got = impl_for(H, dh).forward(g_(q), g_(k), g_(v), AttentionMetadata())
    assert got.shape == (B, Sq, H, dh)
    check(got, ref)


@pytest.mark.parametrize("B,S,H", [(2, 333, 24), (1, 4096 + 154, 24)])
def test_self_attention_head_size_64(B, S, H):
    """sd3_transformer.py:108 geometry (24 heads of 64): the joint self-attention at a head size the tuned kernels are not built for."""
    q, k, v = rnd((B, S, H, 64), 4), rnd((B, S, H, 64), 5), rnd((B, S, H, 64), 6)
    ref = O.sdpa_nhd_general(q, k, v, 0.125)
    check(impl_for(H, 64).forward(g_(q), g_(k), g_(v), None), ref)


@pytest.mark.parametrize("dh", [64, 128])
def test_key_padding_mask_bool_broadcast_over_heads_and_queries(dh):
    """The mask multi-prompt callers build (z_image_transformer.py:609-661: [B, S] bool, True = real token) as SDPA takes it:
    [B, 1, 1, S_k].  Equals attention over the unpadded keys."""
    from vllm_omni_amd.diffusion.attention.backends.abstract import AttentionMetadata

    B, Sq, Sk, H = 3, 200, 190, 4
    lens = [190, 77, 1]
    q, k, v = rnd((B, Sq, H, dh), 7), rnd((B, Sk, H, dh), 8), rnd((B, Sk, H, dh), 9)
    mask = torch.zeros(B, 1, 1, Sk, dtype=torch.bool)
    for i, n in enumerate(lens):
        mask[i, ..., :n] = True
    ref = O.sdpa_nhd_general(q, k, v, 1 / math.sqrt(dh), attn_mask=mask)
    for i, n in enumerate(lens):                    # the checker's own sanity: masked keys == absent keys
        alone = O.sdpa_nhd_general(q[i:i + 1], k[i:i + 1, :n], v[i:i + 1, :n], 1 / math.sqrt(dh))
        assert rel_l2(ref[i:i + 1], alone) <= 1e-6
    got = impl_for(H, dh).forward(g_(q), g_(k), g_(v), AttentionMetadata(attn_mask=mask.to(DEV)))
    check(got, ref)


@pytest.mark.parametrize("mdtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", ["bhqk", "1hqk", "qk", "b11k"])
def test_additive_masks_of_every_broadcast_shape(mdtype, shape):
    B, Sq, Sk, H, dh = 2, 150, 210, 3, 128
    q, k, v = rnd((B, Sq, H, dh), 10), rnd((B, Sk, H, dh), 11), rnd((B, Sk, H, dh), 12)
    full = {"bhqk": (B, H, Sq, Sk), "1hqk": (1, H, Sq, Sk), "qk": (Sq, Sk), "b11k": (B, 1, 1, Sk)}[shape]
    bias = (torch.randn(full, generator=torch.Generator().manual_seed(13)) * 2.0).to(mdtype)
    bias.view(-1)[::7] = float("-inf")             # additive masks carry -inf for "do not attend"
    from vllm_omni_amd.diffusion.attention.backends.abstract import AttentionMetadata

    ref = O.sdpa_nhd_general(q, k, v, 1 / math.sqrt(dh), attn_mask=bias.float())
    got = impl_for(H, dh).forward(g_(q), g_(k), g_(v), AttentionMetadata(attn_mask=bias.to(DEV)))
    check(got, ref)


def test_dense_bool_mask_with_fully_masked_rows_gives_zeros_there():
    from vllm_omni_amd.diffusion.attention.backends.abstract import AttentionMetadata

    B, Sq, Sk, H, dh = 1, 140, 100, 2, 64
    q, k, v = rnd((B, Sq, H, dh), 14), rnd((B, Sk, H, dh), 15), rnd((B, Sk, H, dh), 16)
    mask = torch.rand(B, H, Sq, Sk, generator=torch.Generator().manual_seed(17)) > 0.4
    mask[0, 0, 5] = False
    mask[0, 1, 139] = False
    ref = O.sdpa_nhd_general(q, k, v, 0.125, attn_mask=mask)
    got = impl_for(H, dh).forward(g_(q), g_(k), g_(v), AttentionMetadata(attn_mask=mask.to(DEV)))
    check(got, ref)
    assert float(got[0, 5, 0].abs().max()) == 0.0 and float(got[0, 139, 1].abs().max()) == 0.0


@pytest.mark.parametrize("dh,Sq,Sk", [(128, 300, 300), (64, 513, 513), (128, 100, 260), (64, 260, 100)])
def test_causal_top_left_aligned_like_torch(dh, Sq, Sk):
    """`causal=True` reaches SDPA as is_causal (sdpa.py:61): key j takes part in query i iff j <= i, also when S_q != S_kv."""
    B, H = 2, 3
    q, k, v = rnd((B, Sq, H, dh), 18), rnd((B, Sk, H, dh), 19), rnd((B, Sk, H, dh), 20)
    ref = O.sdpa_nhd_general(q, k, v, 1 / math.sqrt(dh), is_causal=True)
    check(impl_for(H, dh, causal=True).forward(g_(q), g_(k), g_(v), None), ref)


def test_grouped_kv_heads():
    B, Sq, Sk, H, Hkv, dh = 2, 130, 200, 8, 2, 128
    q, k, v = rnd((B, Sq, H, dh), 21), rnd((B, Sk, Hkv, dh), 22), rnd((B, Sk, Hkv, dh), 23)
    ref = O.sdpa_nhd_general(q, k, v, 1 / math.sqrt(dh))
    check(impl_for(H, dh, Hkv=Hkv).forward(g_(q), g_(k), g_(v), None), ref)


def test_ragged_items_through_the_ops_wrapper():
    """cu_seqlens_q / cu_seqlens_k with different lengths per item (what a step-batched caller passes), strided q (a column
    slice of a fused projection), a score spike late in the key sequence (the running max moves after many tiles)."""
    from vllm_omni_amd import ops

    H, dh = 4, 128
    ql, kl = [130, 1, 300], [77, 512, 65]
    qkv = rnd((sum(ql), 3 * H * dh), 24, 0.5)
    q = qkv[:, H * dh:2 * H * dh]
    k, v = rnd((sum(kl), H * dh), 25, 0.5), rnd((sum(kl), H * dh), 26)
    k[77 + 500, :dh] = bf16_round(q[130, :dh] * 30.0)
    cq = torch.tensor([0] + list(torch.tensor(ql).cumsum(0)), dtype=torch.int32, device=DEV)
    ck = torch.tensor([0] + list(torch.tensor(kl).cumsum(0)), dtype=torch.int32, device=DEV)
    got = ops.flash_attn_general(g_(qkv)[:, H * dh:2 * H * dh], g_(k), g_(v), cq, ck, H, H, max(ql), max(kl), 1 / math.sqrt(dh))
    torch.cuda.synchronize()
    qo = ko = 0
    for a, b in zip(ql, kl):
        ref = O.sdpa_nhd_general(q[qo:qo + a].reshape(1, a, H, dh), k[ko:ko + b].reshape(1, b, H, dh),
                                 v[ko:ko + b].reshape(1, b, H, dh), 1 / math.sqrt(dh)).reshape(a, H * dh)
        check(got[qo:qo + a], ref)
        qo, ko = qo + a, ko + b


def test_qwen_image_geometry_still_takes_the_tuned_kernel(monkeypatch):
    """No mask, S_q == S_kv, head size 128: the backend must keep dispatching to omni_flash_attn_fwd (the roofline kernels)."""
    from vllm_omni_amd import ops

    called = []
    orig = ops.flash_attn_varlen
    monkeypatch.setattr(ops, "flash_attn_varlen", lambda *a, **k: called.append(1) or orig(*a, **k))
    q = g_(rnd((1, 300, 2, 128), 27))
    impl_for(2, 128).forward(q, q, q, None)
    assert called


def test_backend_reports_its_head_sizes_truthfully():
    from vllm_omni_amd.diffusion.attention.selector import get_attn_backend

    be = get_attn_backend(64)
    assert be.get_name() == "CDNA4_FLASH" and be.get_supported_head_sizes() == [64, 128]
    assert be.supports_head_size(64) and be.supports_head_size(128) and not be.supports_head_size(96)
    with pytest.raises(NotImplementedError, match="head sizes"):
        be.get_impl_cls()(num_heads=2, head_size=96, softmax_scale=0.1)
