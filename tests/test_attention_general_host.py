"""CPU: the checker of the general attention path is pinned to the op the reference calls.

`SDPAImpl.forward` (vllm_omni/diffusion/attention/backends/sdpa.py:46-66) is a permute around
`torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=..., dropout_p=0.0, is_causal=..., scale=...)`.  torch is the
same torch here, so `O.sdpa_nhd_general` (the spelled-out restatement the GPU tests check the HIP kernel against) is compared
with that very call on the host for every feature the kernel claims: cross-attention, bool / additive masks of every broadcast
shape, is_causal with S_q != S_kv, grouped K / V heads.  Also: the CDNA4_FLASH backend's mask normalisation (host arithmetic)."""
import math

import pytest
import torch
import torch.nn.functional as F

import qwen_image_oracle as O


def _ref_sdpa_impl(q, k, v, scale, mask=None, causal=False):
    """The reference's SDPAImpl.forward, verbatim in behaviour (sdpa.py:53-66)."""
    q, k, v = (x.permute(0, 2, 1, 3) for x in (q, k, v))
    if k.shape[1] != q.shape[1]:
        k, v = k.repeat_interleave(q.shape[1] // k.shape[1], 1), v.repeat_interleave(q.shape[1] // v.shape[1], 1)
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=causal, scale=scale).permute(0, 2, 1, 3)


@pytest.mark.parametrize("case", ["cross", "bool_b11k", "bool_dense", "add_qk", "add_bhqk", "causal_sq", "causal_wide", "causal_tall", "gqa"])
def test_general_sdpa_oracle_equals_the_torch_op_the_reference_calls(case):
    g = torch.Generator().manual_seed(3)
    B, Sq, Sk, H, dh = 2, 37, 53, 4, 64
    if case == "causal_sq":
        Sk = Sq
    if case == "causal_tall":
        Sq, Sk = 53, 37
    Hkv = 2 if case == "gqa" else H
    q, k, v = torch.randn(B, Sq, H, dh, generator=g), torch.randn(B, Sk, Hkv, dh, generator=g), torch.randn(B, Sk, Hkv, dh, generator=g)
    mask, causal = None, case.startswith("causal")
    if case == "bool_b11k":
        mask = torch.zeros(B, 1, 1, Sk, dtype=torch.bool)
        mask[0, ..., :40] = True
        mask[1, ..., :7] = True
    elif case == "bool_dense":
        mask = torch.rand(B, H, Sq, Sk, generator=g) > 0.3
        mask[..., 0] = True                                    # no dead rows: torch's math path would return NaN there
    elif case == "add_qk":
        mask = torch.randn(Sq, Sk, generator=g)
    elif case == "add_bhqk":
        mask = torch.randn(B, H, Sq, Sk, generator=g)
        mask[..., 1::3] = float("-inf")
    want = _ref_sdpa_impl(q, k, v, 1 / math.sqrt(dh), mask, causal)
    got = O.sdpa_nhd_general(q, k, v, 1 / math.sqrt(dh), attn_mask=mask, is_causal=causal)
    if case == "causal_tall":                                  # queries beyond the key count still see keys 0..Sk-1; none is dead
        assert torch.isfinite(want).all()
    assert torch.allclose(got, want, atol=2e-6, rtol=1e-5), float((got - want).abs().max())


def test_plain_oracle_is_the_general_one_without_options():
    g = torch.Generator().manual_seed(4)
    q, k, v = (torch.randn(1, 20, 2, 32, generator=g) for _ in range(3))
    assert torch.allclose(O.sdpa_nhd(q, k, v, 0.2), O.sdpa_nhd_general(q, k, v, 0.2), atol=1e-6)


def test_backend_mask_normalisation_strides():
    from vllm_omni_amd.diffusion.attention.backends.cdna4_flash import _mask_strides

    B, H, Sq, Sk = 2, 3, 5, 7
    m, st = _mask_strides(torch.zeros(B, 1, 1, Sk, dtype=torch.bool), B, H, Sq, Sk)
    assert st == (Sk, 0, 0, 1) and m.shape == (B, H, Sq, Sk)
    assert _mask_strides(torch.zeros(Sq, Sk), B, H, Sq, Sk)[1] == (0, 0, Sk, 1)
    assert _mask_strides(torch.zeros(1, H, Sq, Sk), B, H, Sq, Sk)[1] == (0, Sq * Sk, Sk, 1)
    t = torch.zeros(B, H, Sk, Sq).transpose(2, 3)             # a non-contiguous dense mask: its strides go through as they are
    assert _mask_strides(t, B, H, Sq, Sk)[1] == (H * Sq * Sk, Sq * Sk, 1, Sq)
    assert _mask_strides(torch.zeros(Sq, Sk, dtype=torch.float16), B, H, Sq, Sk)[0].dtype == torch.float32
    with pytest.raises(ValueError, match="broadcast"):
        _mask_strides(torch.zeros(B, H, Sq, Sk + 1), B, H, Sq, Sk)
    with pytest.raises(ValueError, match="at most 4"):
        _mask_strides(torch.zeros(1, 1, 1, 1, Sk), B, H, Sq, Sk)
