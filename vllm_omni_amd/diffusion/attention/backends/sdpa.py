"""TORCH_SDPA backend — the reference's default `SDPAImpl` (vllm_omni/diffusion/attention/backends/sdpa.py:32-66):
`F.scaled_dot_product_attention` on [B, S, H, dh] HOST tensors.  It exists for CPU-only hosts (the strategy / plug-in tests
run on it; selectable there with DIFFUSION_ATTENTION_BACKEND=TORCH_SDPA) and REFUSES device tensors: on a GPU the only attention
is the HIP kernel (CDNA4_FLASH) — an environment variable must not silently swap it for a torch op (no dual backend).  The native
DiT forward never comes through here; it calls omni_flash_attn_fwd directly."""
import torch
import torch.nn.functional as F

from .abstract import AttentionBackend, AttentionImpl, AttentionMetadata


class SDPAImpl(AttentionImpl):
    def forward(self, query, key, value, attn_metadata: AttentionMetadata = None) -> torch.Tensor:
        if query.is_cuda or key.is_cuda or value.is_cuda:
            from .... import _native

            raise _native.OmniNativeError("TORCH_SDPA serves host tensors only (CPU-side tests of the plug-in surface); device tensors "
                                          "run on the CDNA4_FLASH backend - unset DIFFUSION_ATTENTION_BACKEND")
        q, k, v = (t.permute(0, 2, 1, 3) for t in (query, key, value))
        mask = attn_metadata.attn_mask if attn_metadata is not None else None
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=self.causal,
                                             scale=self.softmax_scale)
        return out.permute(0, 2, 1, 3)


class SDPABackend(AttentionBackend):
    accept_output_buffer = True
    NAME = "TORCH_SDPA"
    IMPL = SDPAImpl
