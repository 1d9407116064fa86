"""TORCH_SDPA backend — the reference's default `SDPAImpl` (vllm_omni/diffusion/attention/backends/sdpa.py:32-66):
`F.scaled_dot_product_attention` on [B, S, H, dh] tensors.  Selectable with DIFFUSION_ATTENTION_BACKEND=TORCH_SDPA; it is
what CPU-only hosts get (the strategy / plug-in tests run on it) and it is NEVER used by the native DiT forward, which calls
omni_flash_attn_fwd directly."""
import torch
import torch.nn.functional as F

from .abstract import AttentionBackend, AttentionImpl, AttentionMetadata


class SDPAImpl(AttentionImpl):
    def forward(self, query, key, value, attn_metadata: AttentionMetadata = None) -> torch.Tensor:
        q, k, v = (t.permute(0, 2, 1, 3) for t in (query, key, value))
        mask = attn_metadata.attn_mask if attn_metadata is not None else None
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=self.causal,
                                             scale=self.softmax_scale)
        return out.permute(0, 2, 1, 3)


class SDPABackend(AttentionBackend):
    accept_output_buffer = True
    NAME = "TORCH_SDPA"
    IMPL = SDPAImpl
