"""CDNA4_FLASH attention backend: omni_flash_attn_fwd behind the reference's AttentionBackend/AttentionImpl
contract (vllm_omni/diffusion/attention/backends/abstract.py; sibling of sdpa.py / flash_attn.py / sage_attn.py).

q/k/v arrive as [B, S, H, dh] ("NHD"); the kernel consumes exactly that layout (flattened over B,S with uniform
cu_seqlens), so unlike SDPAImpl there is no permute (sdpa.py:53,65).  Non-causal, no mask: Qwen-Image never
passes one (the block ignores encoder_hidden_states_mask, qwen_image_transformer.py:545).
"""
import torch

from .... import ops
from .abstract import AttentionBackend, AttentionImpl, AttentionMetadata


class CDNA4FlashImpl(AttentionImpl):
    def __init__(self, num_heads: int, head_size: int, softmax_scale: float, causal: bool = False,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None:
        super().__init__(num_heads, head_size, softmax_scale, causal, num_kv_heads, prefix)
        if causal:
            raise NotImplementedError("the diffusion path is non-causal")
        if head_size != 128:
            raise NotImplementedError("CDNA4_FLASH is built for head_size 128")
        if self.num_kv_heads != num_heads:
            raise NotImplementedError("GQA is not on the Qwen-Image path")
        self._cu = {}

    def forward(self, query, key, value, attn_metadata: AttentionMetadata = None) -> torch.Tensor:
        if attn_metadata is not None and attn_metadata.attn_mask is not None:
            raise NotImplementedError("attention masks are not on the Qwen-Image path")
        B, S, H, dh = query.shape
        if key.shape[1] != S:
            raise NotImplementedError("q and k/v sequence lengths must match (joint self-attention)")
        cu = self._cu.get((B, S, query.device))
        if cu is None:
            cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(query.device)
            self._cu[(B, S, query.device)] = cu
        q2, k2, v2 = (t.reshape(B * S, H * dh) for t in (query, key, value))
        out = ops.flash_attn_varlen(q2, k2, v2, cu, H, S, self.softmax_scale)
        return out.view(B, S, H, dh)


class CDNA4FlashBackend(AttentionBackend):
    accept_output_buffer = True
    NAME = "CDNA4_FLASH"
    IMPL = CDNA4FlashImpl
    HEAD_SIZES = (128,)
