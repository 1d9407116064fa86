"""CDNA4_FLASH attention backend: the HIP attention kernels behind the reference's AttentionBackend / AttentionImpl
contract (vllm_omni/diffusion/attention/backends/abstract.py; sibling of sdpa.py / flash_attn.py / sage_attn.py).

It accepts what the reference's default `SDPAImpl.forward` accepts (sdpa.py:46-66) — the backend selector is process-global
and cached (attention/selector.py:49-77), so whatever is registered also serves the other in-tree DiTs:

  * joint self-attention, head size 128, no mask (Qwen-Image, qwen_image_transformer.py:437-443): the two tuned kernels
    (`omni_flash_attn_fwd`: csrc/attention_w64.hip / attention.hip);
  * everything else — S_q != S_kv (cross-attention, wan2_2_transformer.py:243,340), head size 64 (sd3_transformer.py:108),
    `attn_metadata.attn_mask` (bool = attend / additive float, any shape broadcastable to [B, H, S_q, S_k]), `causal=True`,
    num_kv_heads < num_heads: `omni_flash_attn_general` (csrc/attention_general.hip).

q/k/v arrive as [B, S, H, dh] ("NHD"); the kernels consume exactly that layout (flattened over B, S with uniform
cu_seqlens), so unlike SDPAImpl there is no permute (sdpa.py:53,65).
"""
import torch

from .... import ops
from .abstract import AttentionBackend, AttentionImpl, AttentionMetadata

HEAD_SIZES = (64, 128)


def _mask_strides(mask: torch.Tensor, B: int, H: int, Sq: int, Sk: int) -> tuple[torch.Tensor, tuple[int, int, int, int]]:
    """The mask as SDPA reads it: broadcast to [B, H, S_q, S_k] (torch semantics: trailing dimensions align), returned with
    its element strides — a broadcast dimension has stride 0 and costs nothing."""
    if mask.dtype not in (torch.bool, torch.uint8, torch.bfloat16, torch.float32):
        mask = mask.to(torch.float32)            # fp16 / fp64 additive masks: one exact-or-rounded copy, the kernel adds fp32
    m = mask
    while m.dim() < 4:
        m = m.unsqueeze(0)
    if m.dim() != 4:
        raise ValueError(f"attn_mask must have at most 4 dimensions, got {tuple(mask.shape)}")
    try:
        m = m.expand(B, H, Sq, Sk)
    except RuntimeError as e:
        raise ValueError(f"attn_mask of shape {tuple(mask.shape)} does not broadcast to [{B}, {H}, {Sq}, {Sk}]") from e
    return m, tuple(m.stride())


class CDNA4FlashImpl(AttentionImpl):
    def __init__(self, num_heads: int, head_size: int, softmax_scale: float, causal: bool = False,
                 num_kv_heads: int | None = None, prefix: str = "", **extra_impl_args) -> None:
        super().__init__(num_heads, head_size, softmax_scale, causal, num_kv_heads, prefix)
        if head_size not in HEAD_SIZES:
            raise NotImplementedError(f"CDNA4_FLASH is built for head sizes {HEAD_SIZES}, got {head_size}")
        if num_heads % self.num_kv_heads:
            raise ValueError("num_heads must be a multiple of num_kv_heads")
        self._cu = {}

    def _cu_seqlens(self, B: int, S: int, device) -> torch.Tensor:
        cu = self._cu.get((B, S, device))
        if cu is None:
            cu = (torch.arange(B + 1, dtype=torch.int32) * S).to(device)
            self._cu[(B, S, device)] = cu
        return cu

    def forward(self, query, key, value, attn_metadata: AttentionMetadata = None) -> torch.Tensor:
        mask = attn_metadata.attn_mask if attn_metadata is not None else None
        B, Sq, H, dh = query.shape
        Sk, Hkv = key.shape[1], key.shape[2]
        if H != self.num_heads or dh != self.head_size or Hkv != self.num_kv_heads or value.shape != key.shape or key.shape[0] != B:
            raise ValueError(f"q {tuple(query.shape)} / k {tuple(key.shape)} / v {tuple(value.shape)} do not match "
                             f"num_heads {self.num_heads}, num_kv_heads {self.num_kv_heads}, head_size {self.head_size}")
        q2 = query.reshape(B * Sq, H * dh)
        k2, v2 = key.reshape(B * Sk, Hkv * dh), value.reshape(B * Sk, Hkv * dh)
        cu_q = self._cu_seqlens(B, Sq, query.device)
        if mask is None and not self.causal and Sq == Sk and dh == 128 and Hkv == H:
            out = ops.flash_attn_varlen(q2, k2, v2, cu_q, H, Sq, self.softmax_scale)          # the tuned self-attention kernels
        else:
            m, strides = (None, None) if mask is None else _mask_strides(mask.to(query.device), B, H, Sq, Sk)
            out = ops.flash_attn_general(q2, k2, v2, cu_q, self._cu_seqlens(B, Sk, query.device), H, Hkv, Sq, Sk,
                                         self.softmax_scale, causal=self.causal, mask=m, mask_strides=strides)
        return out.view(B, Sq, H, dh)


class CDNA4FlashBackend(AttentionBackend):
    accept_output_buffer = True
    NAME = "CDNA4_FLASH"
    IMPL = CDNA4FlashImpl
    HEAD_SIZES = HEAD_SIZES
