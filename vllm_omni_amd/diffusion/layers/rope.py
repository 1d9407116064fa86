"""RotaryEmbedding — mirror of vllm_omni/diffusion/layers/rope.py:68-151 with a native forward_hip.

The reference's HIP path needs flash-attn's Triton `apply_rotary` (rope.py:83-86,108-127); here it is
omni_rope_interleaved (interleaved / GPT-J pairing only, which is what Qwen-Image uses: is_neox_style=False).
"""
import torch

from ... import ops
from .custom_op import CustomOp


def apply_rotary_emb_torch(x, cos, sin, interleaved=False):
    """x [B,S,H,dh]; cos/sin [S, dh/2].  (reference rope.py:12-36)"""
    if not interleaved:
        c, s = torch.cat([cos, cos], -1)[None, :, None, :], torch.cat([sin, sin], -1)[None, :, None, :]
        x1, x2 = x.chunk(2, dim=-1)
        return x * c + torch.cat((-x2, x1), dim=-1) * s
    c = cos.repeat_interleave(2, -1)[None, :, None, :]
    s = sin.repeat_interleave(2, -1)[None, :, None, :]
    rot = torch.stack((-x[..., 1::2], x[..., 0::2]), dim=-1).flatten(-2)
    return x * c + rot * s


class RotaryEmbedding(CustomOp):
    def __init__(self, is_neox_style: bool = False) -> None:
        super().__init__()
        self.is_neox_style = is_neox_style
        self.interleaved = not is_neox_style

    def forward_hip(self, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
        if not self.interleaved:
            raise NotImplementedError("neox-style rotation is not on the Qwen-Image path")
        if cos.dim() == 3:
            cos, sin = cos[0], sin[0]
        return ops.rope_interleaved(x.contiguous(), cos.to(x.dtype), sin.to(x.dtype))

    def forward_native(self, x, cos, sin):
        return apply_rotary_emb_torch(x, cos, sin, interleaved=self.interleaved)
