"""AdaLayerNorm — mirror of vllm_omni/diffusion/layers/adalayernorm.py:10-102 with a real forward_hip.

out = LayerNorm(x; eps, no affine) * (1 + scale) + shift, returns (out, gate[B,1,D]); mod_params [B, 3D] chunks
as (shift, scale, gate).  The reference's forward_hip just calls forward_native (3 elementwise passes);
here forward_hip is ONE fused kernel (omni_adaln_modulate).  The `index` (Layered zero_cond_t) variant selects
between two modulation sets per token: it maps onto the kernel's per-row item map.
"""
import torch
import torch.nn as nn

from ... import ops
from .custom_op import CustomOp


class AdaLayerNorm(CustomOp):
    def __init__(self, hidden_size: int, elementwise_affine: bool = False, eps: float = 1e-6) -> None:
        super().__init__()
        if elementwise_affine:
            raise NotImplementedError("Qwen-Image uses elementwise_affine=False everywhere")
        self.eps = eps
        self.hidden_size = hidden_size
        self.layernorm = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=eps)

    def forward_hip(self, x: torch.Tensor, mod_params: torch.Tensor, index: torch.Tensor = None):
        B, S, D = x.shape
        mod = mod_params.contiguous()
        if index is None:
            y = ops.adaln_modulate(x.reshape(B * S, D), mod[:, D:], mod, mod_item_stride=3 * D, rows_per_item=S)
            gate = mod[:, 2 * D:].unsqueeze(1)
        else:
            # mod_params is [2B, 3D]: rows [0,B) apply where index == 0, rows [B,2B) where index == 1
            actual = mod.shape[0] // 2
            item = (torch.arange(actual, device=x.device, dtype=torch.int32)[:, None] + actual * (index != 0).int())
            y = ops.adaln_modulate(x.reshape(B * S, D), mod[:, D:], mod, mod_item_stride=3 * D,
                                   row_item_map=item.reshape(-1).contiguous())
            g = mod[:, 2 * D:]
            gate = torch.where((index == 0).unsqueeze(-1), g[:actual].unsqueeze(1), g[actual:].unsqueeze(1))
        return y.view(B, S, D), gate

    def forward_native(self, x: torch.Tensor, mod_params: torch.Tensor, index: torch.Tensor = None):
        shift, scale, gate = mod_params.chunk(3, dim=-1)
        if index is not None:
            actual = shift.size(0) // 2
            sel = (index == 0).unsqueeze(-1)
            shift = torch.where(sel, shift[:actual].unsqueeze(1), shift[actual:].unsqueeze(1))
            scale = torch.where(sel, scale[:actual].unsqueeze(1), scale[actual:].unsqueeze(1))
            gate = torch.where(sel, gate[:actual].unsqueeze(1), gate[actual:].unsqueeze(1))
        else:
            shift, scale, gate = shift.unsqueeze(1), scale.unsqueeze(1), gate.unsqueeze(1)
        return self.layernorm(x) * (1 + scale) + shift, gate
