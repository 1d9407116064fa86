"""3-axis rotary position tables for Qwen-Image (host side; computed once per (grid, text length), cached).

Semantics follow QwenEmbedRope with scale_rope=True (reference
vllm_omni/diffusion/models/qwen_image/qwen_image_transformer.py:179-285): head_dim 128 = 64 rotation pairs split
(8, 28, 28) over the (frame, height, width) axes, theta 10000; height/width indices are centred
([-(n - n//2) .. n//2 - 1]); text tokens sit on the diagonal (same index on all three axes) starting at
max(h//2, w//2).  The reference keeps a complex64 table and casts cos/sin to the activation dtype before
rotating (:403-406); we store the bf16 cos/sin directly, [text rows | image rows] in one table so that a single
int32 per joint row (`joint_pos`) addresses it.
"""
from __future__ import annotations

import functools

import torch

AXES_DIM = (16, 56, 56)
THETA = 10000.0


def _axis_angles(index: torch.Tensor, dim: int) -> torch.Tensor:
    inv_freq = 1.0 / torch.pow(torch.tensor(THETA, dtype=torch.float32),
                               torch.arange(0, dim, 2, dtype=torch.float32) / dim)
    return index.to(torch.float32)[:, None] * inv_freq[None, :]


def normalize_grids(grid) -> tuple[tuple[int, ...], ...]:
    """A token grid is one (frames, h, w) triple or a SEQUENCE of them: the Edit pipelines put the target latents and the
    condition images on one sequence axis, each entry with its own frame index (reference :231-250, idx -> frame offset).
    Entries of FOUR numbers (frames, h, w, frame_index) are the Layered variant's (`QwenEmbedLayer3DRope`, :65-176): the
    frame index is explicit (layer l -> l, the condition image -> -1) and the text positions start behind max(h/2, w/2,
    number of layers) — see `layered_grids`."""
    if len(grid) in (3, 4) and all(isinstance(v, int) for v in grid):
        return (tuple(grid),)
    out = tuple(tuple(int(v) for v in g) for g in grid)
    if not out or any(len(g) not in (3, 4) for g in out) or len({len(g) for g in out}) != 1:
        raise ValueError(f"bad token grid {grid!r}")
    return out


def layered_grids(grids) -> tuple[tuple[int, int, int, int], ...]:
    """img_shapes of the Layered pipeline — [(1, h, w)] * (layers + 1) + [(1, h_c, w_c)] — as explicit-frame-index entries:
    entry idx < last sits at frame idx, the LAST entry (the condition image) at frame -1 (reference :117-129)."""
    g = normalize_grids(grids)
    if any(len(e) == 4 for e in g):
        return g
    n = len(g) - 1
    return tuple((f, h, w, (idx if idx != n else -1)) for idx, (f, h, w) in enumerate(g))


def grid_tokens(grid) -> int:
    return sum(e[0] * e[1] * e[2] for e in normalize_grids(grid))


@functools.lru_cache(maxsize=64)
def _rope_table(grids: tuple[tuple[int, ...], ...], n_txt_pos: int) -> tuple[torch.Tensor, torch.Tensor]:
    parts, start = [], 0
    layered = len(grids[0]) == 4
    for idx, e in enumerate(grids):
        f, h, w = e[:3]
        f0 = e[3] if layered else idx                         # entry idx starts at frame position idx (:268); Layered: explicit
        if layered and f != 1:
            raise ValueError("Layered RoPE entries hold one frame each (reference :155,177: one row of the frame table)")
        fi = torch.arange(f0, f0 + f)
        hi = torch.arange(h) - (h - h // 2)
        wi = torch.arange(w) - (w - w // 2)
        af, ah, aw = (_axis_angles(i, d) for i, d in zip((fi, hi, wi), AXES_DIM))
        parts.append(torch.cat([
            af[:, None, None, :].expand(f, h, w, -1),
            ah[None, :, None, :].expand(f, h, w, -1),
            aw[None, None, :, :].expand(f, h, w, -1)], dim=-1).reshape(f * h * w, -1))
        start = max(start, h // 2, w // 2)                    # text positions start behind the largest half-extent (:251-257)
    if layered:
        start = max(start, len(grids) - 1)                    # ... and behind the layer count (:136: max(max_vid_index, layer_num))
    ti = torch.arange(start, start + n_txt_pos)
    ang_txt = torch.cat([_axis_angles(ti, d) for d in AXES_DIM], dim=-1)
    ang = torch.cat([ang_txt] + parts, dim=0)
    return torch.cos(ang), torch.sin(ang)


def rope_table(grid, n_txt_pos: int) -> tuple[torch.Tensor, torch.Tensor]:
    """(cos, sin) fp32 CPU tensors of shape [n_txt_pos + n_image_tokens, 64]; rows [0, n_txt_pos) are text positions."""
    return _rope_table(normalize_grids(grid), int(n_txt_pos))
