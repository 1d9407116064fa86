"""Ragged, token-major batch descriptors for the native DiT runner (host logic, no GPU needed).

The reference runs one request at a time (`reqs[0]`, vllm_omni/diffusion/worker/gpu_worker.py:128-130) and
zero-pads unequal prompts when B > 1 (pipeline_qwen_image.py:386-392), after which padded text rows ARE
attended to (the block ignores encoder_hidden_states_mask, qwen_image_transformer.py:545).  Parity target is
therefore per-request B=1 semantics; this module lets several requests (and the two true-CFG branches of one
request) share a DiT forward WITHOUT padding: rows of all items are concatenated, and small int32 maps tell the
kernels which item / joint row / RoPE position every token belongs to.

Item i contributes T_i text rows and S_img image rows.  Three row spaces:
  image stream  [n_img_rows]  item-major                 (residual stream `hidden_states`)
  text stream   [n_txt_rows]  item-major                 (residual stream `encoder_hidden_states`)
  joint         [n_joint_rows] per item [text_i ; image_i]  (q/k/v/attention; reference order, :412-416)
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch


@dataclass
class RaggedBatch:
    txt_lens: list[int]
    s_img: int
    temb_rows: list[int]          # item -> row of the timestep-embedding table (items may share one)
    n_temb: int
    grid: tuple                   # (frames, h/16, w/16) latent token grid, or a tuple of such triples (target + condition
                                  # images on one sequence axis), identical for all items
    txt_pos_end: int              # rope table rows [0, txt_pos_end) are text positions
    img_start: int = 0            # first image token of the grid held by this batch (sequence-parallel chunk)
    cu_seqlens: np.ndarray = field(repr=False, default=None)
    img_item: np.ndarray = field(repr=False, default=None)
    txt_item: np.ndarray = field(repr=False, default=None)
    img_joint_row: np.ndarray = field(repr=False, default=None)
    txt_joint_row: np.ndarray = field(repr=False, default=None)
    joint_pos: np.ndarray = field(repr=False, default=None)

    @property
    def n_items(self) -> int:
        return len(self.txt_lens)

    @property
    def n_img_rows(self) -> int:
        return self.n_items * self.s_img

    @property
    def n_txt_rows(self) -> int:
        return int(sum(self.txt_lens))

    @property
    def n_joint_rows(self) -> int:
        return self.n_img_rows + self.n_txt_rows

    @property
    def max_seqlen(self) -> int:
        return max(self.txt_lens) + self.s_img

    def device_maps(self, device) -> dict[str, torch.Tensor]:
        names = ("cu_seqlens", "img_item", "txt_item", "img_joint_row", "txt_joint_row", "joint_pos")
        return {n: torch.from_numpy(getattr(self, n)).to(device) for n in names}


def build_ragged_batch(txt_lens: list[int], grid: tuple[int, int, int], temb_rows: list[int] | None = None,
                       txt_pos_end: int | None = None, img_rows: tuple[int, int] | None = None) -> RaggedBatch:
    """Build the int32 row maps for items with text lengths `txt_lens` on a common latent grid.
    `img_rows = (start, count)`: every item holds only that slice of the grid's image tokens (a sequence-parallel rank's
    chunk, reference qwen_image_transformer.py:735-738,772-781); RoPE positions stay those of the full grid."""
    if not txt_lens or any(t <= 0 for t in txt_lens):
        raise ValueError("every item needs at least one text token")
    from .models.qwen_image.rope import grid_tokens, normalize_grids

    grid = normalize_grids(grid)
    grid = grid[0] if len(grid) == 1 else grid          # one image: the plain (f, h, w) triple; several: a tuple of triples
    total = grid_tokens(grid)
    img_start, s_img = (0, total) if img_rows is None else (int(img_rows[0]), int(img_rows[1]))
    if img_start < 0 or s_img <= 0 or img_start + s_img > total:
        raise ValueError("img_rows outside the token grid")
    n = len(txt_lens)
    temb_rows = list(range(n)) if temb_rows is None else list(temb_rows)
    if len(temb_rows) != n:
        raise ValueError("temb_rows must have one entry per item")
    tmax = max(txt_lens)
    txt_pos_end = tmax if txt_pos_end is None else txt_pos_end
    if txt_pos_end < tmax:
        raise ValueError("txt_pos_end smaller than the longest prompt")
    seq = np.asarray([t + s_img for t in txt_lens], dtype=np.int64)
    cu = np.zeros(n + 1, dtype=np.int32)
    cu[1:] = np.cumsum(seq)
    img_item = np.repeat(np.asarray(temb_rows, dtype=np.int32), s_img)
    txt_item = np.repeat(np.asarray(temb_rows, dtype=np.int32), txt_lens)
    img_joint = np.concatenate([cu[i] + txt_lens[i] + np.arange(s_img, dtype=np.int32) for i in range(n)])
    txt_joint = np.concatenate([cu[i] + np.arange(txt_lens[i], dtype=np.int32) for i in range(n)])
    joint_pos = np.empty(int(cu[-1]), dtype=np.int32)
    for i in range(n):
        joint_pos[cu[i]: cu[i] + txt_lens[i]] = np.arange(txt_lens[i], dtype=np.int32)
        joint_pos[cu[i] + txt_lens[i]: cu[i + 1]] = txt_pos_end + img_start + np.arange(s_img, dtype=np.int32)
    return RaggedBatch(txt_lens=list(txt_lens), s_img=s_img, temb_rows=temb_rows, n_temb=max(temb_rows) + 1,
                       grid=tuple(grid), txt_pos_end=txt_pos_end, img_start=img_start, cu_seqlens=cu,
                       img_item=img_item.astype(np.int32), txt_item=txt_item.astype(np.int32),
                       img_joint_row=img_joint.astype(np.int32), txt_joint_row=txt_joint.astype(np.int32),
                       joint_pos=joint_pos)
