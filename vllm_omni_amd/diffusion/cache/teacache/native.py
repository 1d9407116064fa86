"""Device-resident TeaCache state of one step-batch (omni_teacache in include/omni_cdna4.h).

The reference keeps `TeaCacheState` on the host (state.py) and decides with `.cpu().item()` per forward (hook.py:195-206).
Here every item (request x CFG branch) of a step-batch owns a slot in a few device arrays; `omni_dit_forward` evaluates the
decision with two small kernels and gates the block stack per item on the device, so a denoising loop with TeaCache issues a
fixed, sync-free launch sequence (hipGraph-capturable) and step-batched requests keep their own B=1 decisions."""
from __future__ import annotations

import torch

from .... import _native as N
from ...batch import RaggedBatch
from .config import TeaCacheConfig

BF16 = torch.bfloat16


class TeaCacheDeviceState:
    def __init__(self, config: TeaCacheConfig, rb: RaggedBatch, D: int, device):
        self.config = config
        n, Ri, Rt = rb.n_items, rb.n_img_rows, rb.n_txt_rows
        if n > 64:
            raise ValueError("TeaCache device state supports at most 64 items per step-batch")
        self.n_items, self.rows = n, (Ri, Rt)
        self.prev_mod = torch.zeros(Ri * D, dtype=BF16, device=device)
        self.prev_res = torch.zeros(Ri * D, dtype=BF16, device=device)
        self.acc = torch.zeros(n, dtype=torch.float32, device=device)
        self.cnt = torch.zeros(n, dtype=torch.int32, device=device)
        self.skip = torch.zeros(n, dtype=torch.int32, device=device)
        self.skip_total = torch.zeros(n, dtype=torch.int32, device=device)
        self.scratch = torch.zeros(2 * n, dtype=torch.float32, device=device)
        self.tile_img = torch.zeros((Ri + 255) // 256, dtype=torch.int32, device=device)
        self.tile_txt = torch.zeros((Rt + 255) // 256, dtype=torch.int32, device=device)
        cu = [0]
        for t in rb.txt_lens:
            cu.append(cu[-1] + int(t))
        self.txt_cu = torch.tensor(cu, dtype=torch.int32, device=device)
        s = N.TeaCache()
        s.rel_l1_thresh = float(config.rel_l1_thresh)
        for i, c in enumerate(config.coefficients):
            s.coeff[i] = float(c)
        s.prev_mod, s.prev_res = self.prev_mod.data_ptr(), self.prev_res.data_ptr()
        s.acc_dist, s.cnt, s.skip, s.skip_total = (t.data_ptr() for t in (self.acc, self.cnt, self.skip, self.skip_total))
        s.scratch, s.tile_skip_img, s.tile_skip_txt = self.scratch.data_ptr(), self.tile_img.data_ptr(), self.tile_txt.data_ptr()
        s.txt_cu = self.txt_cu.data_ptr()
        self._struct = s

    # ---- per-item state hand-over (continuous step batching: a sample keeps its TeaCache history when the batch it runs in
    # is re-composed).  prev_mod has the layout the first AdaLN writes: K32-blocked [D/32][rows][32] when D % 32 == 0.
    def _item_views(self, i: int):
        Ri = self.rows[0]
        S = Ri // self.n_items
        D = self.prev_res.numel() // Ri
        res = self.prev_res.view(Ri, D)[i * S:(i + 1) * S]
        mod = self.prev_mod.view(D // 32, Ri, 32)[:, i * S:(i + 1) * S] if D % 32 == 0 else self.prev_mod.view(Ri, D)[i * S:(i + 1) * S]
        return mod, res

    def export_item(self, i: int) -> dict:
        mod, res = self._item_views(i)
        return dict(mod=mod.clone(), res=res.clone(), acc=self.acc[i:i + 1].clone(), cnt=self.cnt[i:i + 1].clone(),
                    skip_total=self.skip_total[i:i + 1].clone())

    def import_item(self, i: int, saved: dict | None) -> None:
        """Load a sample's history into slot i; None = a fresh sample (cnt = 0 forces a full compute first)."""
        if saved is None:
            for t in (self.acc, self.cnt, self.skip_total, self.skip):
                t[i:i + 1].zero_()
            return
        mod, res = self._item_views(i)
        mod.copy_(saved["mod"])
        res.copy_(saved["res"])
        self.acc[i:i + 1].copy_(saved["acc"])
        self.cnt[i:i + 1].copy_(saved["cnt"])
        self.skip_total[i:i + 1].copy_(saved["skip_total"])

    def struct_for(self, rb: RaggedBatch) -> N.TeaCache:
        if (rb.n_img_rows, rb.n_txt_rows) != self.rows or rb.n_items != self.n_items:
            raise ValueError("TeaCache state was built for a different batch")
        return self._struct

    def reset(self) -> None:
        """New generation: counters and accumulators to zero (state.py:31-37); stale residuals are unreachable (cnt = 0
        forces a full compute first)."""
        for t in (self.acc, self.cnt, self.skip, self.skip_total, self.scratch):
            t.zero_()

    def skipped_forwards(self) -> list[int]:
        """Per item: how many forwards reused the cached residual since the last reset (one host read, after the loop)."""
        return self.skip_total.tolist()
