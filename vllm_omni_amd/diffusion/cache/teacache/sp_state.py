"""TeaCache state of ONE item on ONE sequence-parallel rank (Ulysses; SURVEY.md 8f N2 x N3).

The reference's hook (vllm_omni/diffusion/cache/teacache/hook.py:82-168,170-217) decides from the relative L1 distance of
consecutive modulated inputs of block 0 — a MEAN over all image rows — and caches the block stack's residual.  Its extractor
does not know about sequence parallelism (extractors.py:189-245 walks the un-sharded forward), so the reference defines no
SP behaviour to match; the contract here is the reference's own SP contract (SP == non-SP,
tests/e2e/offline_inference/test_sequence_parallel.py:128-147): every rank sums its row slice, one all-reduce of two floats makes
the sums global (`QwenImageTransformer2DModel._sp_forward_gen`), and `decide` applies the SAME rule, with the same bf16
roundings, as the single-device paths (native.py / csrc/elementwise.hip `teacache_decide_kernel`, hook.py) — so all ranks skip
or compute together and the skip pattern is the single-device one.  The cached residual and the previous modulated input are
this rank's row slice."""
from __future__ import annotations

import torch

from .config import TeaCacheConfig


def _bf16(x: float) -> float:
    return float(torch.tensor(x, dtype=torch.float32).bfloat16().float())


class TeaCacheSPState:
    def __init__(self, config: TeaCacheConfig):
        self.config = config
        self.reset()

    def reset(self) -> None:
        self.cnt, self.acc, self.skipped = 0, 0.0, 0
        self.prev_mod: torch.Tensor | None = None       # [S_loc, D] bf16
        self.prev_res: torch.Tensor | None = None       # [S_loc, D] bf16
        self.decisions: list[bool] = []                 # True = computed

    def first(self) -> None:
        """First forward of a generation: always compute (hook.py:185-188)."""
        self.acc = 0.0
        self.cnt += 1
        self.decisions.append(True)

    def decide(self, sums: torch.Tensor, count: int) -> bool:
        """sums = [sum |cur - prev|, sum |prev|] over ALL ranks' rows, count = elements of the whole modulated input.
        Returns True = compute.  One host read per forward, like the reference's `.cpu().item()`."""
        sd, sp = (float(v) for v in sums.detach().float().cpu().tolist())
        # the reference forms the ratio from bf16 tensors: .abs().mean() -> bf16, (+ 1e-8) -> bf16, division -> bf16
        rel = _bf16(_bf16(sd / count) / _bf16(_bf16(sp / count) + 1e-8))
        r = torch.tensor(float(self.config.coefficients[0]), dtype=torch.float32)
        for c in self.config.coefficients[1:]:             # numpy.poly1d order (highest power first), float32 Horner as on the device
            r = r * torch.tensor(rel, dtype=torch.float32) + torch.tensor(float(c), dtype=torch.float32)
        self.acc = float(torch.tensor(self.acc, dtype=torch.float32) + r.abs())
        self.cnt += 1
        if self.acc < float(self.config.rel_l1_thresh):
            self.skipped += 1
            self.decisions.append(False)
            return False
        self.acc = 0.0
        self.decisions.append(True)
        return True


class TeaCacheSPStats:
    """What `pipeline.last_teacache_state` exposes after a sequence-parallel loop (the device state's statistics surface)."""

    def __init__(self, states: list[TeaCacheSPState]):
        self.states = states

    def skipped_forwards(self) -> list[int]:
        return [s.skipped for s in self.states]
