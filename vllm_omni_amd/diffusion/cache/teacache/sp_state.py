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
        # the device kernel's arithmetic, operation for operation in fp32 (csrc/elementwise.hip teacache_decide_kernel:
        # `rb(rb(sd * inv_count) / rb(rb(sp * inv_count) + 1e-8f))`, rb = round to bf16; then an fp32 Horner) — python doubles
        # here (sd / count, + 1e-8 in double) could land on the other side of a bf16 rounding tie or of the threshold
        f32 = torch.float32
        s2 = sums.detach().to("cpu", f32)
        inv = torch.tensor(1.0, dtype=f32) / torch.tensor(float(count), dtype=f32)
        rb = lambda t: t.bfloat16().to(f32)  # noqa: E731
        rel = rb(rb(s2[0] * inv) / rb(rb(s2[1] * inv) + torch.tensor(1e-8, dtype=f32)))
        r = torch.tensor(float(self.config.coefficients[0]), dtype=f32)
        for c in self.config.coefficients[1:]:             # numpy.poly1d order (highest power first); mul then add, as `r * rel + c`
            r = r * rel + torch.tensor(float(c), dtype=f32)
        self.acc = float(torch.tensor(self.acc, dtype=f32) + r.abs())
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
