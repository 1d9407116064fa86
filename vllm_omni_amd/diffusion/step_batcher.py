"""Continuous step-batching of denoising loops inside one GPU worker (new design — the reference's worker
runs `reqs[0]` only and its scheduler broadcasts one request at a time, vllm_omni/diffusion/worker/gpu_worker.py:128-130,
scheduler.py:51-75).

Every denoising loop is a sequence of DiT forwards whose inputs at step i depend only on that request's own latents, so the
forwards of DIFFERENT requests at DIFFERENT step indices can share one ragged DiT forward: each request owns one row of the
timestep-embedding table (`temb_rows`), its CFG pair shares that row, and the fused CFG + Euler kernel takes a per-request
`dt`.  Requests join the running batch between two steps and leave it when their last step is done — no request waits for
another request's loop to finish (the weights, 41 GB per forward, are streamed once for all of them: at 256^2 a lone request is
weight-bandwidth bound).  Per-request results are those of a solo run (B=1 semantics: no padding, no cross-request attention).

The batcher is model-agnostic: the pipeline supplies `resolve_request` (request -> samples), `denoise_one_step` (one forward +
update for a list of active samples) and `finish_request`."""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Any


@dataclass
class ActiveSample:
    tag: Any                      # request handle chosen by the caller
    sample: dict                  # pipeline.resolve_request() entry (latents, prompt rows, grid, cfg ...)
    step: int = 0
    n_steps: int = 0
    state: dict = field(default_factory=dict)      # per-sample schedule tensors (pipeline-owned)
    seq: int = 0                  # arrival order


class ContinuousStepBatcher:
    def __init__(self, pipeline, max_items: int | None = None):
        self.pipeline = pipeline
        cap = max_items or int(getattr(getattr(pipeline, "od_config", None), "max_step_batch", 4) or 4)
        self.max_samples = max(1, cap)
        self.active: list[ActiveSample] = []
        self._pending: dict[Any, dict] = {}        # tag -> {"req": request, "left": samples still running, "done": {k: lat}}
        self._seq = itertools.count()

    # ------------------------------------------------------------------ admission
    def add(self, req, tag) -> None:
        """Validate + expand the request now (errors surface at admission, not in the middle of someone else's step)."""
        samples = self.pipeline.resolve_request(req)
        self._pending[tag] = {"req": req, "left": len(samples), "done": {}, "n": len(samples)}
        for sm in samples:
            a = ActiveSample(tag=tag, sample=sm, seq=next(self._seq))
            self.pipeline.begin_sample(a)
            self.active.append(a)

    def has_work(self) -> bool:
        return bool(self.active)

    def outstanding_steps(self) -> int:
        return sum(a.n_steps - a.step for a in self.active)

    # ------------------------------------------------------------------ one scheduling quantum
    def step(self) -> list[tuple[Any, Any]]:
        """Advance the OLDEST compatible group of active samples by one denoising step; return finished (tag, output)s."""
        if not self.active:
            return []
        self.active.sort(key=lambda a: a.seq)
        head = self.active[0]
        key = self.pipeline.batch_key(head)
        group = [a for a in self.active if self.pipeline.batch_key(a) == key][: self.max_samples]
        try:
            self.pipeline.denoise_one_step(group)
        except Exception as e:  # noqa: BLE001 — a failing step aborts the REQUESTS that were in it, nothing else
            return self.abort({a.tag for a in group}, f"{type(e).__name__}: {e}")
        finished = []
        for a in group:
            a.step += 1
            if a.step >= a.n_steps:
                self.active.remove(a)
                p = self._pending[a.tag]
                p["done"][a.sample["k"]] = self.pipeline.sample_result(a)
                p["left"] -= 1
                if p["left"] == 0:
                    del self._pending[a.tag]
                    finished.append((a.tag, self.pipeline.finish_request(p["req"], [p["done"][k] for k in range(p["n"])],
                                                                         a.sample)))
        return finished

    def abort(self, tags, error: str) -> list[tuple[Any, Any]]:
        """Drop every sample of the given requests (all samples of a request succeed together or not at all) and report ONE
        error output per request."""
        from .data import DiffusionOutput

        tags = set(tags)
        for a in [a for a in self.active if a.tag in tags]:
            self.active.remove(a)
            release = getattr(self.pipeline, "_export_sample", None)
            if release is not None:
                try:
                    release(a)
                except Exception:  # noqa: BLE001
                    pass
        for t in tags:
            self._pending.pop(t, None)
        return [(t, DiffusionOutput(error=error)) for t in sorted(tags, key=str)]

    def drain(self) -> list[tuple[Any, Any]]:
        out = []
        while self.active:
            out += self.step()
        return out
