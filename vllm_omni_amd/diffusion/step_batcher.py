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

import collections
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
    def __init__(self, pipeline, max_items: int | None = None, max_steps_in_flight: int | None = None):
        self.pipeline = pipeline
        cfg = getattr(pipeline, "od_config", None)
        cap = max_items or int(getattr(cfg, "max_step_batch", 4) or 4)
        self.max_samples = max(1, cap)
        self.active: list[ActiveSample] = []
        self._pending: dict[Any, dict] = {}        # tag -> {"req": request, "left": samples still running, "done": {k: lat}}
        self._seq = itertools.count()
        # fairness across batch keys (resolutions / CFG settings): the key served longest ago goes next
        self._tick = 0
        self._last_served: dict[Any, int] = {}
        # bounded run-ahead: the host enqueues a step in ~10 ms, the GPU needs ~500 ms for it.  Unthrottled, the host composes
        # batches many steps ahead of GPU time — with whatever requests had arrived at HOST time: staggered arrivals then run
        # their first steps almost alone (round 3's serving line lost ~5 % to this).  One event per enqueued step; `ready()`
        # is false while `max_steps_in_flight` of them are still pending, so a newcomer joins within that many steps.
        self.max_steps_in_flight = int(max_steps_in_flight or getattr(cfg, "max_steps_in_flight", 2) or 2)
        self._inflight: collections.deque = collections.deque()        # (start event, end event, samples in the step)
        # serving statistics (bench.py's per-worker busy fraction): device seconds inside steps vs the span they cover
        self._stats = {"steps": 0, "sample_steps": 0, "busy_s": 0.0, "first": None, "last": None}

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

    # ------------------------------------------------------------------ run-ahead throttle
    def _device_is_gpu(self) -> bool:
        dev = getattr(self.pipeline, "device", None)
        return getattr(dev, "type", "cpu") == "cuda"

    def ready(self) -> bool:
        """May the host enqueue another step now?  (False: `max_steps_in_flight` steps are still queued on the device.)"""
        while self._inflight and self._inflight[0][1].query():
            self._retire(self._inflight.popleft())
        return len(self._inflight) < self.max_steps_in_flight

    def wait_ready(self, idle=None) -> None:
        """Block until `ready()`; `idle()` (e.g. the worker's inbox poll) runs while waiting, else the oldest event is awaited."""
        while not self.ready():
            if idle is None:
                self._inflight[0][1].synchronize()
            else:
                idle()

    def _retire(self, rec) -> None:
        e0, e1, n = rec
        st = self._stats
        st["steps"] += 1
        st["sample_steps"] += n
        st["busy_s"] += e0.elapsed_time(e1) * 1e-3
        st["first"] = st["first"] or e0
        st["last"] = e1

    def stats(self, reset: bool = False) -> dict:
        """{"steps", "sample_steps", "busy_s", "span_s", "busy_frac"} over the retired steps (device time from HIP events: the
        start event of a step executes when the stream reaches it, so end - start is the step's own execution time and
        span = first start .. last end includes whatever the device idled in between)."""
        if self._device_is_gpu():
            while self._inflight:
                self._inflight[0][1].synchronize()
                self._retire(self._inflight.popleft())
        st = self._stats
        span = st["first"].elapsed_time(st["last"]) * 1e-3 if st["first"] is not None else 0.0
        out = {"steps": st["steps"], "sample_steps": st["sample_steps"], "busy_s": st["busy_s"], "span_s": span,
               "busy_frac": (st["busy_s"] / span) if span > 0 else None}
        if reset:
            self._stats = {"steps": 0, "sample_steps": 0, "busy_s": 0.0, "first": None, "last": None}
        return out

    # ------------------------------------------------------------------ one scheduling quantum
    def next_group(self) -> list[ActiveSample]:
        """The samples of the next forward.  Batch keys (what one ragged forward can mix: token grid, CFG on / scale) are served
        ROUND-ROBIN — the key whose last forward lies furthest back goes next, so traffic at one resolution never waits behind
        another resolution's whole loop; within a key admission is FIFO: the `max_samples` oldest samples run, a waiting sample
        takes the first slot that frees up (continuous batching with a bounded batch, not time slicing — time slicing would
        re-compose the batch every step for no gain in throughput)."""
        by_key: dict[Any, list[ActiveSample]] = {}
        for a in sorted(self.active, key=lambda a: a.seq):
            by_key.setdefault(self.pipeline.batch_key(a), []).append(a)
        for k in [k for k in self._last_served if k not in by_key]:
            del self._last_served[k]
        key = min(by_key, key=lambda k: (self._last_served.get(k, -1), by_key[k][0].seq))
        self._tick += 1
        self._last_served[key] = self._tick
        return by_key[key][: self.max_samples]

    def step(self) -> list[tuple[Any, Any]]:
        """Advance one group of active samples by one denoising step; return finished (tag, output)s."""
        if not self.active:
            return []
        group = self.next_group()
        e0 = None
        if self._device_is_gpu():
            import torch

            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        try:
            self.pipeline.denoise_one_step(group)
        except Exception as e:  # noqa: BLE001 — a failing step aborts the REQUESTS that were in it, nothing else
            return self.abort({a.tag for a in group}, f"{type(e).__name__}: {e}")
        finished = []
        for a in group:
            if a.tag not in self._pending:                  # a sibling sample's failure aborted this request earlier in the loop
                continue
            a.step += 1
            if a.step >= a.n_steps:
                self.active.remove(a)
                p = self._pending[a.tag]
                try:
                    p["done"][a.sample["k"]] = self.pipeline.sample_result(a)
                except Exception as e:  # noqa: BLE001
                    finished += self.abort({a.tag}, f"{type(e).__name__}: {e}")
                    continue
                p["left"] -= 1
                if p["left"] == 0:
                    del self._pending[a.tag]
                    # the decode runs here, after the request left `active` and `_pending`: a failure (e.g. out of memory while
                    # decoding a large image) must still produce this request's answer, and only this request's
                    try:
                        out = self.pipeline.finish_request(p["req"], [p["done"][k] for k in range(p["n"])], a.sample)
                    except Exception as e:  # noqa: BLE001
                        from .data import DiffusionOutput

                        out = DiffusionOutput(error=f"{type(e).__name__}: {e}")
                    finished.append((a.tag, out))
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._inflight.append((e0, e1, len(group)))
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
            self.wait_ready()
            out += self.step()
        return out
