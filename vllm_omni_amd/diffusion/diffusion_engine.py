"""DiffusionEngine — mirror of vllm_omni/diffusion/diffusion_engine.py:56-363 with the SURVEY.md §8e dispatcher.

Same outer contract: `DiffusionEngine(od_config)` spawns one worker process per GPU and waits for "ready";
`step(requests)` pre-processes, runs, post-processes and returns `OmniRequestOutput`(s); `collective_rpc` broadcasts a
method call; `close()` shuts the workers down.

What is new (the reference replicates ONE request on every rank: scheduler.py:55-62, F6/F7): requests are DISPATCHED — each
whole request goes to the rank with the fewest outstanding denoising steps x tokens — and every worker step-batches its
requests continuously (step_batcher.py), decodes its own images and returns them through its result queue.  `submit()` /
`poll()` expose the same machinery asynchronously (requests may arrive while others are mid-loop)."""
from __future__ import annotations

import itertools
import os
import queue
import time
from typing import Any, Callable

import torch
import torch.multiprocessing as mp

from ..outputs import OmniRequestOutput
from .data import DiffusionOutput, OmniDiffusionConfig
from .request import OmniDiffusionRequest
from .worker.gpu_worker import SHUTDOWN, WorkerProc


def _cpu(v):
    """Device tensors (also inside lists / tuples) -> host tensors; everything else unchanged."""
    if isinstance(v, torch.Tensor):
        return v.cpu() if v.is_cuda else v
    if isinstance(v, (list, tuple)):
        return type(v)(_cpu(x) for x in v)
    return v


def _request_to_cpu(req: OmniDiffusionRequest) -> OmniDiffusionRequest:
    """Only host objects cross the process boundary to a worker (queues pickle their payload)."""
    for f in ("latents", "prompt_embeds", "prompt_embeds_mask", "negative_prompt_embeds", "negative_prompt_embeds_mask",
              "pil_image", "preprocessed_image", "prompt_image"):
        setattr(req, f, _cpu(getattr(req, f, None)))
    if req.extra:
        req.extra = {k: _cpu(v) for k, v in req.extra.items()}
    if isinstance(req.generator, torch.Generator):          # generators do not pickle: carry the seed instead
        req.seed, req.generator = (req.seed if req.seed is not None else req.generator.initial_seed()), None
    return req


def _free_port() -> int:
    """A TCP port nobody listens on right now (the reference probes upwards from a base port, data.py:360-391 `settle_port`;
    here the kernel picks one): the workers' rendezvous port when `od_config.master_port` is not given.  Two engines started in
    the same millisecond — `bench.py --gpus N` next to another job — no longer meet on a clock-derived port."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


class DiffusionEngine:
    def __init__(self, od_config: OmniDiffusionConfig, pipeline_factory: Callable[[], Any] | None = None,
                 post_process_func: Callable | None | str = "default", pre_process_func: Callable | None = None,
                 start_timeout_s: float = 600.0):
        self.od_config = od_config
        from .registry import get_diffusion_post_process_func, get_diffusion_pre_process_func

        if post_process_func == "default":              # reference diffusion_engine.py:68-69 / registry.py:135-146
            post_process_func = get_diffusion_post_process_func(od_config)
        if pre_process_func is None:
            pre_process_func = get_diffusion_pre_process_func(od_config)
        self.post_process_func, self.pre_process_func = post_process_func, pre_process_func
        self.num_gpus = int(od_config.num_gpus or 1)
        self.sp_degree = int(getattr(od_config.parallel_config, "ulysses_degree", 1) or 1)
        if self.num_gpus % self.sp_degree:
            raise ValueError(f"num_gpus {self.num_gpus} is not a multiple of ulysses_degree {self.sp_degree}")
        self.num_groups = self.num_gpus // self.sp_degree    # dispatch units: one data-parallel group = sp_degree ranks
        self._ctx = mp.get_context("spawn")
        self._inbox = [self._ctx.Queue() for _ in range(self.num_gpus)]
        self._outbox = self._ctx.Queue()
        self._ready = self._ctx.Queue()
        self._ids = itertools.count()
        self._load = [0.0] * self.num_groups            # outstanding steps x tokens per data-parallel group
        self._cost: dict[int, tuple[int, float]] = {}   # request id -> (rank, cost)
        self._results: dict[int, DiffusionOutput] = {}
        self._rpc_results: dict[int, dict[int, Any]] = {}
        self._closed = False
        port = od_config.master_port or _free_port()
        self._processes = [self._ctx.Process(target=WorkerProc.worker_main, name=f"DiffusionWorker-{r}", daemon=True,
                                             args=(r, self.num_gpus, od_config, self._inbox[r], self._outbox, self._ready,
                                                   pipeline_factory, port))
                           for r in range(self.num_gpus)]
        for p in self._processes:
            p.start()
        got = 0
        deadline = time.time() + start_timeout_s
        while got < self.num_gpus:
            try:
                m = self._ready.get(timeout=max(0.1, deadline - time.time()))
            except queue.Empty:
                self.close()
                raise TimeoutError("diffusion workers did not come up") from None
            if m["status"] != "ready":
                self.close()
                raise RuntimeError(f"worker {m['rank']} failed to start: {m.get('error')}")
            got += 1

    @staticmethod
    def make_engine(config: OmniDiffusionConfig, **kw) -> "DiffusionEngine":
        return DiffusionEngine(config, **kw)

    # ------------------------------------------------------------------ dispatcher
    @staticmethod
    def _request_cost(req: OmniDiffusionRequest) -> float:
        h, w = req.height or 1024, req.width or 1024
        n = max(1, int(req.num_outputs_per_prompt or 1)) * (len(req.prompt) if isinstance(req.prompt, list) else 1)
        return float((req.num_inference_steps or 50) * (h // 16) * (w // 16) * n)

    def submit(self, req: OmniDiffusionRequest) -> int:
        """Hand one request to the least-loaded rank; returns its ticket."""
        rid = next(self._ids)
        if self.sp_degree > 1 and req.seed is None and req.generator is None and req.latents is None:
            # the request is fanned out to every rank of a sequence-parallel group and each rank draws the initial noise
            # itself: without a shared seed they would denoise DIFFERENT latents and mix their predictions.  seed=None is the
            # default of the entry points, so it is made concrete here, once, before the fan-out.
            req.seed = int.from_bytes(os.urandom(4), "little")
        grp = min(range(self.num_groups), key=lambda r: (self._load[r], r))
        cost = self._request_cost(req)
        self._load[grp] += cost
        self._cost[rid] = (grp, cost)
        msg = {"type": "add", "id": rid, "request": _request_to_cpu(req)}
        for r in range(grp * self.sp_degree, (grp + 1) * self.sp_degree):       # every rank of a sequence-parallel group
            self._inbox[r].put(msg)
        return rid

    def _recv(self, timeout: float | None):
        """One message from the workers, or None (split from `_handle` so that an asynchronous front end can block on the
        queue WITHOUT holding the lock it takes around the dispatcher's bookkeeping)."""
        try:
            return self._outbox.get(timeout=timeout)
        except queue.Empty:
            return None

    def _pump(self, timeout: float | None) -> bool:
        m = self._recv(timeout)
        if m is None:
            return False
        self._handle(m)
        return True

    def _handle(self, m: dict) -> None:
        if m["type"] == "done":
            grp, cost = self._cost.pop(m["id"], (m["rank"] // self.sp_degree, 0.0))
            self._load[grp] = max(0.0, self._load[grp] - cost)
            self._results[m["id"]] = m["output"]
        elif m["type"] == "rpc_result":
            self._rpc_results.setdefault(m["id"], {})[m["rank"]] = m["result"]

    def to_request_output(self, req: OmniDiffusionRequest, out: DiffusionOutput) -> OmniRequestOutput:
        """DiffusionOutput of one request -> OmniRequestOutput (post-processing as in `step`)."""
        if out.error:
            raise RuntimeError(out.error)
        prompt = req.prompt[0] if isinstance(req.prompt, list) and req.prompt else req.prompt
        images = out.output
        if images is not None and req.output_type != "latent" and self.post_process_func is not None:
            images = self.post_process_func(images)
        imgs = [] if images is None else (list(images) if not isinstance(images, list) else images)
        return OmniRequestOutput.from_diffusion(request_id=req.request_id or "", images=imgs, prompt=prompt, metrics={},
                                                latents=out.output if req.output_type == "latent" else None)

    def poll(self, rid: int, timeout: float | None = None) -> DiffusionOutput | None:
        deadline = None if timeout is None else time.time() + timeout
        while rid not in self._results:
            left = None if deadline is None else deadline - time.time()
            if left is not None and left <= 0:
                return None
            self._pump(left if left is not None else 1.0)
            if any(not p.is_alive() for p in self._processes) and rid not in self._results and self._outbox.empty():
                raise RuntimeError("a diffusion worker died")
        return self._results.pop(rid)

    def add_req_and_wait_for_response(self, requests: list[OmniDiffusionRequest]) -> list[DiffusionOutput]:
        ids = [self.submit(r) for r in requests]
        return [self.poll(i) for i in ids]

    # ------------------------------------------------------------------ reference-shaped blocking call
    def step(self, requests: list[OmniDiffusionRequest]):
        """diffusion_engine.py:74-170: one OmniRequestOutput for a single request, a list for several; errors are logged and
        swallowed into `None` like the reference (:168-170)."""
        try:
            if self.pre_process_func is not None:
                requests = self.pre_process_func(requests)
            outs = self.add_req_and_wait_for_response(requests)
            results = [self.to_request_output(req, out) for req, out in zip(requests, outs)]
            return results[0] if len(results) == 1 else results
        except Exception as e:  # noqa: BLE001
            print(f"[DiffusionEngine] Generation failed: {e}")
            return None

    def collective_rpc(self, method: str, args: tuple = (), kwargs: dict | None = None, timeout: float | None = None,
                       unique_reply_rank: int | None = None):
        """Broadcast `method(*args, **kwargs)` to every worker (`GPUWorker` first, then its pipeline), gather the replies
        (reference :275-340)."""
        rid = next(self._ids)
        for q in self._inbox:
            q.put({"type": "rpc", "id": rid, "method": method, "args": args, "kwargs": kwargs or {},
                   "output_rank": unique_reply_rank})
        want = 1 if unique_reply_rank is not None else self.num_gpus
        deadline = None if timeout is None else time.time() + timeout
        while len(self._rpc_results.get(rid, {})) < want:
            left = None if deadline is None else deadline - time.time()
            if left is not None and left <= 0:
                raise TimeoutError(f"RPC call to {method} timed out.")
            self._pump(left if left is not None else 1.0)
        res = self._rpc_results.pop(rid)
        return res[unique_reply_rank] if unique_reply_rank is not None else [res[r] for r in sorted(res)]

    def close(self, timeout_s: float = 30.0) -> None:
        if self._closed:
            return
        self._closed = True
        for q in self._inbox:
            try:
                q.put(SHUTDOWN)
            except Exception:  # noqa: BLE001
                pass
        for p in self._processes:
            p.join(timeout=timeout_s)
            if p.is_alive():
                p.terminate()

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close(timeout_s=2.0)
        except Exception:  # noqa: BLE001
            pass
