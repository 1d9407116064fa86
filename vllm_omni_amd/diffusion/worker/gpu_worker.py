"""GPUWorker for MI355X — mirror of vllm_omni/diffusion/worker/gpu_worker.py:32-137.

Same shape: one process per GPU, `init_device_and_model()` sets the device, joins the process group and builds
the pipeline; `execute_model(reqs, od_config)` runs a batch under inference mode and returns a DiffusionOutput;
runtime errors are stringified into `DiffusionOutput.error` (:266-274).

What is new: the reference executes only `reqs[0]` and runs it on EVERY rank (:128-130; F6/F7).  Here all
requests are used: they are sharded across the DP ranks (least-loaded), each rank step-batches its share,
finished latents are all-gathered over RCCL, and rank `output_rank` VAE-decodes / returns them.
"""
from __future__ import annotations

import os
import queue
import traceback

import torch

from ..data import DiffusionOutput, OmniDiffusionConfig
from ..distributed import data_parallel as dp
from ..request import OmniDiffusionRequest


class GPUWorker:
    def __init__(self, local_rank: int, rank: int, od_config: OmniDiffusionConfig, pipeline=None):
        self.local_rank, self.rank, self.od_config = local_rank, rank, od_config
        self.pipeline = pipeline
        self.world = 1
        self.sp_degree, self.sp_group, self.dp_rank, self.dp_world = 1, None, rank, 1

    def init_device_and_model(self, pipeline_factory=None) -> None:
        devs = getattr(self.od_config, "devices", None)
        if devs:                                             # explicit rank -> device map (reference: runtime.devices)
            self.local_rank = int(devs[self.rank % len(devs)])
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
        self.rank, self.world, _ = dp.init_distributed(backend=getattr(self.od_config, "dist_backend", None),
                                                       timeout_s=self.od_config.dist_timeout, local_device=self.local_rank)
        # world = data-parallel groups x ulysses_degree (reference DiffusionParallelConfig); consecutive ranks form an SP group
        self.sp_degree = int(getattr(self.od_config.parallel_config, "ulysses_degree", 1) or 1)
        if self.sp_degree > 1 and self.world % self.sp_degree:
            raise ValueError(f"{self.world} ranks cannot form groups of ulysses_degree {self.sp_degree}")
        self.sp_group, self.dp_rank, self.dp_world = dp.init_sp_groups(self.world, self.sp_degree, self.rank)
        if self.pipeline is None:
            if pipeline_factory is None:
                import os

                dev = torch.device("cuda", self.local_rank)
                if self.od_config.model and os.path.isdir(self.od_config.model):
                    # a diffusers-layout checkpoint directory: registry class + weights from transformer/ and vae/
                    # (reference gpu_worker.py:100-113 -> DiffusersPipelineLoader.load_model)
                    from ..model_loader import DiffusersPipelineLoader

                    pipeline_factory = lambda: DiffusersPipelineLoader().load_model(self.od_config, dev)  # noqa: E731
                else:
                    from ..registry import initialize_model  # arch name -> pipeline class (reference registry.py:81-94)

                    pipeline_factory = lambda: initialize_model(self.od_config, device=dev)  # noqa: E731
            self.pipeline = pipeline_factory()
            # vae_use_slicing / vae_use_tiling reach the VAE whichever factory built the pipeline (reference registry.py:88-92 does
            # it inside initialize_model; a user `pipeline_factory` bypasses that): a flag that cannot be honoured is refused
            # loudly instead of running a 2048^2 decode untiled
            from ..registry import apply_vae_memory_flags

            want = bool(getattr(self.od_config, "vae_use_slicing", False) or getattr(self.od_config, "vae_use_tiling", False))
            vae = getattr(self.pipeline, "vae", None)
            if want and (vae is None or not (hasattr(vae, "use_slicing") and hasattr(vae, "use_tiling"))):
                raise NotImplementedError(f"{type(self.pipeline).__name__}'s VAE has no use_slicing / use_tiling switches: "
                                          "vae_use_slicing / vae_use_tiling cannot be honoured for this pipeline")
            if want:
                apply_vae_memory_flags(self.pipeline, self.od_config)
        if self.sp_degree > 1:
            if not hasattr(self.pipeline, "sp_degree"):
                raise NotImplementedError(f"{type(self.pipeline).__name__} has no sequence-parallel denoise loop")
            self.pipeline.sp_group, self.pipeline.sp_degree = self.sp_group, self.sp_degree

    def is_ready(self) -> bool:
        return self.pipeline is not None

    def generate(self, requests: list[OmniDiffusionRequest]) -> DiffusionOutput:
        """Reference name for `execute_model(requests, self.od_config)` (gpu_worker.py:109-119)."""
        return self.execute_model(requests, self.od_config)

    def shutdown(self) -> None:
        """Leave the process group (reference :139-140 `destroy_distributed_env`)."""
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()

    @torch.inference_mode()
    def execute_model(self, reqs: list[OmniDiffusionRequest], od_config: OmniDiffusionConfig | None = None,
                      output_rank: int = 0, decode: bool = True) -> DiffusionOutput:
        """Every rank receives the same request list (broadcast RPC, as in the reference), serves its shard and joins ONE
        all-gather of the finished latents.  A failure on one rank must not strand the others inside the collective
        (the reference has no collective here, so it has no such hazard): requests are validated on EVERY rank before
        sharding (same verdict everywhere), and after the local denoise the ranks agree on an error flag (a 4-byte
        all-reduce) before anyone enters the gather."""
        try:
            if not reqs:
                return DiffusionOutput(error="empty request list")
            for r in reqs:                                   # identical on all ranks: raises everywhere or nowhere
                self.pipeline._req_params(r)
                if self.world > 1 and ((r.num_outputs_per_prompt or 1) != 1 or (isinstance(r.prompt, list) and len(r.prompt) != 1)):
                    raise NotImplementedError("a data-parallel batch carries one sample per request (one gather row each)")
            pl = self.pipeline
            # gather key of a request: (height, width, packed-latent rows per sample, images per sample).  The text-to-image
            # and Edit pipelines finish with (h/16)(w/16) rows and one image; the Layered pipeline with (layers + 1) frames of
            # that many rows and one image per layer — every rank derives the key from the request alone.
            shape_of = []
            for r in reqs:
                h, w = (r.height or 1024), (r.width or 1024)
                rows = int(pl.latent_rows(r)) if hasattr(pl, "latent_rows") else (h // 16) * (w // 16)
                n_img = int(pl.images_per_sample(r)) if hasattr(pl, "images_per_sample") else 1
                shape_of.append((h, w, rows, n_img))
            costs = [float((r.num_inference_steps or 50) * rows) for r, (_h, _w, rows, _n) in zip(reqs, shape_of)]
            # requests are sharded over the data-parallel GROUPS; the ranks of a sequence-parallel group run the same share in
            # lockstep and only the group's first rank contributes rows to the gather
            dp_assign = dp.shard_requests(costs, self.dp_world)
            P = self.sp_degree
            assign = [dp_assign[r // P] if r % P == 0 else [] for r in range(self.world)]
        except Exception as e:  # same policy as the reference busy loop: report, do not kill the worker
            return DiffusionOutput(error=f"{type(e).__name__}: {e}")
        mine_idx = dp_assign[self.dp_rank]
        mine = [reqs[i] for i in mine_idx]
        err, outs = None, []
        try:
            cb = getattr(self.pipeline, "cache_backend", None)
            if mine and cb is not None and getattr(cb, "enabled", False):      # reference gpu_worker.py:132-134
                cb.refresh(self.pipeline, mine[0].num_inference_steps or 50)
            outs = self.pipeline.generate(mine, output_type="latent") if mine else []
        except Exception as e:
            err = f"rank {self.rank}: {type(e).__name__}: {e}"
        failed = dp.any_rank_failed(err is not None, self.pipeline.device)
        if failed:
            return DiffusionOutput(error=err or "another data-parallel rank failed; batch aborted on all ranks")
        try:
            dev = self.pipeline.device
            contributes = self.rank % self.sp_degree == 0    # not the group's first rank: nothing to contribute
            mine_out = dict(zip(mine_idx, outs)) if contributes else {}
            lat: list[torch.Tensor | None] = [None] * len(reqs)
            imgs: list[torch.Tensor | None] = [None] * len(reqs)
            # one gather per key (a gather needs one row shape), in sorted order: the same sequence of collectives on every
            # rank.  Mixed-resolution batches are legal (the reference has no collective here, so no such constraint).
            for key in sorted(set(shape_of)):
                h, w, S, n_img = key
                sub = [[i for i in a if shape_of[i] == key] for a in assign]
                my = sub[self.rank]
                local = (torch.cat([mine_out[i].output.reshape(-1, S, 64) for i in my]) if my
                         else torch.empty((0, S, 64), dtype=torch.bfloat16, device=dev))
                gathered = dp.gather_latents(local.contiguous(), [len(a) for a in sub])
                rows = dp.unshard_indexed(gathered, sub)
                for i, t in rows.items():
                    lat[i] = t
                if not decode:
                    continue
                # every rank now holds every latent of this key: the VAE decodes are dealt round-robin over the ranks
                # (round 3: `output_rank` decoded all of them serially) and the pixels return in one more gather.  The decode
                # is fallible (a 1024^2 decode needs ~1 GB of activations): like the denoise phase above, the ranks agree on
                # a failure flag BEFORE anyone enters the pixel gather, so a rank whose decode raised cannot strand the others
                # inside the collective.
                order = sorted(rows)
                deal = [[i for n, i in enumerate(order) if n % self.world == r] for r in range(self.world)]
                px, derr = [], None
                try:
                    px = [self._decode_one(reqs[i], lat[i].unsqueeze(0), h, w) for i in deal[self.rank]]
                    for t in px:                      # a wrong element count must surface BEFORE the barrier below, not as a
                        if t.numel() != n_img * 3 * h * w:   # reshape error on one rank while the others sit in the gather
                            raise ValueError(f"decode returned {tuple(t.shape)} for a sample of {n_img} x 3 x {h} x {w}")
                except Exception as e:  # noqa: BLE001
                    derr = f"rank {self.rank}: decode failed: {type(e).__name__}: {e}"
                if dp.any_rank_failed(derr is not None, dev):
                    return DiffusionOutput(error=derr or "another data-parallel rank failed to decode; batch aborted on all ranks")
                loc = (torch.cat(px).reshape(len(px), 1, -1) if px
                       else torch.empty((0, 1, n_img * 3 * h * w), dtype=torch.bfloat16, device=dev))
                allpx = dp.gather_latents(loc.to(torch.bfloat16).contiguous(), [len(d) for d in deal])
                for i, t in dp.unshard_indexed(allpx, deal).items():
                    imgs[i] = t.reshape(3, h, w) if n_img == 1 else t.reshape(n_img, 3, h, w)
            if self.rank != output_rank:
                return DiffusionOutput(output=None)
            res = lat if not decode else imgs
            if len(set(shape_of)) == 1:
                return DiffusionOutput(output=torch.stack(res))
            return DiffusionOutput(output=res)              # mixed resolutions: a list, request order
        except Exception as e:
            return DiffusionOutput(error=f"{type(e).__name__}: {e}")

    def _decode_one(self, req: OmniDiffusionRequest, lat: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """Packed latents of ONE sample [1, rows, 64] -> its image(s) [n_img, 3, h, w] through the pipeline's own decode
        (`decode_request` where a pipeline's samples are not one (h/16)(w/16)-row image: the Layered variant)."""
        if hasattr(self.pipeline, "decode_request"):
            return self.pipeline.decode_request(req, lat)
        return self.pipeline.decode_latents(lat, h, w)


# ---------------------------------------------------------------------------------------------------------------------
# Worker process: one per GPU (reference WorkerProc, vllm_omni/diffusion/worker/gpu_worker.py:143-314).  The reference's
# busy loop dequeues ONE broadcast RPC at a time from vLLM's shm MessageQueue and runs it to completion; here the loop is
# the scheduling quantum of the continuous step batcher: between two denoising steps it drains its inbox, so newly arrived
# requests join the running batch and finished ones are returned at once.  Queues are torch.multiprocessing queues
# (CPU tensors only cross the process boundary); vLLM's MessageQueue / zmq are not part of this build.
# ---------------------------------------------------------------------------------------------------------------------
SHUTDOWN = {"type": "shutdown"}


def _to_cpu(x):
    """Device tensors -> host tensors, through lists / tuples / DiffusionOutput (a mixed-resolution `execute_model` returns a LIST
    of device tensors; only CPU tensors may cross the process boundary)."""
    if isinstance(x, torch.Tensor):
        return x.detach().to("cpu")
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cpu(v) for v in x)
    if isinstance(x, DiffusionOutput):
        return DiffusionOutput(output=_to_cpu(x.output), error=x.error, trajectory_timesteps=x.trajectory_timesteps,
                               trajectory_latents=_to_cpu(x.trajectory_latents))
    return x


class WorkerProc:
    def __init__(self, rank: int, od_config: OmniDiffusionConfig, inbox, outbox, pipeline=None, pipeline_factory=None):
        from ..step_batcher import ContinuousStepBatcher

        self.rank, self.inbox, self.outbox = rank, inbox, outbox
        self.worker = GPUWorker(local_rank=rank, rank=rank, od_config=od_config, pipeline=pipeline)
        self.worker.init_device_and_model(pipeline_factory)
        self.batcher = ContinuousStepBatcher(self.worker.pipeline, max_items=od_config.max_step_batch,
                                             max_steps_in_flight=getattr(od_config, "max_steps_in_flight", 2))
        # results on their way to the host: (message, event).  A finished request's image is copied to PINNED memory with a
        # non-blocking copy and handed to the result queue when the copy's event has completed — the loop never blocks on a
        # device-to-host copy (round 3: one blocking `.to("cpu")` per finished request, behind everything already queued).
        self._outgoing: list[tuple[dict, object]] = []

    # -- results ----------------------------------------------------------------------------------------------------
    def _stage(self, x):
        """Device tensors -> pinned host tensors, asynchronously (the caller records ONE event after staging a message)."""
        if isinstance(x, torch.Tensor):
            if not x.is_cuda:
                return x.detach()
            host = torch.empty(x.shape, dtype=x.dtype, device="cpu", pin_memory=True)
            host.copy_(x.detach(), non_blocking=True)
            return host
        if isinstance(x, (list, tuple)):
            return type(x)(self._stage(v) for v in x)
        if isinstance(x, DiffusionOutput):
            return DiffusionOutput(output=self._stage(x.output), error=x.error, trajectory_timesteps=x.trajectory_timesteps,
                                   trajectory_latents=self._stage(x.trajectory_latents))
        return x

    def _send(self, msg: dict, key: str = "output") -> None:
        if torch.cuda.is_available() and getattr(self.worker.pipeline, "device", torch.device("cpu")).type == "cuda":
            msg[key] = self._stage(msg[key])
            ev = torch.cuda.Event()
            ev.record()
            self._outgoing.append((msg, ev))
        else:
            msg[key] = _to_cpu(msg[key])
            self.outbox.put(msg)

    def _flush(self, block: bool = False) -> None:
        """Hand every message whose copy has completed to the result queue, in order."""
        while self._outgoing:
            msg, ev = self._outgoing[0]
            if block:
                ev.synchronize()
            elif not ev.query():
                return
            self._outgoing.pop(0)
            self.outbox.put(msg)

    # -- message handling -------------------------------------------------------------------------------------------
    def _handle(self, msg) -> bool:
        """Returns False on shutdown."""
        kind = msg.get("type")
        if kind == "shutdown":
            return False
        if kind == "add" and self.worker.sp_degree > 1:
            # sequence-parallel group: every rank of the group receives the same requests in the same order and runs each to
            # completion in lockstep (a continuous batcher would have to agree on every step's composition across ranks);
            # the group's first rank answers
            try:
                req = msg["request"]
                if req.seed is None and req.generator is None and req.latents is None:
                    # every rank of the group would draw its OWN initial noise (the engine seeds such requests before the
                    # fan-out; a request that reaches a worker unseeded some other way must not silently produce garbage)
                    raise ValueError("a sequence-parallel request needs seed, generator or latents: the group's ranks must "
                                     "start from the same noise")
                out = self.worker.pipeline.generate([req])[0]
            except Exception as e:  # noqa: BLE001
                out = DiffusionOutput(error=f"{type(e).__name__}: {e}")
            if self.rank % self.worker.sp_degree == 0:
                self._send({"type": "done", "id": msg["id"], "rank": self.rank, "output": out, "outstanding_steps": 0})
        elif kind == "add":
            try:
                self.batcher.add(msg["request"], tag=msg["id"])
            except Exception as e:  # admission errors are per request: report, keep serving (reference :266-274)
                self.outbox.put({"type": "done", "id": msg["id"], "rank": self.rank,
                                 "output": DiffusionOutput(error=f"{type(e).__name__}: {e}")})
        elif kind == "rpc":
            try:
                if msg["method"] == "serving_stats":        # this process's step batcher (bench.py: per-worker busy fraction)
                    fn = self.batcher.stats
                else:
                    fn = getattr(self.worker, msg["method"], None) or getattr(self.worker.pipeline, msg["method"])
                res = fn(*msg.get("args", ()), **msg.get("kwargs", {}))
                if msg.get("output_rank") is None or msg["output_rank"] == self.rank:
                    self.outbox.put({"type": "rpc_result", "id": msg["id"], "rank": self.rank, "result": _to_cpu(res)})
            except Exception as e:
                self.outbox.put({"type": "rpc_result", "id": msg["id"], "rank": self.rank,
                                 "result": DiffusionOutput(error=f"{type(e).__name__}: {e}\n{traceback.format_exc()}")})
        else:
            self.outbox.put({"type": "error", "rank": self.rank, "error": f"unknown message type {kind!r}"})
        return True

    POLL_S = 0.001        # inbox wait while the device is busy (throttled) or results are in flight

    def worker_busy_loop(self) -> None:
        """idle: block for work.  busy: look at the inbox between two denoising steps (requests that arrive now join the batch
        at the next step), enqueue the next step only while fewer than `max_steps_in_flight` are queued on the device, and hand
        finished results over as their host copies complete."""
        while True:
            self._flush()
            busy = self.batcher.has_work() or bool(self._outgoing)
            throttled = self.batcher.has_work() and not self.batcher.ready()
            try:
                if not busy:
                    msg = self.inbox.get(timeout=None)
                elif throttled or not self.batcher.has_work():
                    msg = self.inbox.get(timeout=self.POLL_S)   # nothing to enqueue right now: wait here, not in a spin
                else:
                    msg = self.inbox.get_nowait()
                if not self._handle(msg):
                    self._flush(block=True)
                    return
                continue                                        # keep draining before the next step
            except queue.Empty:
                pass
            if throttled or not self.batcher.has_work():
                continue
            try:
                finished = self.batcher.step()              # a failing step aborts only the requests that were in it
            except Exception as e:  # noqa: BLE001 — scheduler bug: abort everything once per request, keep the worker alive
                finished = self.batcher.abort({a.tag for a in self.batcher.active} | set(self.batcher._pending),
                                              f"{type(e).__name__}: {e}")
            for tag, out in finished:
                self._send({"type": "done", "id": tag, "rank": self.rank, "output": out,
                            "outstanding_steps": self.batcher.outstanding_steps()})

    @staticmethod
    def worker_main(rank: int, world: int, od_config: OmniDiffusionConfig, inbox, outbox, ready, pipeline_factory=None,
                    master_port: int = 29533) -> None:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(master_port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            if torch.cuda.is_available():
                from ..distributed.numa import pin_to_gpu_numa

                devs = getattr(od_config, "devices", None)
                pin_to_gpu_numa(int(devs[rank % len(devs)]) if devs else rank)                       # host threads next to this rank's GPU (distributed/numa.py)
            proc = WorkerProc(rank, od_config, inbox, outbox, pipeline_factory=pipeline_factory)
        except Exception as e:
            ready.put({"rank": rank, "status": "failed", "error": f"{type(e).__name__}: {e}\n{traceback.format_exc()}"})
            return
        ready.put({"rank": rank, "status": "ready"})
        try:
            proc.worker_busy_loop()
        finally:
            if torch.distributed.is_initialized():
                torch.distributed.destroy_process_group()
