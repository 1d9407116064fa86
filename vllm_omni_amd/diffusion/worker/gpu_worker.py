"""GPUWorker for MI355X — mirror of vllm_omni/diffusion/worker/gpu_worker.py:32-137.

Same shape: one process per GPU, `init_device_and_model()` sets the device, joins the process group and builds
the pipeline; `execute_model(reqs, od_config)` runs a batch under inference mode and returns a DiffusionOutput;
runtime errors are stringified into `DiffusionOutput.error` (:266-274).

What is new: the reference executes only `reqs[0]` and runs it on EVERY rank (:128-130; F6/F7).  Here all
requests are used: they are sharded across the DP ranks (least-loaded), each rank step-batches its share,
finished latents are all-gathered over RCCL, and rank `output_rank` VAE-decodes / returns them.
"""
from __future__ import annotations

import torch

from ..data import DiffusionOutput, OmniDiffusionConfig
from ..distributed import data_parallel as dp
from ..request import OmniDiffusionRequest


class GPUWorker:
    def __init__(self, local_rank: int, rank: int, od_config: OmniDiffusionConfig, pipeline=None):
        self.local_rank, self.rank, self.od_config = local_rank, rank, od_config
        self.pipeline = pipeline
        self.world = 1

    def init_device_and_model(self, pipeline_factory=None) -> None:
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
        self.rank, self.world, _ = dp.init_distributed(timeout_s=self.od_config.dist_timeout)
        if self.pipeline is None:
            if pipeline_factory is None:
                from ..models.qwen_image.pipeline_qwen_image import QwenImagePipeline

                pipeline_factory = lambda: QwenImagePipeline(od_config=self.od_config,  # noqa: E731
                                                             device=torch.device("cuda", self.local_rank))
            self.pipeline = pipeline_factory()

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
            shapes = {((r.height or 1024), (r.width or 1024)) for r in reqs}
            if self.world > 1 and len(shapes) != 1:
                raise NotImplementedError("a DP batch must share one resolution (one gather shape)")
            costs = [float((r.num_inference_steps or 50) * ((r.height or 1024) // 16) * ((r.width or 1024) // 16))
                     for r in reqs]
            assign = dp.shard_requests(costs, self.world)
        except Exception as e:  # same policy as the reference busy loop: report, do not kill the worker
            return DiffusionOutput(error=f"{type(e).__name__}: {e}")
        mine = [reqs[i] for i in assign[self.rank]]
        err, outs = None, []
        try:
            outs = self.pipeline.generate(mine, output_type="latent") if mine else []
        except Exception as e:
            err = f"rank {self.rank}: {type(e).__name__}: {e}"
        failed = dp.any_rank_failed(err is not None, self.pipeline.device)
        if failed:
            return DiffusionOutput(error=err or "another data-parallel rank failed; batch aborted on all ranks")
        try:
            h, w = next(iter(shapes))
            S = (h // 16) * (w // 16)
            dev = self.pipeline.device
            local = torch.cat([o.output for o in outs]) if outs else torch.empty((0, S, 64), dtype=torch.bfloat16, device=dev)
            gathered = dp.gather_latents(local.contiguous(), [len(a) for a in assign])
            lat = torch.stack(dp.unshard(gathered, assign))
            if self.rank != output_rank:
                return DiffusionOutput(output=None)
            if not decode:
                return DiffusionOutput(output=lat)
            imgs = [self.pipeline.decode_latents(lat[i:i + 1], h, w) for i in range(lat.shape[0])]
            return DiffusionOutput(output=torch.cat(imgs))
        except Exception as e:
            return DiffusionOutput(error=f"{type(e).__name__}: {e}")
