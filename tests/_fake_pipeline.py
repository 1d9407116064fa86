"""CPU stand-in for QwenImagePipeline behind the step-batcher / worker / engine contracts (host-logic tests only).
'Denoising' is a per-sample recurrence that depends on the sample's own latents, prompt rows and step index — so any
cross-request leakage, wrong step index or lost update shows up as a wrong number."""
import torch

from vllm_omni_amd.diffusion.data import DiffusionOutput, OmniDiffusionConfig


class FakePipeline:
    device = torch.device("cpu")

    def __init__(self, od_config=None):
        self.od_config = od_config or OmniDiffusionConfig(max_step_batch=3)
        self.transformer = type("T", (), {"teacache": None})()
        self.steps_run = []                       # (tuple of tags, tuple of step indices) per forward

    def _req_params(self, r):
        if r.prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        return (r.height or 64, r.width or 64, r.num_inference_steps, 4.0, r.negative_prompt_embeds is not None)

    def resolve_request(self, req, index=0):
        h, w, steps, cfg, do_cfg = self._req_params(req)
        S = (h // 16) * (w // 16)
        n = int(req.num_outputs_per_prompt or 1)
        g = torch.Generator().manual_seed(req.seed or 0)
        return [dict(req=index, k=k, height=h, width=w, steps=steps, cfg=cfg, do_cfg=do_cfg, grid=(1, h // 16, w // 16),
                     lat=torch.randn(S, 4, generator=g), pos=req.prompt_embeds.reshape(-1, req.prompt_embeds.shape[-1]),
                     neg=None) for k in range(n)]

    def begin_sample(self, a):
        a.n_steps = a.sample["steps"]
        a.state = dict(lat=a.sample["lat"].clone())

    @staticmethod
    def batch_key(a):
        return (a.sample["grid"], a.sample["do_cfg"], a.sample["cfg"])

    def denoise_one_step(self, group):
        self.steps_run.append((tuple(a.tag for a in group), tuple(a.step for a in group)))
        for a in group:
            a.state["lat"] = a.state["lat"] * 0.9 + 0.01 * (a.step + 1) * a.sample["pos"].sum()

    def sample_result(self, a):
        return a.state["lat"].clone()

    def finish_request(self, req, latents, sample):
        return DiffusionOutput(output=torch.stack(latents))

    def generate(self, reqs, output_type="latent"):
        from vllm_omni_amd.diffusion.step_batcher import ContinuousStepBatcher

        outs = {}
        for i, r in enumerate(reqs):
            b = ContinuousStepBatcher(self, max_items=1)
            b.add(r, tag=i)
            outs[i] = b.drain()[0][1]
        return [outs[i] for i in range(len(reqs))]

    def decode_latents(self, lat, h, w):
        return lat


def make():
    return FakePipeline()
