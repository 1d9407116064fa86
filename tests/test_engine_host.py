"""CPU: continuous step batching, the worker busy loop and the dispatching engine, on a fake pipeline (host logic only)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _req(seed, steps, T=3, n=1, hw=64, rid=None):
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    g = torch.Generator().manual_seed(100 + seed)
    return OmniDiffusionRequest(height=hw, width=hw, num_inference_steps=steps, seed=seed, num_outputs_per_prompt=n,
                                prompt_embeds=torch.randn(1, T, 8, generator=g), output_type="latent", request_id=rid)


def _solo(req):
    from _fake_pipeline import FakePipeline

    return FakePipeline().generate([req])[0].output


def test_continuous_step_batcher_staggered_arrivals_equal_solo_runs():
    from _fake_pipeline import FakePipeline
    from vllm_omni_amd.diffusion.step_batcher import ContinuousStepBatcher

    pipe = FakePipeline()
    b = ContinuousStepBatcher(pipe, max_items=3)
    reqs = {"a": _req(1, 5), "b": _req(2, 3, T=7), "c": _req(3, 4, n=2), "d": _req(4, 2, hw=128)}
    done = {}
    b.add(reqs["a"], "a")
    done.update(b.step())                       # a: step 0 alone
    b.add(reqs["b"], "b")                       # b joins while a is at step 1
    done.update(b.step())
    b.add(reqs["c"], "c")                       # two samples; only one fits next to a and b (cap 3)
    b.add(reqs["d"], "d")                       # different resolution: its own forwards, after the older group
    done.update(b.step())
    assert pipe.steps_run[1] == (("a", "b"), (1, 0)) and pipe.steps_run[2] == (("a", "b", "c"), (2, 1, 0))
    done.update(b.drain())
    assert set(done) == set(reqs) and not b.has_work() and b.outstanding_steps() == 0
    for k, r in reqs.items():
        assert torch.equal(done[k].output, _solo(r)), k
    assert done["c"].output.shape[0] == 2
    # a forward never mixes resolutions, never exceeds the cap, and step indices inside a forward may differ
    assert all(len(tags) <= 3 for tags, _ in pipe.steps_run)
    assert any(len(set(steps)) > 1 for _, steps in pipe.steps_run)
    with pytest.raises(ValueError):
        from vllm_omni_amd.diffusion.request import OmniDiffusionRequest
        b.add(OmniDiffusionRequest(height=64, width=64), "bad")      # admission error, batcher state untouched
    assert not b.has_work()


def test_engine_two_workers_dispatch_staggered_requests_and_rpc():
    """2 worker processes (gloo ranks), least-outstanding dispatch, requests submitted while others run, results equal
    solo runs, a failing request is isolated, broadcast RPC reaches every worker, clean shutdown."""
    import _fake_pipeline
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.diffusion_engine import DiffusionEngine
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    eng = DiffusionEngine(OmniDiffusionConfig(num_gpus=2, max_step_batch=3, dist_timeout=60),
                          pipeline_factory=_fake_pipeline.make, post_process_func=None, start_timeout_s=120)
    try:
        reqs = [_req(i, steps) for i, steps in enumerate([6, 2, 5, 3, 4, 2])]
        ids = [eng.submit(r) for r in reqs[:3]]
        ranks = [eng._cost[i][0] for i in ids]
        assert set(ranks) == {0, 1}                                       # spread, not replicated
        first = eng.poll(ids[1], timeout=60)                              # a short one finishes while long ones still run
        assert first is not None and first.error is None
        ids += [eng.submit(r) for r in reqs[3:]]                          # arrive mid-flight
        bad = eng.submit(OmniDiffusionRequest(height=64, width=64, num_inference_steps=2))     # no prompt: admission error
        outs = {ids[1]: first}
        for i in ids:
            if i not in outs:
                outs[i] = eng.poll(i, timeout=60)
        for i, r in zip(ids, reqs):
            assert outs[i].error is None and torch.equal(outs[i].output, _solo(r))
        err = eng.poll(bad, timeout=60)
        assert err.error is not None and "prompt" in err.error
        assert all(abs(x) < 1e-6 for x in eng._load)                      # every ticket was settled
        # blocking reference-shaped call: single request -> one OmniRequestOutput, several -> list
        one = eng.step([_req(9, 2, rid="r9")])
        assert one.request_id == "r9" and one.final_output_type == "image" and torch.equal(one.latents, _solo(_req(9, 2)))
        many = eng.step([_req(10, 2), _req(11, 3)])
        assert isinstance(many, list) and len(many) == 2
        assert eng.step([OmniDiffusionRequest(height=64, width=64)]) is None     # reference behaviour: logged, None
        res = eng.collective_rpc("batch_key_probe") if False else eng.collective_rpc("is_ready")
        assert res == [True, True]
    finally:
        eng.close()
    assert all(not p.is_alive() for p in eng._processes)


def test_async_omni_diffusion_concurrent_generates_are_served_together():
    """AsyncOmniDiffusion (reference entrypoints/async_omni_diffusion.py:117): concurrent `generate` calls resolve to their own
    results (request ids kept, outputs equal the solo runs), a failing request only fails its own awaitable, close() stops it."""
    import asyncio

    import _fake_pipeline
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.entrypoints.async_omni_diffusion import AsyncOmniDiffusion

    eng = AsyncOmniDiffusion(od_config=OmniDiffusionConfig(num_gpus=1, max_step_batch=3, dist_timeout=60),
                             pipeline_factory=_fake_pipeline.make)
    eng.engine.post_process_func = None
    g = torch.Generator().manual_seed(0)
    embeds = [torch.randn(1, 3 + i, 8, generator=g) for i in range(4)]

    async def run():
        calls = [eng.generate(prompt=f"p{i}", request_id=f"r{i}", num_inference_steps=3 + i, height=64, width=64, seed=i,
                              prompt_embeds=embeds[i], output_type="latent") for i in range(4)]
        calls.append(eng.generate(prompt=None, request_id="bad", num_inference_steps=2, height=64, width=64))   # nothing to encode
        return await asyncio.gather(*calls, return_exceptions=True)

    try:
        res = asyncio.run(run())
        assert eng.is_running
        for i in range(4):
            from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

            solo = _solo(OmniDiffusionRequest(height=64, width=64, num_inference_steps=3 + i, seed=i, prompt_embeds=embeds[i],
                                              output_type="latent"))
            assert res[i].request_id == f"r{i}" and res[i].metrics["num_inference_steps"] == 3 + i
            assert torch.equal(res[i].latents, solo)
        assert isinstance(res[4], RuntimeError)
    finally:
        eng.close()
    assert eng.is_stopped and all(not p.is_alive() for p in eng.engine._processes)
