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
    b.add(reqs["d"], "d")                       # different resolution: its own forwards, ROUND-ROBIN with the older group
    done.update(b.step())                       # the key that was never served goes first ...
    done.update(b.step())                       # ... then the 64x64 group again
    assert pipe.steps_run[1] == (("a", "b"), (1, 0))
    assert pipe.steps_run[2] == (("d",), (0,)) and pipe.steps_run[3] == (("a", "b", "c"), (2, 1, 0))
    done.update(b.step())
    assert pipe.steps_run[4] == (("d",), (1,))  # d (2 steps) is done after its second turn: it did not wait for a's 5 steps
    assert "d" in done and "a" not in done
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


def test_a_failing_finish_request_answers_that_request_and_no_other():
    """Round-3 advisor finding: the VAE decode runs inside step(), after the sample left `active` and `_pending`; if it raises
    (out of memory on a large image, say) the request must still get its (error) answer, the results collected in the same step
    must survive and unrelated requests must keep running."""
    from _fake_pipeline import FakePipeline
    from vllm_omni_amd.diffusion.step_batcher import ContinuousStepBatcher

    class Flaky(FakePipeline):
        def finish_request(self, req, latents, sample):
            if req.request_id == "boom":
                raise MemoryError("decode failed")
            return super().finish_request(req, latents, sample)

    pipe = Flaky()
    b = ContinuousStepBatcher(pipe, max_items=4)
    reqs = {"ok1": _req(1, 2), "boom": _req(2, 2, rid="boom"), "ok2": _req(3, 2), "late": _req(4, 4)}
    for k, r in reqs.items():
        b.add(r, k)
    done = dict(b.drain())
    assert set(done) == set(reqs)                                        # every request is answered exactly once
    assert done["boom"].error and "MemoryError" in done["boom"].error
    for k in ("ok1", "ok2", "late"):                                     # same step as the failure / still running at that time
        assert done[k].error is None and torch.equal(done[k].output, _solo(reqs[k])), k
    assert not b.has_work() and not b._pending


def test_a_failing_sample_of_a_two_sample_request_aborts_it_once_and_spares_the_rest():
    """Two samples of ONE request (num_outputs_per_prompt = 2) finish in the same step and the first one's `sample_result`
    raises: the abort removes the sibling from `active` and the request from `_pending` — the loop over the step's group must
    skip that sibling (it used to raise ValueError out of step(), which the worker answers by aborting EVERY request)."""
    from _fake_pipeline import FakePipeline
    from vllm_omni_amd.diffusion.step_batcher import ContinuousStepBatcher

    class Flaky(FakePipeline):
        def sample_result(self, a):
            if a.tag == "boom" and a.sample["k"] == 0:
                raise RuntimeError("device fault")
            return super().sample_result(a)

    b = ContinuousStepBatcher(Flaky(), max_items=4)
    reqs = {"boom": _req(1, 2, n=2), "ok": _req(2, 2), "late": _req(3, 3)}
    for k, r in reqs.items():
        b.add(r, k)
    done = b.drain()
    assert sorted(t for t, _ in done) == sorted(reqs)                    # one answer per request, none twice
    done = dict(done)
    assert done["boom"].error and "device fault" in done["boom"].error
    for k in ("ok", "late"):
        assert done[k].error is None and torch.equal(done[k].output, _solo(reqs[k])), k
    assert not b.has_work() and not b._pending


def test_batch_keys_are_served_round_robin():
    """Mixed-resolution traffic: the forwards alternate between the keys instead of finishing the head group's whole loop first
    (round-3 verdict item 13), and more samples than `max_samples` of one key queue FIFO for the first free slot."""
    from _fake_pipeline import FakePipeline
    from vllm_omni_amd.diffusion.step_batcher import ContinuousStepBatcher

    pipe = FakePipeline()
    b = ContinuousStepBatcher(pipe, max_items=2)
    b.add(_req(1, 4), "s1"); b.add(_req(2, 4), "s2"); b.add(_req(3, 2), "s3")     # three at 64x64: s3 waits for a slot
    b.add(_req(4, 3, hw=128), "L1")                                               # one at 128x128
    b.add(_req(5, 3, hw=256), "X1")                                               # one at 256x256
    done = dict(b.drain())
    order = [tags for tags, _ in pipe.steps_run]
    assert order[:6] == [("s1", "s2"), ("L1",), ("X1",), ("s1", "s2"), ("L1",), ("X1",)]
    assert order.index(("s3",)) > order.index(("s1", "s2")) and ("s1", "s3") not in order      # FIFO: joined when s1, s2 left
    for k, r in {"s1": _req(1, 4), "s3": _req(3, 2), "L1": _req(4, 3, hw=128), "X1": _req(5, 3, hw=256)}.items():
        assert torch.equal(done[k].output, _solo(r)), k


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


def test_unseeded_request_gets_one_seed_before_the_sequence_parallel_fan_out():
    """Round-3 advisor finding: a request without seed / generator / latents (the default of the entry points) was fanned out to
    every rank of a sequence-parallel group, each rank drew its own initial noise, and the all-gathered predictions mixed
    different images.  The engine now makes the seed concrete once, before the fan-out; seeded requests are left alone."""
    import itertools

    from vllm_omni_amd.diffusion.diffusion_engine import DiffusionEngine
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    class Q(list):
        def put(self, m):
            self.append(m)

    eng = DiffusionEngine.__new__(DiffusionEngine)
    eng.sp_degree, eng.num_groups, eng._ids, eng._load, eng._cost = 2, 2, itertools.count(), [0.0, 0.0], {}
    eng._inbox = [Q(), Q(), Q(), Q()]
    eng._closed = True
    r = OmniDiffusionRequest(height=64, width=64, num_inference_steps=2, prompt_embeds=torch.zeros(1, 1, 8))
    assert r.seed is None
    eng.submit(r)
    a, b = eng._inbox[0][0]["request"], eng._inbox[1][0]["request"]
    assert a.seed is not None and a.seed == b.seed and not eng._inbox[2]
    r2 = OmniDiffusionRequest(height=64, width=64, num_inference_steps=2, prompt_embeds=torch.zeros(1, 1, 8), seed=7)
    eng.submit(r2)
    assert eng._inbox[2][0]["request"].seed == 7 and eng._inbox[3][0]["request"].seed == 7
    eng.sp_degree, eng.num_groups, eng._load = 1, 4, [0.0] * 4       # plain data parallel: one rank owns the request, nothing to agree on
    r3 = OmniDiffusionRequest(height=64, width=64, num_inference_steps=2, prompt_embeds=torch.zeros(1, 1, 8))
    eng.submit(r3)
    assert r3.seed is None


def test_request_accepts_the_reference_spellings_of_picture_and_variant_inputs():
    """A caller written against the reference passes `pil_image=`, `layers=`, `resolution=`, `cfg_normalize=`, `use_en_prompt=`
    (vllm_omni/diffusion/request.py:33-38,70-79; read at pipeline_qwen_image_edit.py:64, pipeline_qwen_image_layered.py:69,668-671)
    and the reference's default `prompt_embeds=[]`; this build's pipelines read `extra[...]`.  Both spellings must reach the same
    code, an explicit `extra` entry wins, and the entry points keep the fields instead of dropping them as unknown kwargs."""
    from PIL import Image

    from vllm_omni_amd.diffusion.diffusion_engine import _request_to_cpu
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image_edit_plus import QwenImageEditPlusPipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest
    from vllm_omni_amd.entrypoints.omni_diffusion import prepare_requests

    pic = Image.new("RGB", (640, 400))
    r = prepare_requests("a prompt", pil_image=pic, layers=3, resolution=1024, cfg_normalize=True, use_en_prompt=True,
                         prompt_embeds=[], no_such_field=1)
    assert r.extra == {"image": pic, "layers": 3, "resolution": 1024, "cfg_normalize": True, "use_en_prompt": True}
    assert r.prompt_embeds is None and r.pil_image is pic
    assert OmniDiffusionRequest(pil_image=pic, extra={"image": "explicit"}).extra["image"] == "explicit"
    t = torch.zeros(1, 3, 8, 8)
    assert OmniDiffusionRequest(pil_image=pic, preprocessed_image=t).extra["image"] is t      # the pre-processed picture wins
    one = torch.zeros(1, 4, 8)
    assert OmniDiffusionRequest(prompt_embeds=[one]).prompt_embeds is one
    # Edit-Plus: a LIST of pictures in `pil_image` is what `_prompt_pictures` resizes for the vision tower
    shell = QwenImageEditPlusPipeline.__new__(QwenImageEditPlusPipeline)
    got = shell._prompt_pictures(OmniDiffusionRequest(prompt="x", pil_image=[pic, Image.new("RGB", (300, 300))]))
    assert len(got) == 2
    # the engine sends host objects only: pictures pass through, tensors (also inside lists in `extra`) stay host tensors
    q = _request_to_cpu(OmniDiffusionRequest(prompt="x", pil_image=[pic], extra={"image_latents": [torch.zeros(2, 64)]}))
    assert q.pil_image == [pic] and not q.extra["image_latents"][0].is_cuda
