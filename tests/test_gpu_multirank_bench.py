"""GPU (one device is enough): the N-rank control flow of bench.py and of the serving engine, everything except RCCL itself.

No box this repository has seen so far had more than one GPU, so `bench.py --gpus N` (self-launch, env:// rendezvous, per-rank
HIP-event statistics, the latent gather, value = N*K*R / max-over-ranks seconds, the N-worker serving line) had never executed
anywhere before the driver's own scaling run.  Here both ranks share device 0 and the process group is gloo (RCCL refuses two
ranks on one GPU): the same code path with the collective staged through the host."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_device_over_gloo():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "1", "--warmup", "1",
           "--requests", "2", "--no-cpu-baseline", "--dist-backend", "gloo", "--share-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                                 # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["warmup"] == 1 and j["scaling"] == "weak" and j["finite_outputs"]
    assert j["config"]["global_batch"] == 4 and j["config"]["parallelism"] == "dp2"
    assert j["value"] > 0 and abs(j["value"] - 2 * 1 * 2 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]   # N*K*R / seconds
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    for r in pr:
        assert r["denoise_s"] > 0 and r["vae_decode_s"] > 0 and r["gather_s"] >= 0 and r["seconds"] <= j["ms_per_step"] * 1e-3 * 1.001
    assert "roofline" in j and j["roofline"]["bound"] == "mfma"
    # the record proves what the collective carried: every rank received, in slot r, exactly the rows rank r contributed
    col = j["collective"]
    assert col["world"] == 2 and col["backend"] == "gloo" and col["gathered_checksum_matches_all_ranks"] and col["per_rank_payloads_distinct"]
    # the serving topology at N = 2: two worker processes, dispatched requests, per-worker device-busy fractions
    sec = j.get("secondary", {})
    assert "engine_error" not in sec, sec.get("engine_error")
    assert sec["engine_images_per_sec"] > 0 and len(sec["engine_workers"]) == 2
    for w in sec["engine_workers"]:
        assert w["steps"] > 0 and 0 < w["busy_frac"] <= 1.0 + 1e-6 and 1.0 <= w["mean_batch"] <= 2.0


def test_bench_two_ranks_under_torch_distributed_run():
    """The launcher the DRIVER uses for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (ranks, local ranks and the rendezvous come from its environment, not from bench.py's own
    self-launch) — on one device over gloo."""
    from test_host_logic import _free_port

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "1", "--steps", "1",
           "--warmup", "1", "--requests", "1", "--no-cpu-baseline", "--no-engine", "--dist-backend", "gloo", "--share-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["finite_outputs"]
    assert j["collective"]["world"] == 2 and j["collective"]["gathered_checksum_matches_all_ranks"]
    assert [r["rank"] for r in j["per_rank"]] == [0, 1]


def test_bench_eight_ranks_on_one_device_over_gloo():
    """The REAL width of the driver's scaling run (8 ranks: ports, spawn, NUMA pinning, the 8-way gather and its checksum proof,
    max-over-ranks timing) on one device with one layer; the serving line is skipped (eight worker processes x their own weights
    add nothing the 2-rank test does not cover).  Reference spawn shape: vllm_omni/diffusion/diffusion_engine.py:211-270."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--layers", "1", "--steps", "1", "--warmup", "1",
           "--requests", "1", "--no-cpu-baseline", "--no-engine", "--dist-backend", "gloo", "--share-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 8 and j["config"]["parallelism"] == "dp8" and j["finite_outputs"]
    assert [r["rank"] for r in j["per_rank"]] == list(range(8))
    assert all(r["denoise_s"] > 0 and r["vae_decode_s"] > 0 for r in j["per_rank"])
    col = j["collective"]
    assert col["world"] == 8 and col["gathered_checksum_matches_all_ranks"] and col["per_rank_payloads_distinct"]
    assert abs(j["value"] - 8 * 1 * 1 / (j["ms_per_step"] * 1e-3)) < 1e-6 * j["value"]


def test_engine_two_workers_share_one_device():
    """DiffusionEngine with num_gpus = 2, devices = [0, 0], gloo: dispatch + continuous batching + result return through two
    worker processes on real kernels; every request equals its solo run."""
    import _gpu_factory
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig
    from vllm_omni_amd.diffusion.diffusion_engine import DiffusionEngine
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    BF16 = torch.bfloat16
    g = torch.Generator().manual_seed(3)

    def req(steps):
        return OmniDiffusionRequest(height=128, width=128, num_inference_steps=steps, true_cfg_scale=4.0, output_type="latent",
                                    latents=torch.randn(1, 64, 64, generator=g).to(BF16),
                                    prompt_embeds=torch.randn(1, 9, 128, generator=g).to(BF16),
                                    negative_prompt_embeds=torch.randn(1, 5, 128, generator=g).to(BF16))

    reqs = [req(s) for s in (4, 2, 3, 5, 2, 3)]
    solo_pipe = _gpu_factory.make_small_pipeline()
    import copy

    solo = [solo_pipe.generate([copy.deepcopy(r)], output_type="latent")[0].output.float().cpu() for r in reqs]
    eng = DiffusionEngine(OmniDiffusionConfig(num_gpus=2, devices=[0, 0], dist_backend="gloo", max_step_batch=2, dist_timeout=120),
                          pipeline_factory=_gpu_factory.make_small_pipeline, post_process_func=None, start_timeout_s=300)
    try:
        ids = [eng.submit(r) for r in reqs]
        assert {eng._cost[i][0] for i in ids} == {0, 1}
        outs = [eng.poll(i, timeout=300) for i in ids]
        for o, s in zip(outs, solo):
            assert o is not None and o.error is None
            err = float((o.output.float().cpu() - s).norm() / s.norm())
            assert err <= 2e-2, err                       # other batch compositions regroup GEMM rows; same bar as the 1-worker test
        stats = eng.collective_rpc("serving_stats")
        assert len(stats) == 2 and all(st["steps"] > 0 and st["busy_frac"] is not None for st in stats)
    finally:
        eng.close()
