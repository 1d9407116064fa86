#!/usr/bin/env python3
"""bench.py — images/sec of the Qwen-Image DiT denoising path on N MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment launches itself: the parent spawns N ranks of this
file (env:// rendezvous on 127.0.0.1, one rank per GPU, same model as the reference's engine spawning one WorkerProc per
GPU, diffusion_engine.py:203-260) and relays rank 0's JSON line; under torch.distributed.run the ranks are used as given.

One "step" = every rank serves one STEP-BATCH of R (default 5) independent 1024x1024 requests end to end on the
hot path: 20 denoising steps with true-CFG (2 DiT forwards of the 60-layer Qwen-Image transformer per request per
step; all 2R items of the step-batch share ONE ragged DiT forward, per-request B=1 semantics), fused CFG+Euler
updates, VAE decode of every image, and (N > 1) the RCCL all-gather of the finished latents.  R=5 is chosen for
tile quantisation: 10 items = 163 row-tiles of 256, which fills the 256 CUs to 95.5-99.7 % in all four GEMM shapes and
the attention grid to 99.6 % (R=3: 92-98 % / 95.6 %; R=1: 77-93 %).  Inputs
(seeded noise, synthetic prompt embeddings T=64, random-init bf16 weights of the real architecture) are resident
in HBM before the timed region.  value = N*K*R images / max-over-ranks seconds.  Data-parallel by request
(weak scaling): no collective inside the denoise loop.

Also printed in the same JSON line:
  roofline     — dominant kernel = gemm_bf16_pp_kernel<GELU> (the ping-pong MFMA GEMM; MLP up-projection, 27 % of the DiT FLOPs, one shape
                 per launch so rocprofv3's per-kernel average is shape-pure): algorithmic 2*M*N*K flop per launch
                 / average launch duration measured here with HIP events on the launch stream.
  cpu_baseline — the fp32 CPU oracle (kind "port": /root/reference does not exist on the GPU box, so the shim-imported
                 reference cannot be timed there) on this box's host cores on a bounded sample of 4 full-width blocks,
                 extrapolated linearly in layers/forwards (rank 0, N=1 only).
  secondary    — (N=1 only) the no-CFG variant of the same workload (1 forward per step), BASELINE config 1
                 (256x256, 4 steps, batch 1) eager vs hipGraph replay, TeaCache rel_l1_thresh 0.2, and the SERVING path:
                 DiffusionEngine.submit / poll (a worker process running the continuous step batcher) fed 2R requests that
                 arrive staggered, against the static loop of the headline line.
  per_rank     — every rank's wall seconds of the timed region and, from HIP events on its stream, the seconds it spent in the
                 denoise loops, in the latent all-gather and in the VAE decodes (a sub-linear 1 -> N curve can be attributed).
  --sp P       — (opt-in, needs --gpus P) after the headline measurement the P ranks form ONE Ulysses group and time a
                 single 2048x2048 true-CFG request sequence-parallel: `secondary.sp_*`; the 1-GPU value
                 of the same request is `secondary.res2048_bf16_ms_per_denoise_step` of an N = 1 run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HEIGHT = WIDTH = 1024
STEPS_DENOISE = 20
T_TXT = 64
LAYERS = 60
TRUE_CFG = 4.0
PFLOP_PER_IMAGE = 2.777e15          # 40 x 69.310 TF (DiT) + 4.71 TF (VAE)
PEAK_BF16 = 2.5e15                  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def cpu_baseline(layers_sample: int = 4) -> dict:
    """Oracle DiT blocks at full width on the host cores; extrapolated to images/sec."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import qwen_image_oracle as O

    try:
        cores = len(os.sched_getaffinity(0))        # cores this process may actually use (cgroup / affinity aware)
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))                  # torch CPU GEMM stops scaling (and oversubscribes) beyond this
    torch.set_num_threads(cores)
    O.FUSED_SDPA = True     # attention through F.scaled_dot_product_attention, the op the reference's SDPAImpl calls (sdpa.py:55-63):
    #                         on the authoring host the port then runs within a few % of the shim-imported reference itself
    #                         (tools/time_reference_cpu.py)
    D, S_img = 3072, (HEIGHT // 16) * (WIDTH // 16)
    P = O.make_dit_params(layers_sample, seed=1234)
    g = torch.Generator().manual_seed(0)
    hidden, enc, temb = torch.randn(1, S_img, D, generator=g), torch.randn(1, T_TXT, D, generator=g), torch.randn(1, D, generator=g)
    vid, txt = O.rope_tables(1, HEIGHT // 16, WIDTH // 16, T_TXT)
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(layers_sample):
            enc, hidden = O.dit_block(P, i, hidden, enc, temb, vid, txt, 24)
        dt = (time.perf_counter() - t0) / layers_sample
    sec_per_image = dt * LAYERS * STEPS_DENOISE * 2          # 2 forwards per step (true-CFG), blocks dominate
    return {"value": 1.0 / sec_per_image, "unit": "images/sec", "cores": cores, "cores_available": os.cpu_count(), "kind": "port",
            "sample": f"{layers_sample} full-width DiT block(s) (S_img=4096, T=64, fp32) = {dt:.2f} s/block, "
                      f"extrapolated x{LAYERS} layers x{STEPS_DENOISE * 2} forwards"}


def measure_roofline(dev, R: int = 5) -> dict:
    """HIP-event time of the dominant kernel (MLP-up GEMM + GELU at the step-batch shape, production layouts) measured IN THE
    KERNEL MIX OF A DiT BLOCK: a launched-alone loop of one kernel settles at its own clock / power point (the part runs at its
    package power limit), up to 3 % off what the same launch takes between the other kernels of a layer (round-5 first profile:
    2247 us alone vs 2173 us inside the bench under rocprofv3).  So the timed launch sits in a replica of a block's launch
    sequence — QKV GEMM, joint attention, out-proj GEMM, [MLP-up: timed], MLP-down GEMM, all at the bench's shapes — and one
    event pair brackets the MLP-up launch of every iteration, on the stream it is launched on."""
    import math

    from vllm_omni_amd import ops

    BF = torch.bfloat16
    D, H, Mi, Mt = 3072, 24, 2 * R * 4096, 2 * R * T_TXT
    N, K = 4 * D, D
    g = torch.Generator(device=dev).manual_seed(7)

    def rn(rows, cols, s=1.0):
        return (torch.randn(rows, cols, device=dev, generator=g) * s).to(BF)

    blk = ops.w_to_k32_blocked
    # the timed launch: A, W and the GELU output K32-blocked, as in the layers
    xi, xt = blk(rn(Mi, K)), blk(rn(Mt, K))
    wi, wt = blk(rn(N, K, 0.02)), blk(rn(N, K, 0.02))
    b = torch.zeros(N, device=dev, dtype=BF)
    oi, ot = torch.empty(Mi, N, device=dev, dtype=BF), torch.empty(Mt, N, device=dev, dtype=BF)

    def mlp_up():
        ops.gemm([ops.GemmGroupArgs(xi, wi, b, oi, a_k32_blocked=True, out_k32_blocked=True),
                  ops.GemmGroupArgs(xt, wt, b, ot, a_k32_blocked=True, out_k32_blocked=True)], ops.EPI_BIAS_GELU_TANH,
                 w_k32_blocked=True)

    # the neighbours of that launch in a block (plain bias epilogues: they only set the mix the timed launch runs in)
    w_qkv, w_o, w_dn = blk(rn(3 * D, D, 0.02)), blk(rn(D, D, 0.02)), blk(rn(D, 4 * D, 0.02))
    b3, b1 = torch.zeros(3 * D, device=dev, dtype=BF), torch.zeros(D, device=dev, dtype=BF)
    qkv = torch.empty(Mi, 3 * D, device=dev, dtype=BF)
    o1, o2 = torch.empty(Mi, D, device=dev, dtype=BF), torch.empty(Mi, D, device=dev, dtype=BF)
    q, k, v = rn(Mi + Mt, D), rn(Mi + Mt, D), rn(Mi + Mt, D)
    S = 4096 + T_TXT
    cu = (torch.arange(2 * R + 1, dtype=torch.int32) * S).to(dev)
    att = torch.empty(Mi + Mt, D, device=dev, dtype=BF)

    def neighbours_before():
        ops.gemm([ops.GemmGroupArgs(xi, w_qkv, b3, qkv, a_k32_blocked=True)], ops.EPI_BIAS, w_k32_blocked=True)
        ops.flash_attn_varlen(q, k, v, cu, H, S, 1.0 / math.sqrt(128), out=att)
        ops.gemm([ops.GemmGroupArgs(xi, w_o, b1, o1, a_k32_blocked=True)], ops.EPI_BIAS, w_k32_blocked=True)

    def neighbours_after():
        ops.gemm([ops.GemmGroupArgs(oi, w_dn, b1, o2, a_k32_blocked=True)], ops.EPI_BIAS, w_k32_blocked=True)

    def block_mix(ev=None):
        neighbours_before()
        if ev is not None:
            ev[0].record(s)
        mlp_up()
        if ev is not None:
            ev[1].record(s)
        neighbours_after()

    s = torch.cuda.current_stream()
    for _ in range(3):
        block_mix()
    iters = 30
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for ev in evs:
        block_mix(ev)
    torch.cuda.synchronize()
    times = sorted(e0.elapsed_time(e1) * 1e-3 for e0, e1 in evs)
    sec = sum(times) / iters
    # the same launch back to back, alone (what rounds 1-4 reported as `avg_launch_us`)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        mlp_up()
    e1.record(s)
    torch.cuda.synchronize()
    sec_alone = e0.elapsed_time(e1) * 1e-3 / iters
    flops = 2.0 * (Mi + Mt) * N * K
    ach = flops / sec / 1e12
    # HBM/fabric bytes per launch of this very kernel + shape: bench.py cannot run rocprofv3 on itself, so it reports the
    # committed PMC measurement (tools/profile_round.sh -> tools/summarize_prof.py; 2 x FETCH_SIZE + WRITE_SIZE, separate passes)
    traffic, traffic_src = None, None
    kname = "gemm_bf16_pp_kernel<OMNI_EPI_BIAS_GELU_TANH>"
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_roofline_traffic_pmc.json")) as fh:
                j = json.load(fh)
            if kname.split("<")[0] in j.get("kernel", "") and f"M={Mi}+{Mt} " in j.get("kernel", ""):   # same kernel AND shape
                traffic, traffic_src = j["traffic_bytes"], f"profiles/{tag}_roofline_traffic_pmc.json (rocprofv3 --pmc, {j['kernel']})"
                break
        except (OSError, KeyError, ValueError):
            pass
    return {"bound": "mfma", "kernel": f"{kname} M={Mi}+{Mt} N=12288 K=3072 (W, A, out K32-blocked)",
            "achieved": ach, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": ach / (PEAK_BF16 / 1e12),
            "flop_per_launch": flops, "avg_launch_us": sec * 1e6, "median_launch_us": times[iters // 2] * 1e6,
            "launches_timed": iters,
            "timing": "HIP events around the launch inside a replica of a DiT block's kernel sequence (QKV GEMM, attention, "
                      "out-proj GEMM, [MLP-up], MLP-down GEMM at the bench shapes) on the launch stream",
            "avg_launch_us_back_to_back_alone": sec_alone * 1e6,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes": 2.0 * ((Mi + Mt) * K + 2 * N * K + (Mi + Mt) * N)}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: spawn the N ranks ourselves (one process per GPU, env:// rendezvous on
    127.0.0.1) and pass rank 0's stdout (the JSON line) through.  First contact with a real multi-GPU node must not be mute:
    every rank's stderr (and the stdout of ranks != 0) goes to `gpurun_out/bench_n{N}_rank{r}.log` (NCCL_DEBUG=WARN unless the
    caller set it), and as soon as ONE rank exits non-zero the others are terminated and the tail of the failing rank's log is
    printed — rank 0 does not sit in a collective until the driver's timeout."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    logdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
        path = os.path.join(logdir, f"bench_n{n}_rank{r}.log")
        fh = open(path, "w")
        logs.append((path, fh))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else fh, stderr=fh))
    rc, failed = 0, None
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and failed is None:
                failed, rc = r, abs(code) or 1
                for q in alive:                       # the rest would wait for the dead rank in a collective
                    procs[q].terminate()
        if alive:
            time.sleep(0.2)
    for _path, fh in logs:
        fh.close()
    if failed is not None:
        path = logs[failed][0]
        try:
            tail = open(path).read()[-3000:]
        except OSError:
            tail = ""
        print(f"bench.py: rank {failed} of {n} exited with status {rc}; the other ranks were terminated.  Its log ({path}) ends:\n{tail}",
              file=sys.stderr, flush=True)
    return rc


class _Watchdog:
    """First-contact guard for N > 1: if `what` has not finished within `seconds`, say which rank is stuck where and exit non-zero
    (self_launch then takes the other ranks down).  A failed RCCL bootstrap otherwise shows up as a silent hang."""

    def __init__(self, seconds: float, what: str):
        import threading

        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.what, self.seconds = what, seconds

    def _fire(self):
        print(f"bench.py watchdog: rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} still in '{self.what}' "
              f"after {self.seconds:.0f} s (MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}, device "
              f"{os.environ.get('LOCAL_RANK')}, HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}) — giving up",
              file=sys.stderr, flush=True)
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def _engine_pipeline(layers: int, R: int):
    """Pipeline factory of the serving-path line (runs inside the engine's worker process)."""
    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig, TransformerConfig
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline

    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = OmniDiffusionConfig(model="Qwen/Qwen-Image(random-init)", max_step_batch=R,
                              tf_model_config=TransformerConfig.from_dict({"num_layers": layers}))
    pipe = QwenImagePipeline(od_config=cfg, device=dev)
    pipe.transformer.init_random_(seed=1234)
    pipe.vae.init_random_(seed=4321)
    return pipe


def engine_line(R: int, layers: int, static_images_per_sec: float, n_workers: int = 1, devices=None, dist_backend=None) -> dict:
    """The SERVING path: DiffusionEngine -> one WorkerProc per GPU -> ContinuousStepBatcher (static per-composition buffers,
    device-side schedule vectors, bounded host run-ahead; reference spawn shape diffusion_engine.py:211-270, loop shape
    gpu_worker.py:226-290), fed 4R requests PER WORKER whose arrivals are staggered, so the running batches are re-composed
    while requests are mid-loop; the dispatcher hands each request to the least-loaded worker.  Images are decoded in the
    workers and returned through the result queue, as a server would.  Per worker: device-busy fraction from HIP events."""
    import functools

    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig, TransformerConfig
    from vllm_omni_amd.diffusion.diffusion_engine import DiffusionEngine
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    cfg = OmniDiffusionConfig(model="Qwen/Qwen-Image(random-init)", max_step_batch=R, num_gpus=n_workers, devices=devices,
                              dist_backend=dist_backend, tf_model_config=TransformerConfig.from_dict({"num_layers": layers}))
    eng = DiffusionEngine(cfg, pipeline_factory=functools.partial(_engine_pipeline, layers, R), post_process_func=None)
    try:
        g = torch.Generator().manual_seed(5)
        S = (HEIGHT // 16) * (WIDTH // 16)

        def req():
            return OmniDiffusionRequest(height=HEIGHT, width=WIDTH, num_inference_steps=STEPS_DENOISE, true_cfg_scale=TRUE_CFG,
                                        latents=torch.randn(1, S, 64, generator=g).to(torch.bfloat16),
                                        prompt_embeds=torch.randn(1, T_TXT, 3584, generator=g).to(torch.bfloat16),
                                        negative_prompt_embeds=torch.randn(1, T_TXT, 3584, generator=g).to(torch.bfloat16),
                                        output_type="pt")

        for o in eng.add_req_and_wait_for_response([req() for _ in range(R * n_workers)]):   # warm-up: one full batch per worker
            if o.error:
                raise RuntimeError(o.error)
        eng.collective_rpc("serving_stats", kwargs={"reset": True})
        n, gap = 4 * R * n_workers, 0.1 / n_workers         # four batches' worth per worker; the same arrival rate per worker
        t0 = time.perf_counter()
        ids = []
        for i in range(n):                                   # one request every `gap` seconds: the batches grow while they run
            ids.append(eng.submit(req()))
            time.sleep(gap)
        outs = [eng.poll(i) for i in ids]
        dt = time.perf_counter() - t0
        if any(o is None or o.error for o in outs):
            raise RuntimeError(str([o.error for o in outs if o is not None and o.error]))
        stats = eng.collective_rpc("serving_stats")
        workers = [{"rank": r, "steps": st["steps"], "mean_batch": st["sample_steps"] / max(1, st["steps"]),
                    "device_busy_s": st["busy_s"], "device_span_s": st["span_s"], "busy_frac": st["busy_frac"]}
                   for r, st in enumerate(stats)]
        return {"engine_images_per_sec": n / dt, "engine_vs_static_loop": (n / dt) / static_images_per_sec,
                "engine_workers": workers,
                "engine_note": f"DiffusionEngine.submit/poll, {n_workers} worker process(es), {n} requests of 1024^2 x 20 steps true-CFG "
                               f"arriving {gap:.3f} s apart (continuous step batching, max {R} per forward, host at most 2 steps "
                               "ahead of the device), images decoded in the workers and returned as CPU tensors; wall time from the "
                               "first submit to the last result; engine_workers[].busy_frac = device seconds inside denoise steps / "
                               "span from the first to the last step (HIP events)"}
    finally:
        eng.close()


def engine_line_guarded(R: int, layers: int, static_images_per_sec: float, n_workers: int, devices, dist_backend,
                        timeout_s: float) -> dict:
    """`engine_line` in a child interpreter of its own session, with a deadline.  The serving topology (one worker process per
    GPU, their own process group) is first-contact code on a multi-GPU node: whatever it does — a hung rendezvous, a crashed
    worker — must not take the headline line with it.  The child gets the torchrun / rank environment removed (it spawns its own
    workers on a fresh port), prints the engine dict as one JSON line, and is killed with its whole process group (the workers)
    when the deadline passes; its stderr goes to gpurun_out/bench_engine_n<N>.log."""
    import signal
    import subprocess

    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
            "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}
    spec = json.dumps({"R": R, "layers": layers, "static": static_images_per_sec, "n_workers": n_workers, "devices": devices,
                       "dist_backend": dist_backend})
    logdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(logdir, exist_ok=True)
    path = os.path.join(logdir, f"bench_engine_n{n_workers}.log")
    with open(path, "w") as fh:
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--engine-child", spec], env=env, stdout=subprocess.PIPE,
                             stderr=fh, start_new_session=True, text=True)
        try:
            out, _ = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            p.wait()
            return {"engine_error": f"no result within {timeout_s:.0f} s: the engine process group was killed ({path})"}
    for ln in reversed(out.splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                break
    return {"engine_error": f"engine child exited with status {p.returncode} and no result line ({path})"}


def secondary_lines(pipe, dev, R: int, layers: int) -> dict:
    """SURVEY.md §8d secondary measurements on one GPU (rank 0, N = 1): no-CFG line and BASELINE config 1 eager vs graph."""
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    out = {}
    g = torch.Generator().manual_seed(3)

    def reqs(n, hw, steps, cfg, T=T_TXT):
        S = (hw // 16) ** 2
        rs = []
        for _ in range(n):
            kw = dict(negative_prompt_embeds=torch.randn(1, T, 3584, generator=g).to(dev, torch.bfloat16)) if cfg else {}
            rs.append(OmniDiffusionRequest(height=hw, width=hw, num_inference_steps=steps, true_cfg_scale=TRUE_CFG,
                                           latents=torch.randn(1, S, 64, generator=g).to(dev, torch.bfloat16),
                                           prompt_embeds=torch.randn(1, T, 3584, generator=g).to(dev, torch.bfloat16),
                                           output_type="latent", **kw))
        return rs

    def timed(fn, n=1):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    # no-CFG: 2R requests per step-batch keeps the same 2R items per forward as the headline line
    nocfg = reqs(2 * R, HEIGHT, STEPS_DENOISE, cfg=False)
    old_cap = pipe.od_config.max_step_batch
    pipe.od_config.max_step_batch = 2 * R
    t = timed(lambda: [pipe.decode_latents(o.output, HEIGHT, WIDTH) for o in pipe.generate(nocfg, output_type="latent")])
    pipe.od_config.max_step_batch = old_cap
    out["no_cfg_images_per_sec"] = 2 * R / t
    out["no_cfg_note"] = f"1024^2, 20 steps, 1 forward/step, {2 * R} requests step-batched, incl. VAE decode"
    # latency of ONE 1024^2 request (true-CFG: a 2-item forward, 33 row tiles: the N = 3072 GEMMs fill 1.55 -> 2 rounds)
    single = reqs(1, HEIGHT, STEPS_DENOISE, cfg=True)
    t = timed(lambda: pipe.decode_latents(pipe.generate(single, output_type="latent")[0].output, HEIGHT, WIDTH))
    out["single_request_1024px_seconds_per_image"] = t
    # small step-batches (a lightly loaded worker): r requests of the headline workload step-batched, images/s and the fraction
    # of the bf16 MFMA roofline.  r = 2 / 4 take the GEMM tail split (12 .. 96 tail tiles), r = 1 .. 4 the attention short-block
    # split (DESIGN.md kernel table; profiles/r06_ab_splits_thin_tail_rule.log)
    out["r1_images_per_sec"] = 1.0 / t
    for r in (2, 3):
        few = reqs(r, HEIGHT, STEPS_DENOISE, cfg=True)
        tr_ = timed(lambda: [pipe.decode_latents(o.output, HEIGHT, WIDTH) for o in pipe.generate(few, output_type="latent")])
        out[f"r{r}_images_per_sec"] = r / tr_
    for r in (1, 2, 3):
        out[f"r{r}_mfma_roofline_frac"] = out[f"r{r}_images_per_sec"] * PFLOP_PER_IMAGE * (layers / LAYERS) / PEAK_BF16
    # BASELINE config 1 at real depth: 256x256, 4 steps, batch 1, true-CFG; eager launches vs hipGraph replay
    one = reqs(1, 256, 4, cfg=True)
    for name, flag in (("eager", False), ("hipgraph", True)):
        pipe.od_config.use_hip_graph = flag
        t = timed(lambda: pipe.decode_latents(pipe.generate(one, output_type="latent")[0].output, 256, 256), n=5)
        out[f"config1_256px_4step_{name}_ms_per_image"] = t * 1e3
    pipe.od_config.use_hip_graph = None
    # the same request when an earlier request of the same schedule (resolution, step count) left its modulation table behind
    # (od_config.cache_modulation_tables, the serving default): the 13.6 GB pass over the modulation weights is not repeated
    pipe.od_config.cache_modulation_tables = True
    t = timed(lambda: pipe.decode_latents(pipe.generate(one, output_type="latent")[0].output, 256, 256), n=5)
    pipe.od_config.cache_modulation_tables = False
    pipe.transformer._mod_tables.clear()
    out["config1_256px_4step_warm_schedule_ms_per_image"] = t * 1e3
    out["config1_note"] = (f"256x256, 4 steps, true-CFG (8 forwards of {layers} layers), batch 1, + VAE decode; eager / hipgraph: every "
                           "image pays its own modulation-table pass; warm_schedule: the table of this (resolution, steps) schedule is "
                           "reused from an earlier request (a function of weights and schedule only)")
    # BASELINE config 1 is weight-bandwidth bound on paper: its roofline is HBM.  Algorithmic bytes per image = the
    # DiT weights once per RAGGED forward (the CFG pair shares one weight stream: 4 forwards x 40.86 GB at 60 layers, of which the
    # 13.6 GB of modulation weights are read once per request by the table pass instead) — at 640 rows the layer is in fact as
    # MFMA-bound as HBM-bound: the fraction below is against the HBM peak only.
    gb = (4 * (40.86 - 13.6) + 13.6) * layers / LAYERS
    t1 = out["config1_256px_4step_eager_ms_per_image"] * 1e-3
    out["config1_hbm_roofline"] = {"bound": "hbm", "achieved": gb / t1, "peak": 8000.0, "unit": "GB/s", "frac": gb / t1 / 8000.0,
                                   "algorithmic_gb_per_image": gb}
    # the same small requests step-batched: one weight stream (41 GB per forward) serves 16 requests
    many = reqs(16, 256, 4, cfg=True)
    pipe.od_config.max_step_batch = 16
    t = timed(lambda: [pipe.decode_latents(o.output, 256, 256) for o in pipe.generate(many, output_type="latent")], n=2)
    pipe.od_config.max_step_batch = old_cap
    out["config1_256px_4step_stepbatched16_images_per_sec"] = 16 / t
    # BASELINE config 5 geometry in bf16: 2048x2048 (16384 image tokens per item: attention is 47 % of the FLOPs), one
    # request with true-CFG, THE CONFIG'S 50 STEPS run for real (round 4 timed 3 steps and extrapolated) + its VAE decode.
    # Warm-up = the same request with 2 steps (same launch shapes), then ONE timed 50-step generation.
    def run2048():
        pipe.generate(reqs(1, 2048, 2, cfg=True), output_type="latent")
        big50 = reqs(1, 2048, 50, cfg=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = pipe.generate(big50, output_type="latent")[0].output
        torch.cuda.synchronize()
        t_loop = time.perf_counter() - t0
        img = pipe.decode_latents(o, 2048, 2048)                          # warm-up of the 2048^2 decode shapes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        img = pipe.decode_latents(o, 2048, 2048)
        torch.cuda.synchronize()
        return t_loop, time.perf_counter() - t0, bool(torch.isfinite(img.float()).all())

    t50, tv, ok = run2048()
    flop_step = 2 * 423.01e12 * layers / LAYERS     # 423.01 TFLOP per forward at 2048^2 (60 layers), two forwards per step
    out["res2048_bf16_ms_per_denoise_step"] = t50 / 50 * 1e3
    out["res2048_bf16_vae_decode_ms"] = tv * 1e3
    out["res2048_bf16_50step_seconds_per_image"] = t50 + tv
    out["res2048_bf16_50step_images_per_sec"] = 1.0 / (t50 + tv)
    out["res2048_bf16_dit_mfma_roofline_frac"] = flop_step / (t50 / 50) / 2.5e15
    out["res2048_finite_outputs"] = ok
    out["res2048_note"] = ("2048x2048, 50 steps, true-CFG, batch 1: ONE full 50-step generation timed (the modulation table of the "
                           "request included) + one measured VAE decode; *_50step_images_per_sec = 1 / (loop + decode)")
    # BASELINE config 5 proper: the same request with the block GEMMs in fp8 (e4m3 weights per output channel, activations
    # quantised per token in front of each GEMM, scaled MFMA at twice the bf16 rate; attention, norms, residuals stay bf16).
    # Every fp8 throughput figure carries its accuracy: the final latent of ONE headline request (1024^2, 20 steps, true-CFG)
    # in that mode against the bf16 path (the fp32 oracle is test infrastructure; tests/test_gpu_fp8.py bounds the same drift
    # against it).  Two recipes: all four GEMM classes in fp8 (maximum throughput) and the ACCURATE one (attention-side
    # projections only; within 2x of bf16's own drift from fp32 - DESIGN.md 7 item 23).
    tr = pipe.transformer
    probe = reqs(1, HEIGHT, STEPS_DENOISE, cfg=True)
    lat16 = pipe.generate(probe, output_type="latent")[0].output.float()

    def drift():
        got = pipe.generate(probe, output_type="latent")[0].output.float()
        return float((got - lat16).norm() / lat16.norm())

    try:
        head = reqs(R, HEIGHT, STEPS_DENOISE, cfg=True)
        for tag, classes in (("fp8", tr.FP8_CLASSES), ("fp8_accurate", tr.FP8_RECIPE_ACCURATE)):
            tr.enable_fp8(classes)
            t = timed(lambda: [pipe.decode_latents(o.output, HEIGHT, WIDTH) for o in pipe.generate(head, output_type="latent")])
            out[f"{tag}_1024px_images_per_sec"] = R / t
            out[f"{tag}_final_latent_rel_l2_vs_bf16"] = drift()
        tr.enable_fp8()
        t50f, tvf, okf = run2048()
        out["res2048_fp8_ms_per_denoise_step"] = t50f / 50 * 1e3
        out["res2048_fp8_50step_seconds_per_image"] = t50f + tvf
        out["res2048_fp8_50step_images_per_sec"] = 1.0 / (t50f + tvf)
        out["res2048_fp8_finite_outputs"] = okf
        # roofline of the mixed-precision step: GEMM flops against the fp8 peak (5 PF), attention flops against the bf16 peak
        gemm_flop = 2 * (423.01e12 * layers / LAYERS - 4.0 * 24 * (16384 + T_TXT) ** 2 * 128 * layers)
        attn_flop = 2 * 4.0 * 24 * (16384 + T_TXT) ** 2 * 128 * layers
        out["res2048_fp8_roofline_frac"] = (gemm_flop / 5.0e15 + attn_flop / 2.5e15) / (t50f / 50)
        out["res2048_fp8_note"] = ("transformer.enable_fp8(): OCP e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4), bf16 attention / "
                                   "norms / residual streams; the reference has no fp8 path.  fp8_* = all four block-GEMM classes, "
                                   "fp8_accurate_* = qkv + out-proj only; *_final_latent_rel_l2_vs_bf16 = final latent of one headline "
                                   "request (20 steps, true-CFG, random-init weights) in that mode vs the bf16 path - bf16 itself is "
                                   "2.2e-2 from the fp32 oracle on that loop (tests/test_gpu_bench_shape_parity.py); the drift vs fp32 "
                                   "is bounded in tests/test_gpu_fp8.py.  roofline_frac = (GEMM flop / 5 PF + attention flop / 2.5 PF) / "
                                   "measured step time; *_1024px_images_per_sec = the headline workload (R requests, 20 steps, + VAE)")
    finally:
        tr.enable_fp8(False)
    # TeaCache (device-side decisions, no host sync) on the headline workload
    try:
        from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig

        pipe.transformer.teacache = TeaCacheConfig(rel_l1_thresh=0.2)
        head = reqs(R, HEIGHT, STEPS_DENOISE, cfg=True)
        t = timed(lambda: [pipe.decode_latents(o.output, HEIGHT, WIDTH) for o in pipe.generate(head, output_type="latent")])
        skipped = pipe.last_teacache_state.skipped_forwards()
        out["teacache_thresh0.2_images_per_sec"] = R / t
        out["teacache_skipped_forwards_per_item"] = skipped
        out["teacache_note"] = (f"1024^2, 20 steps, true-CFG, {R} requests step-batched, rel_l1_thresh 0.2, Qwen-Image "
                                "coefficients; random-init weights, so the skip pattern is NOT that of a trained model")
    finally:
        pipe.transformer.teacache = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=LAYERS, help=argparse.SUPPRESS)      # dev only; default = real model
    # R = 5: 2R = 10 items x 4160 rows gives 163 m-tiles: the N = 3072 GEMMs fill 7.64 -> 8 rounds of 256 CUs (95.5 %; R = 3:
    # 4.59 -> 5 rounds, 91.9 %), attention 15.94 -> 16 rounds.  Same box: R = 3 0.4232, R = 5 0.4325, R = 7 0.4323 images/s.
    ap.add_argument("--requests", type=int, default=5, help="requests step-batched per rank per step (R)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-engine", action="store_true", help="skip the serving-path (DiffusionEngine) secondary line")
    ap.add_argument("--sp", type=int, default=0, help="P = --gpus: also time one 2048^2 request Ulysses-parallel over all ranks")
    # dev / test only: run the N-rank control flow (self-launch, rendezvous, per-rank stats, gather, JSON) on ONE device —
    # every rank on device 0 over a gloo group (RCCL refuses duplicate GPUs).  tests/test_gpu_multirank_bench.py
    ap.add_argument("--init-timeout", type=float, default=300.0,
                    help="seconds the rendezvous + first RCCL collective may take before the rank reports itself stuck and exits")
    ap.add_argument("--engine-timeout", type=float, default=900.0,
                    help="seconds the serving-path line (a child process group) may take before it is killed and reported as an error")
    ap.add_argument("--engine-child", default=None, help=argparse.SUPPRESS)       # internal: engine_line_guarded's child
    ap.add_argument("--dist-backend", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--share-device", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    from vllm_omni_amd.diffusion.data import OmniDiffusionConfig, TransformerConfig
    from vllm_omni_amd.diffusion.distributed import data_parallel as dp
    from vllm_omni_amd.diffusion.models.qwen_image.pipeline_qwen_image import QwenImagePipeline
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    if args.engine_child:
        spec = json.loads(args.engine_child)
        try:
            res = engine_line(spec["R"], spec["layers"], spec["static"], n_workers=spec["n_workers"], devices=spec["devices"],
                              dist_backend=spec["dist_backend"])
        except Exception as e:  # noqa: BLE001
            res = {"engine_error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the product kernels")
    local_dev = None
    if args.share_device:
        local_dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    with _Watchdog(args.init_timeout, "process-group rendezvous + first RCCL all-reduce"):
        rank, world, local = dp.init_distributed(backend=args.dist_backend, local_device=local_dev)
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:                                  # RCCL builds its communicator on the first collective: do it HERE, watched
            probe = torch.ones(1, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(probe)
            if int(probe.item()) != world:
                raise SystemExit(f"first all-reduce returned {probe.item()} on a world of {world}")
    from vllm_omni_amd.diffusion.distributed.numa import pin_to_gpu_numa

    numa = pin_to_gpu_numa(local)                  # this rank's host threads next to its GPU (distributed/numa.py)
    # idle ranks wait on the CPU (a gloo group): an RCCL barrier spins a kernel on the GPU the serving-path workers then use
    cpu_group = torch.distributed.new_group(backend="gloo") if world > 1 else None

    R = args.requests
    # cache_modulation_tables=False: every timed generation below pays its own pass over the modulation weights (the serving
    # default keeps a schedule's table for later requests; secondary.config1_*_warm_schedule_* shows what that is worth)
    cfg = OmniDiffusionConfig(model="Qwen/Qwen-Image(random-init)", max_step_batch=R, cache_modulation_tables=False,
                              tf_model_config=TransformerConfig.from_dict({"num_layers": args.layers}))
    pipe = QwenImagePipeline(od_config=cfg, device=dev)
    pipe.transformer.init_random_(seed=1234)
    pipe.vae.init_random_(seed=4321)

    S_img = (HEIGHT // 16) * (WIDTH // 16)
    g = torch.Generator().manual_seed(1)
    pos = [torch.randn(1, T_TXT, 3584, generator=g).to(dev, torch.bfloat16) for _ in range(R)]   # one prompt per request
    neg = [torch.randn(1, T_TXT, 3584, generator=g).to(dev, torch.bfloat16) for _ in range(R)]

    # HIP events on this rank's stream around the three parts of a step (no host synchronisation inside the timed region)
    marks: list[list[torch.cuda.Event]] = []

    def one_step(seed: int):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        reqs = []
        for r in range(R):
            lat = torch.randn(1, S_img, 64, generator=torch.Generator().manual_seed(seed * 97 + r)).to(dev, torch.bfloat16)
            reqs.append(OmniDiffusionRequest(height=HEIGHT, width=WIDTH, num_inference_steps=STEPS_DENOISE,
                                             true_cfg_scale=TRUE_CFG, latents=lat, prompt_embeds=pos[r],
                                             negative_prompt_embeds=neg[r], output_type="latent"))
        outs = pipe.generate(reqs, output_type="latent")                      # one step-batched denoise loop
        lat = torch.cat([o.output for o in outs]).contiguous()                # [R, S_img, 64]
        ev[1].record()
        gathered = dp.gather_latents(lat, [R] * world)                        # RCCL all-gather of finished latents
        ev[2].record()
        imgs = [pipe.decode_latents(lat, HEIGHT, WIDTH)]                               # each rank decodes its own R images, one VAE call
        ev[3].record()
        marks.append(ev)
        return gathered, imgs[-1], lat

    for i in range(args.warmup):
        one_step(100 + i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        gathered, img, sent_lat = one_step(42 + rank * 1000 + i)
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0                      # this rank's own seconds (before it waits for the others)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    parts = [sum(ev[k].elapsed_time(ev[k + 1]) for ev in marks) * 1e-3 for k in range(3)]   # denoise, gather, decode
    stats = torch.tensor([elapsed, mine] + parts + [float(numa["node"]) if numa["node"] is not None else -1.0,
                                                    1.0 if numa["pinned"] else 0.0, float(numa["cpus"]), float(local)],
                         dtype=torch.float64)
    if world > 1:
        allstats = [torch.zeros_like(stats) for _ in range(world)]
        torch.distributed.all_gather(allstats, stats, group=cpu_group)
        elapsed = max(float(t[0]) for t in allstats)
    else:
        allstats = [stats]
    per_rank = [{"rank": r, "seconds": float(t[1]), "denoise_s": float(t[2]), "gather_s": float(t[3]), "vae_decode_s": float(t[4]),
                 "numa_node": int(t[5]), "numa_pinned": bool(t[6]), "host_cpus": int(t[7]), "device": int(t[8])}
                for r, t in enumerate(allstats)]
    ok = bool(torch.isfinite(img.float()).all()) and bool(torch.isfinite(gathered.float()).all())

    # N > 1: make the record prove that the collective carried N ranks' data.  Every rank digests (a) the latents it contributed
    # in the last step and (b) each rank-sized slice of what the all-gather returned to it; the digests travel over the CPU (gloo)
    # group, and rank 0 checks that every rank received, in slot r, exactly what rank r sent (different seeds per rank: a gather
    # that returned local data N times, or skipped a rank, cannot pass).
    collective = None
    if world > 1:
        def digest(t):                                     # order-sensitive 62-bit digest of the bf16 bits
            v = t.contiguous().view(torch.int16).to(torch.int64).flatten() & 0xFFFF
            w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521) + 1
            return int(((v * w).sum() % ((1 << 61) - 1)).item())

        sent = digest(sent_lat)                            # what THIS rank handed to the collective in the last step
        got = [digest(gathered[r * R:(r + 1) * R]) for r in range(world)]
        mine_t = torch.tensor([sent] + got, dtype=torch.int64)
        alld = [torch.zeros_like(mine_t) for _ in range(world)]
        torch.distributed.all_gather(alld, mine_t, group=cpu_group)
        sent_by = [int(alld[r][0]) for r in range(world)]
        match = all(int(alld[q][1 + r]) == sent_by[r] for q in range(world) for r in range(world))
        distinct = len(set(sent_by)) == world
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            ver = None
        collective = {"backend": torch.distributed.get_backend(), "world": world, "rccl_version": ver,
                      "nccl_debug": os.environ.get("NCCL_DEBUG"), "rank_logs": f"gpurun_out/bench_n{world}_rank<r>.log",
                      "op": f"all_gather_into_tensor of [{R}, {S_img}, 64] bf16 per rank and step",
                      "gathered_checksum_matches_all_ranks": bool(match), "per_rank_payloads_distinct": bool(distinct),
                      "devices": sorted({int(t[8]) for t in allstats}) if allstats[0].numel() > 8 else None}

    sp_line = None
    if args.sp:
        # Opt-in: ONE request, sequence-parallel over all ranks (every rank runs the same request in lockstep; the two CFG
        # branches' all-to-alls hide behind each other's GEMMs, distributed/sp_driver.py).  3 denoise steps are timed.
        if args.sp != world:
            raise SystemExit(f"--sp {args.sp} needs --gpus {args.sp}")
        gsp = torch.Generator().manual_seed(11)
        big = OmniDiffusionRequest(height=2048, width=2048, num_inference_steps=3, true_cfg_scale=TRUE_CFG, output_type="latent",
                                   latents=torch.randn(1, 16384, 64, generator=gsp).to(dev, torch.bfloat16),
                                   prompt_embeds=torch.randn(1, T_TXT, 3584, generator=gsp).to(dev, torch.bfloat16),
                                   negative_prompt_embeds=torch.randn(1, T_TXT, 3584, generator=gsp).to(dev, torch.bfloat16))
        pipe.sp_group, pipe.sp_degree, pipe._force_sp_path = None, world, True
        try:
            pipe.generate([big], output_type="latent")
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            t1 = time.perf_counter()
            out_sp = pipe.generate([big], output_type="latent")[0].output
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            dt_sp = time.perf_counter() - t1
            sp_line = {"sp_degree": world, "sp_res2048_ms_per_denoise_step": dt_sp / 3 * 1e3,
                       "sp_finite": bool(torch.isfinite(out_sp.float()).all()),
                       "sp_note": "ONE 2048x2048 true-CFG request, Ulysses over all ranks (image rows sharded, heads sharded in "
                                  "attention, 2 all-to-alls per block, CFG branches software-pipelined); compare with "
                                  "secondary.res2048_bf16_ms_per_denoise_step of an N = 1 run (reference's published SP speed-ups "
                                  "at 2048^2: 1.73x / 2.84x / 3.65x on 2 / 4 / 8 GPUs, BASELINE.md)"}
        finally:
            pipe.sp_group, pipe.sp_degree, pipe._force_sp_path = None, 1, False

    if rank == 0:
        value = world * args.steps * R / elapsed
        line = {
            "metric": "images/sec (whole node) @1024^2, 20-step Qwen-Image DiT", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Qwen-Image DiT 1024x1024, 20 steps, true-CFG 4.0 (2 forwards/step), bf16, "
                                   f"{args.layers} layers, T=64 synthetic prompt embeds, + VAE decode; DP={world}, "
                                   f"{R} requests step-batched per rank",
                       "global_batch": world * R, "parallelism": f"dp{world}", "images_per_step": world * R},
            "finite_outputs": ok, "per_rank": per_rank,
            **({"collective": collective} if collective is not None else {}),
            "dit_mfma_roofline_frac": (value / world) * PFLOP_PER_IMAGE * (args.layers / LAYERS) / PEAK_BF16,
            "roofline": measure_roofline(dev, R),
        }
        if world == 1 and not args.no_secondary:
            try:
                line["secondary"] = secondary_lines(pipe, dev, R, args.layers)
            except Exception as e:  # noqa: BLE001 - the headline line must survive a failing secondary measurement
                line["secondary"] = {"error": f"{type(e).__name__}: {e}"}
        if sp_line is not None:
            line.setdefault("secondary", {}).update(sp_line)
    # the serving topology (N = 1 and N > 1): rank 0 drives a DiffusionEngine with one worker process per GPU; the ranks of the
    # static measurement release their weights first and (N > 1) wait on the CPU until rank 0 is through
    run_engine = not args.no_engine and not args.sp
    if run_engine:
        del pipe                                             # every worker process builds its own 41 GB of weights
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        if world > 1:
            torch.distributed.barrier(group=cpu_group)
    if rank == 0:
        if run_engine:
            devices = [r % torch.cuda.device_count() for r in range(world)] if args.share_device else None
            line.setdefault("secondary", {}).update(engine_line_guarded(R, args.layers, value, world, devices, args.dist_backend,
                                                                        args.engine_timeout))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier(group=cpu_group)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
