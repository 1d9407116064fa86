"""CPU (gloo, 2 and 4 ranks): the Ulysses resharding collectives and the parallel-attention strategy.

Mirrors the reference's own multi-GPU unit tests on CPU: tests/diffusion/distributed/test_comm.py (all-to-all twice ==
identity) and tests/diffusion/attention/test_ulysses_sequence_parallel.py (sequence-parallel attention == single-rank
attention, with a replicated joint text prefix, including a text length that is not divisible by the group size)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q_out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), DIFFUSION_ATTENTION_BACKEND="TORCH_SDPA",
                      HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")   # a host-only worker also on a GPU box: the selector refuses
    #                                                                        TORCH_SDPA wherever a GPU is visible
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from vllm_omni_amd.diffusion.attention.backends.abstract import AttentionMetadata
    from vllm_omni_amd.diffusion.attention.layer import Attention
    from vllm_omni_amd.diffusion.data import DiffusionParallelConfig, OmniDiffusionConfig, set_current_omni_diffusion_config
    from vllm_omni_amd.diffusion.distributed.comm import SeqAllToAll4D, SeqAllToAll5D, all_to_all_4D

    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    try:
        bs, s_loc, H, d = 2, 8, 8, 32                    # the reference test's shape: bs 2, seq/rank 8, 8 heads x 32
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(bs, s_loc, H, d, generator=g)
        y = SeqAllToAll4D.apply(None, x, 2, 1)
        assert y.shape == (bs, s_loc * world, H // world, d)
        # every rank's shard r of the gathered sequence is rank r's tensor restricted to my heads
        ref = [torch.randn(bs, s_loc, H, d, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        hp = H // world
        assert torch.equal(y, torch.cat([t[:, :, rank * hp:(rank + 1) * hp] for t in ref], dim=1))
        assert torch.equal(SeqAllToAll4D.apply(None, y, 1, 2), x)                      # twice == identity
        x5 = torch.randn(bs, s_loc, 3, H, d, generator=g)
        y5 = SeqAllToAll5D.apply(None, x5, 3, 1)
        assert y5.shape == (bs, s_loc * world, 3, H // world, d)
        for i in range(3):                                                              # fused == three separate exchanges
            assert torch.equal(y5[:, :, i], all_to_all_4D(x5[:, :, i].contiguous(), 2, 1))
        assert torch.equal(SeqAllToAll5D.apply(None, y5, 1, 3), x5)

        # ---- Ulysses attention == single-rank attention (joint text prefix replicated on all ranks; T = 13 not divisible)
        cfg = OmniDiffusionConfig(parallel_config=DiffusionParallelConfig(ulysses_degree=world))
        with set_current_omni_diffusion_config(cfg):
            sp_attn = Attention(num_heads=H, head_size=d, causal=False, softmax_scale=d ** -0.5)
        assert sp_attn.parallel_strategy.name == "ulysses"
        gg = torch.Generator().manual_seed(7)                                           # same on every rank
        S, T = s_loc * world, 13
        q, k, v = (torch.randn(bs, S, H, d, generator=gg) for _ in range(3))
        tq, tk, tv = (torch.randn(bs, T, H, d, generator=gg) for _ in range(3))
        sl = slice(rank * s_loc, (rank + 1) * s_loc)
        for strategy in ("front", "rear"):
            md = AttentionMetadata(joint_query=tq, joint_key=tk, joint_value=tv, joint_strategy=strategy)
            out = sp_attn(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(), md)
            cat = (lambda j, x_: torch.cat([j, x_], 1)) if strategy == "front" else (lambda j, x_: torch.cat([x_, j], 1))
            full = torch.nn.functional.scaled_dot_product_attention(
                cat(tq, q).permute(0, 2, 1, 3), cat(tk, k).permute(0, 2, 1, 3), cat(tv, v).permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
            # what a rank gets back: its joint rows = [text ; its image shard] ("front") or [its shard ; text] ("rear")
            mine = torch.cat([full[:, :T], full[:, T:][:, sl]], 1) if strategy == "front" else \
                torch.cat([full[:, :S][:, sl], full[:, S:]], 1)
            assert out.shape == mine.shape, (out.shape, mine.shape)
            assert torch.allclose(out, mine, atol=1e-5, rtol=1e-5), float((out - mine).abs().max())
        q_out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        q_out.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ulysses_collectives_and_strategy_on_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(v == "ok" for v in res.values()), res


# ---------------------------------------------------------------------------------------------------------------------
# sp_driver: generators that yield their collectives, software-pipelined over a process group (round 3)
def _toy_forward_gen(rank, P, x, scale):
    """Three segments with two exchanges: out = all_gather(all_to_all(x * scale) + rank) — every value is checkable."""
    send = (x * scale).reshape(P, -1).contiguous()
    got = yield ("all_to_all", send)                       # got[j] = what rank j sent to me
    y = got + float(rank)
    full = yield ("all_gather", y.reshape(-1).contiguous())
    return full.reshape(-1)


def _driver_rank(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from vllm_omni_amd.diffusion.distributed.sp_driver import drive

    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    base = torch.arange(world * 3, dtype=torch.float32) + 100.0 * rank
    gens = [_toy_forward_gen(rank, world, base, s) for s in (1.0, 2.0, -1.0)]      # three pipelined "forwards"
    outs = drive(gens, None)
    solo = [drive([_toy_forward_gen(rank, world, base, s)], None)[0] for s in (1.0, 2.0, -1.0)]
    q.put((rank, [o.tolist() for o in outs], [o.tolist() for o in solo]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sp_driver_pipelines_generators_over_gloo(world):
    from test_host_logic import _free_port

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, outs, solo = q.get(timeout=120)
        res[rank] = (outs, solo)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for s_i, s in enumerate((1.0, 2.0, -1.0)):
        # expected: rank r receives chunk r of every rank j's (base_j * s), adds r; the gather concatenates ranks
        want = []
        for r in range(world):
            for j in range(world):
                bj = torch.arange(world * 3, dtype=torch.float32) + 100.0 * j
                want += (bj * s).reshape(world, -1)[r].add(float(r)).tolist()
        for r in range(world):
            assert res[r][0][s_i] == want, (world, s, r)
            assert res[r][1][s_i] == want                  # pipelined == one at a time


def _sp_worker_rank(rank, world, port, q, degree):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from vllm_omni_amd.diffusion.data import DiffusionOutput, DiffusionParallelConfig, OmniDiffusionConfig
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest
    from vllm_omni_amd.diffusion.worker.gpu_worker import GPUWorker

    class FakeSPPipeline:
        """'Denoised latent' = seed + 1000 * (sum of the SP group's ranks, exchanged over the group the worker handed in): a
        wrong group, a rank that skips its share or a gather from the wrong rank all show."""
        device = torch.device("cpu")
        sp_group, sp_degree = None, 1
        ran = []

        def _req_params(self, r):
            return (r.height, r.width, r.num_inference_steps, 4.0, False)

        def generate(self, reqs, output_type="latent"):
            t = torch.tensor([float(rank)])
            dist.all_reduce(t, group=self.sp_group)
            self.ran += [r.seed for r in reqs]
            return [DiffusionOutput(output=torch.full((1, 16, 64), float(r.seed) + 1000.0 * float(t), dtype=torch.float32).bfloat16())
                    for r in reqs]

        def decode_latents(self, lat, h, w):
            return lat

    cfg = OmniDiffusionConfig(dist_timeout=60, parallel_config=DiffusionParallelConfig(ulysses_degree=degree, data_parallel_size=world // degree))
    pipe = FakeSPPipeline()
    w = GPUWorker(rank, rank, cfg, pipeline=pipe)
    w.init_device_and_model()
    assert pipe.sp_degree == degree and (w.dp_rank, w.dp_world) == (rank // degree, world // degree)
    reqs = [OmniDiffusionRequest(height=64, width=64, num_inference_steps=s, seed=i, prompt_embeds=torch.zeros(1, 1, 8))
            for i, s in enumerate([4, 20, 4, 4])]
    out = w.execute_model(reqs, decode=False)
    q.put((rank, out.error, None if out.output is None else out.output[:, 0, 0].float().tolist(), sorted(pipe.ran)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,degree", [(2, 2), (4, 2)])
def test_worker_shards_requests_over_sp_groups_on_gloo(world, degree):
    """world = data-parallel groups x ulysses_degree: every rank of a group runs the group's share in lockstep (its pipeline's
    collectives run over THAT group), only the group's first rank contributes rows to the latent gather."""
    from test_host_logic import _free_port

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sp_worker_rank, args=(r, world, port, q, degree)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, err, vals, ran = q.get(timeout=180)
        res[rank] = (err, vals, ran)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res[r][0] is None for r in range(world)), res
    for g in range(world // degree):
        shares = {tuple(res[r][2]) for r in range(g * degree, (g + 1) * degree)}
        assert len(shares) == 1                              # the ranks of a group ran the same requests
    assert sorted(s for g in range(world // degree) for s in res[g * degree][2]) == [0, 1, 2, 3]
    got = res[0][1]
    for i, v in enumerate(got):                              # value = seed + 1000 * sum(ranks of the group that ran it)
        g = next(g for g in range(world // degree) if i in res[g * degree][2])
        assert v == pytest.approx(i + 1000.0 * sum(range(g * degree, (g + 1) * degree)), rel=1e-2)


# ---------------------------------------------------------------------------------------------------------------------
# TeaCache under sequence parallelism (round 5): per-rank slices, ONE all-reduced pair of sums, the single-device decision
def _toy_modulated_inputs(steps=12, S=32, D=16):
    g = torch.Generator().manual_seed(77)
    x = torch.randn(S, D, generator=g)
    seq = []
    for t in range(steps):                                   # a drifting "modulated input": small and large steps mixed
        x = x + (0.02 if t % 3 else 0.25) * torch.randn(S, D, generator=g)
        seq.append(x.bfloat16())
    return seq


def _toy_teacache_forward(rank, P, mod_full, st):
    """The TeaCache part of `_sp_forward_gen` on this rank's row slice of a given modulated input."""
    S, D = mod_full.shape
    mod = mod_full[rank * (S // P):(rank + 1) * (S // P)]
    if st.cnt > 0 and st.prev_mod is not None:
        part = torch.stack([(mod - st.prev_mod).abs().float().sum(), st.prev_mod.abs().float().sum()])
        total = yield ("all_reduce", part)
        compute = st.decide(total, S * D)
    else:
        st.first()
        compute = True
    st.prev_mod = mod
    flags = yield ("all_gather", torch.tensor([1.0 if compute else 0.0]))
    return flags.reshape(-1)


def _reference_rule_decisions(seq, cfg):
    """The reference's host hook on the FULL tensors (vllm_omni/diffusion/cache/teacache/hook.py:170-217 via our TeaCacheHook)."""
    from vllm_omni_amd.diffusion.cache.teacache.hook import TeaCacheHook, TeaCacheState

    hook, st, out = TeaCacheHook(cfg), TeaCacheState(), []
    for m in seq:
        out.append(hook._should_compute_full_transformer(st, m))
        st.previous_modulated_input = m
        st.cnt += 1
    return out


def _tc_rank(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.cache.teacache.sp_state import TeaCacheSPState
    from vllm_omni_amd.diffusion.distributed.sp_driver import drive

    dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
    cfg = TeaCacheConfig(rel_l1_thresh=0.3, transformer_type="QwenImageTransformer2DModel")
    seq = _toy_modulated_inputs()
    # two items ("CFG branches") pipelined per step, each with its own state
    sts = [TeaCacheSPState(cfg), TeaCacheSPState(cfg)]
    seen = []
    for m in seq:
        outs = drive([_toy_teacache_forward(rank, world, m, sts[0]), _toy_teacache_forward(rank, world, (m * 1.5).bfloat16(), sts[1])], None)
        seen.append([o.tolist() for o in outs])
    q.put((rank, seen, [s.decisions for s in sts], [s.skipped for s in sts]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_teacache_under_sequence_parallelism_takes_the_single_device_decisions_on_gloo(world):
    from test_host_logic import _free_port

    from vllm_omni_amd.diffusion.cache.teacache.config import TeaCacheConfig
    from vllm_omni_amd.diffusion.cache.teacache.sp_state import TeaCacheSPState
    from vllm_omni_amd.diffusion.distributed.sp_driver import drive_in_process

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tc_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, seen, dec, skipped = q.get(timeout=180)
        res[rank] = (seen, dec, skipped)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = TeaCacheConfig(rel_l1_thresh=0.3, transformer_type="QwenImageTransformer2DModel")
    seq = _toy_modulated_inputs()
    want = [_reference_rule_decisions(seq, cfg), _reference_rule_decisions([(m * 1.5).bfloat16() for m in seq], cfg)]
    assert any(not d for d in want[0][1:]) and any(d for d in want[0][1:])          # the toy sequence both skips and computes
    for r in range(world):
        assert res[r][1] == want, (r, res[r][1], want)                              # every rank == the single-device rule
        assert res[r][2] == [sum(1 for d in w if not d) for w in want]
        for step, per_item in enumerate(res[r][0]):                                 # and every rank SAW every rank agree
            for i, flags in enumerate(per_item):
                assert flags == [1.0 if want[i][step] else 0.0] * world
    # the in-process exchange used by the one-device GPU tests gives the same decisions
    sts = [[TeaCacheSPState(cfg)] for _ in range(world)]
    for m in seq:
        drive_in_process([[_toy_teacache_forward(r, world, m, sts[r][0])] for r in range(world)])
    assert all(sts[r][0].decisions == want[0] for r in range(world))
