"""GPU: the Ulysses sequence-parallel DiT forward on ONE device.

GPU boxes here have a single MI355X, so the multi-rank exchange cannot run over RCCL; the SP forward is a generator that
yields its collectives, and this test drives P generators in lock step, performing the all-to-all / all-gather IN PROCESS
(recv[r][j] = send[j][r]) — every kernel, every reshard and every index permutation of the real path runs; only the wire is
replaced.  Reference test being mirrored: tests/diffusion/attention/test_ulysses_sequence_parallel.py and
tests/e2e/offline_inference/test_sequence_parallel.py (SP == non-SP)."""
import pytest
import torch

import qwen_image_oracle as O
from _util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda:0"


def _emulate(model, P, lat, txt, sig, grid):
    gens = [model._sp_forward_gen(r, P, lat, txt, sig, grid) for r in range(P)]
    msgs = [next(g) for g in gens]
    n_a2a = 0
    while True:
        kinds = {m[0] for m in msgs}
        assert len(kinds) == 1, kinds
        kind = kinds.pop()
        sends = [m[1] for m in msgs]
        if kind == "all_to_all":
            n_a2a += 1
            outs = [torch.stack([sends[j][r] for j in range(P)]) for r in range(P)]
        else:
            outs = [torch.stack(sends) for _ in range(P)]
        nxt, done = [], []
        for g, o in zip(gens, outs):
            try:
                nxt.append(g.send(o))
            except StopIteration as e:
                done.append(e.value)
        if done:
            assert len(done) == P
            return done, n_a2a
        msgs = nxt


@pytest.mark.parametrize("P", [1, 2, 4])
def test_ulysses_forward_equals_single_device_forward(P):
    from vllm_omni_amd.diffusion.models.qwen_image.qwen_image_transformer import QwenImageTransformer2DModel

    heads, joint, layers, grid, T = 4, 128, 3, (1, 16, 8), 13          # T = 13: not divisible by P (replicated text path)
    Pm = O.make_dit_params(layers, seed=1234, bias_std=0.02, norm_jitter=0.1, num_heads=heads, joint_dim=joint)
    m = QwenImageTransformer2DModel(num_layers=layers, num_attention_heads=heads, joint_attention_dim=joint, device=DEV)
    m.load_weights(Pm.items())
    g = torch.Generator().manual_seed(3)
    lat = bf16_round(torch.randn(128, 64, generator=g))
    txt = bf16_round(torch.randn(T, joint, generator=g))
    sig = torch.tensor([0.4375])
    ref = m(hidden_states=lat.to(DEV, BF16).unsqueeze(0), encoder_hidden_states=txt.to(DEV, BF16).unsqueeze(0),
            timestep=sig.to(DEV), img_shapes=[[grid]], txt_seq_lens=[T], return_dict=False)[0][0]
    outs, n_a2a = _emulate(m, P, lat.to(DEV, BF16), txt.to(DEV, BF16), sig.to(DEV), grid)
    torch.cuda.synchronize()
    assert n_a2a == 2 * layers
    for o in outs:
        assert torch.equal(o, outs[0])                                 # every rank ends with the same full prediction
    e = rel_l2(outs[0], ref)
    oracle = O.dit_forward({k: bf16_round(v) for k, v in Pm.items()}, lat.unsqueeze(0), txt.unsqueeze(0), sig, grid,
                           num_heads=heads)[0]
    print(f"P={P}: SP vs single-device rel_l2 {e:.3e}; SP vs fp32 oracle {rel_l2(outs[0], oracle):.3e}")
    assert e <= 4e-3                                                   # same kernels; attention sees another head/tile grouping
    assert rel_l2(outs[0], oracle) <= 1e-2
    if P == 1:
        assert torch.equal(m.forward_sp(lat.to(DEV, BF16), txt.to(DEV, BF16), sig.to(DEV), grid), outs[0])


def test_denoise_loop_on_the_sequence_parallel_path_equals_the_plain_loop():
    """`ulysses_degree > 1` routes `_denoise` through `_denoise_sp` (round-2 verdict: the degree used to be accepted and
    ignored).  On one device the group is a group of one: every kernel, the pipelined driver and the CFG/Euler update of the SP
    loop run; the result must equal the ragged-batch loop (reference contract: SP == non-SP,
    tests/e2e/offline_inference/test_sequence_parallel.py:128-147)."""
    import _gpu_factory
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest

    pipe = _gpu_factory.make_small_pipeline()
    g = torch.Generator().manual_seed(21)
    reqs = [OmniDiffusionRequest(height=128, width=128, num_inference_steps=4, true_cfg_scale=4.0, output_type="latent",
                                 latents=torch.randn(1, 64, 64, generator=g).to(BF16),
                                 prompt_embeds=torch.randn(1, T, 128, generator=g).to(BF16),
                                 negative_prompt_embeds=torch.randn(1, Tn, 128, generator=g).to(BF16))
            for T, Tn in ((7, 3), (19, 12))]
    plain = [o.output for o in pipe.generate(reqs, output_type="latent")]
    pipe._force_sp_path = True
    try:
        sp = [o.output for o in pipe.generate(reqs, output_type="latent")]
    finally:
        pipe._force_sp_path = False
    for a, b in zip(plain, sp):
        e = rel_l2(b, a)
        print(f"SP-path loop vs ragged loop rel_l2 {e:.3e}")
        assert e <= 5e-3


# ---------------------------------------------------------------------------------------------------------------------
# Real ranks over RCCL: runs whenever the box shows >= 2 devices (the driver's SCALE node), skipped on 1-GPU boxes.
def _nccl_rank(rank, world, port, q):
    import os
    import sys

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here, os.path.join(os.path.dirname(here), "oracle")]
    import torch.distributed as dist

    import _gpu_factory
    from vllm_omni_amd.diffusion.data import DiffusionParallelConfig, OmniDiffusionConfig
    from vllm_omni_amd.diffusion.request import OmniDiffusionRequest
    from vllm_omni_amd.diffusion.worker.gpu_worker import GPUWorker

    try:
        torch.cuda.set_device(rank)
        g = torch.Generator().manual_seed(33)
        mk = lambda steps, T, Tn, seed: OmniDiffusionRequest(                                       # noqa: E731
            height=128, width=128, num_inference_steps=steps, true_cfg_scale=4.0, output_type="latent", seed=seed,
            latents=torch.randn(1, 64, 64, generator=g).to(BF16), prompt_embeds=torch.randn(1, T, 128, generator=g).to(BF16),
            negative_prompt_embeds=torch.randn(1, Tn, 128, generator=g).to(BF16))
        reqs = [mk(4, 7, 3, 0), mk(4, 19, 12, 1), mk(3, 5, 5, 2)]
        # (1) Ulysses over the two ranks: one SP group, every request sequence-parallel, rank 0 answers
        cfg = OmniDiffusionConfig(dist_timeout=120, max_step_batch=4, parallel_config=DiffusionParallelConfig(ulysses_degree=world))
        w = GPUWorker(rank, rank, cfg, pipeline=_gpu_factory.make_small_pipeline())
        w.init_device_and_model()
        sp = w.execute_model(reqs, decode=False)
        # (2) the same worker re-wired as plain data parallel over the same ranks (requests sharded, latents all-gathered)
        w.sp_degree, w.sp_group, w.dp_rank, w.dp_world = 1, None, rank, world
        w.pipeline.sp_group, w.pipeline.sp_degree = None, 1
        dpo = w.execute_model(reqs, decode=False)
        # (3) single-rank reference on rank 0
        solo = [o.output for o in w.pipeline.generate(reqs, output_type="latent")] if rank == 0 else None
        torch.cuda.synchronize()
        q.put((rank, sp.error, dpo.error,
               None if sp.output is None else sp.output.float().cpu(), None if dpo.output is None else dpo.output.float().cpu(),
               None if solo is None else torch.cat(solo).float().cpu()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}", None, None, None, None))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL)")
def test_two_real_ranks_over_rccl_sequence_parallel_and_data_parallel_equal_single_rank():
    import torch.multiprocessing as mp

    from test_host_logic import _free_port

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=600)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
    assert res[0][0] is None and res[0][1] is None and res[1][0] is None and res[1][1] is None, (res[0][:2], res[1][:2])
    sp, dpo, solo = res[0][2], res[0][3], res[0][4]
    assert res[1][2] is None and res[1][3] is None                     # only the output rank holds results
    e_sp, e_dp = rel_l2(sp.reshape(solo.shape), solo), rel_l2(dpo.reshape(solo.shape), solo)
    print(f"2 ranks on RCCL: Ulysses vs single rank rel_l2 {e_sp:.3e}; data-parallel vs single rank {e_dp:.3e}")
    assert e_sp <= 5e-3 and e_dp <= 5e-3
