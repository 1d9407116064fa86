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
