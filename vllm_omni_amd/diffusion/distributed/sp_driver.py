"""Driver for sequence-parallel forwards written as GENERATORS that yield their collectives.

`QwenImageTransformer2DModel._sp_forward_gen` (reference attention/parallel/ulysses.py:59-135,
qwen_image_transformer.py:735-742,776-781,800-801) computes up to an exchange point, yields
`("all_to_all", send[P, ...])`, `("all_gather", x)` or `("all_reduce", x)` (sum; TeaCache's two partial sums), and continues
with the received tensor.  This module runs one or
SEVERAL such generators over a process group:

  * one generator: collective, continue, collective, ... (what the reference does: its all-to-alls sit on the critical path);
  * several (the two true-CFG branches of a request, or several requests): SOFTWARE-PIPELINED.  The collective of forward A is
    launched asynchronously (RCCL runs it on its own stream) and, while it is in flight, the compute segment of forward B is
    enqueued on the main stream; then A's result is awaited and A's next segment enqueued while B's collective flies, and so
    on.  Every all-to-all of one branch hides behind the other branch's GEMMs — the overlap §8f N2 asks for, without
    splitting any kernel.  All ranks of the group drive the same generators in the same order, so the collectives match up.

Works on RCCL ("nccl") and gloo; with no process group (or a group of one) every collective is the identity."""
from __future__ import annotations

from typing import Any, Generator

import torch
import torch.distributed as dist


class _Done:
    def __init__(self, out):
        self.out = out

    def wait(self):
        return self.out


def _group_size(group) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _launch(msg, group):
    """Start one collective; returns an object whose .wait() yields the received tensor."""
    kind, t = msg
    P = _group_size(group)
    if kind == "all_to_all":
        if P == 1:
            return _Done(t)
        out = torch.empty_like(t)
        work = dist.all_to_all_single(out, t, group=group, async_op=True)
    elif kind == "all_gather":
        if P == 1:
            return _Done(t.unsqueeze(0))
        t = t.contiguous()
        flat = torch.empty((P * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # concatenation along dim 0
        work = dist.all_gather_into_tensor(flat, t, group=group, async_op=True)
        out = flat.view((P,) + tuple(t.shape))
    elif kind == "all_reduce":
        if P == 1:
            return _Done(t)
        out = t.contiguous().clone()
        work = dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group, async_op=True)
    else:
        raise RuntimeError(f"unknown collective {kind!r}")

    class _Pending:
        def wait(self_inner):
            work.wait()                    # nccl: the current stream waits for the collective; gloo: blocks the host
            return out

    return _Pending()


def drive(gens: list[Generator], group=None) -> list[Any]:
    """Run the generators to completion over `group`, pipelined as described above; returns their return values."""
    n = len(gens)
    results: list[Any] = [None] * n
    pending: list[Any] = [None] * n
    alive = [True] * n
    for i, g in enumerate(gens):           # first segment of every forward; its collective starts right away
        try:
            pending[i] = _launch(next(g), group)
        except StopIteration as e:
            results[i], alive[i] = e.value, False
    while any(alive):
        for i, g in enumerate(gens):
            if not alive[i]:
                continue
            got = pending[i].wait()
            try:
                pending[i] = _launch(g.send(got), group)
            except StopIteration as e:
                results[i], alive[i] = e.value, False
    return results


def drive_in_process(gens_by_rank: list[list[Generator]]) -> list[Any]:
    """Test harness for ONE device: `gens_by_rank[r][i]` is forward i on virtual rank r.  The P generators of a forward run in
    lock step and their collectives are performed in process (all-to-all: recv[r][j] = send[j][r]; all-gather: the stack;
    all-reduce: the sum) — every kernel, reshard and index permutation of the real path runs, only the wire is replaced.
    Every rank must yield the same collective kind at the same point and end with the same result (checked); returns virtual
    rank 0's results."""
    P = len(gens_by_rank)
    results = []
    for i in range(len(gens_by_rank[0])):
        gens = [gens_by_rank[r][i] for r in range(P)]
        msgs, done = [], []
        for g in gens:
            try:
                msgs.append(next(g))
            except StopIteration as e:
                done.append(e.value)
        while not done:
            kinds = {m[0] for m in msgs}
            if len(kinds) != 1:
                raise RuntimeError(f"virtual ranks disagree on the next collective: {kinds}")
            kind, sends = kinds.pop(), [m[1] for m in msgs]
            if kind == "all_to_all":
                outs = [torch.stack([sends[j][r] for j in range(P)]) for r in range(P)]
            elif kind == "all_gather":
                outs = [torch.stack(sends) for _ in range(P)]
            elif kind == "all_reduce":
                tot = torch.stack(sends).sum(0)
                outs = [tot.clone() for _ in range(P)]
            else:
                raise RuntimeError(f"unknown collective {kind!r}")
            msgs = []
            for g, o in zip(gens, outs):
                try:
                    msgs.append(g.send(o))
                except StopIteration as e:
                    done.append(e.value)
        if len(done) != P:
            raise RuntimeError("virtual ranks finished at different points")
        for o in done[1:]:
            if not torch.equal(o, done[0]):
                raise RuntimeError("virtual ranks ended with different results")
        results.append(done[0])
    return results
