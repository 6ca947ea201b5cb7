"""CPU, world_size 2, gloo: the N>1 path of bench.py -- camera sharding and the single flat gradient all-reduce --
plus the densification-statistics exchange that keeps every rank's densify / prune decisions identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from envgs_amd import dist as edist
    r, w, l = edist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    views = edist.shard_views(8, r, w)
    # two "parameter tensors" with rank-dependent grads; one has no grad on rank 1 (its views saw nothing)
    torch.manual_seed(0)
    a = torch.zeros(5, 3, requires_grad=True); b = torch.zeros(7, requires_grad=True)
    a.grad = torch.full((5, 3), float(rank + 1))
    if rank == 0: b.grad = torch.arange(7.0)
    nbytes = edist.allreduce_grads([a, b, None], average=True)
    st = [torch.full((4, 1), float(rank + 1)), torch.full((4, 1), 1.0), torch.full((4,), 0.5 * (rank + 1)), torch.tensor([1, 9, 3, 4]) if rank == 0 else torch.tensor([5, 2, 3, 8])]
    edist.allreduce_densify_stats(*st)
    # plain lists, not tensors: a tensor in a Queue travels as a shared-memory file that is gone if this process exits before the parent reads it
    q.put((rank, views, a.grad.tolist(), b.grad.tolist(), nbytes, [t.tolist() for t in st]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_allreduce_and_view_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    v0, v1 = res[0][1], res[1][1]
    assert sorted(v0 + v1) == list(range(8)) and not set(v0) & set(v1)          # the 8-view batch is partitioned
    for rank, _, ga, gb, nbytes, st in res:
        ga, gb, st = torch.tensor(ga), torch.tensor(gb), [torch.tensor(t) for t in st]
        assert torch.allclose(ga, torch.full((5, 3), 1.5))                      # mean of 1 and 2
        assert torch.allclose(gb, torch.arange(7.0) / 2)                        # rank 1 contributed zeros
        assert nbytes == (15 + 7) * 4                                           # ONE flat bucket
        assert torch.allclose(st[0], torch.full((4, 1), 3.0)) and torch.allclose(st[1], torch.full((4, 1), 2.0))
        assert torch.allclose(st[2], torch.full((4,), 1.5)) and st[3].tolist() == [5, 9, 3, 8]


def _worker_overlap(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from envgs_amd import dist as edist
    edist.init_from_env(backend="gloo")
    torch.manual_seed(1)
    env = [torch.randn(6, 3, requires_grad=True), torch.randn(6, requires_grad=True)]
    base = [torch.randn(4, 2, requires_grad=True), torch.randn(4, requires_grad=True), torch.randn(3, requires_grad=True)]   # the last one stays unused
    red = edist.OverlappedGradReducer([env, base], average=True)
    out = []
    for step in range(2):                                            # hooks must re-arm every step
        for t in env + base:
            t.grad = None
        x = float(rank + 1 + step)
        loss = (env[0] * x).sum() + (env[1] * env[1] * x).sum() + (base[0] * 2 * x).sum() + (base[1] * x).sum() + (env[0] * base[0].sum()).sum()
        loss.backward()
        nbytes = red.finish()
        out.append((nbytes, [None if t.grad is None else t.grad.tolist() for t in env + base]))     # plain lists: no shared-memory handles
    red.remove()
    q.put((rank, out, [t.detach().tolist() for t in env + base]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_overlapped_reducer_matches_plain_average_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=90) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    vals = [torch.tensor(v) for v in res[0][2]]                      # same seed: identical parameters on both ranks
    for step in range(2):
        xs = [1.0 + step, 2.0 + step]
        xm = sum(xs) / 2
        exp = [torch.full((6, 3), xm) + vals[2].sum(), 2 * vals[1] * xm, torch.full((4, 2), 2 * xm) + vals[0].sum(), torch.full((4,), xm), torch.zeros(3)]
        for rank, out, _ in res:
            nbytes, grads = out[step]
            assert nbytes == (18 + 6 + 8 + 4 + 3) * 4                # two flat buckets
            for g, e in zip(grads, exp):
                assert g is not None and torch.allclose(torch.tensor(g), e, atol=1e-6), (step, rank, g, e)


def test_single_process_is_a_noop():
    from envgs_amd import dist as edist
    a = torch.zeros(3, requires_grad=True); a.grad = torch.ones(3)
    assert edist.allreduce_grads([a]) == 0 and torch.equal(a.grad, torch.ones(3))
    assert edist.shard_views(8, 0, 1) == list(range(8))
