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


def _worker_exchange(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from envgs_amd import dist as edist
    edist.init_from_env(backend="gloo")
    torch.manual_seed(1)
    sets = dict(env=[torch.nn.Parameter(torch.randn(6, 3)), torch.nn.Parameter(torch.randn(6))],
                base=[torch.nn.Parameter(torch.randn(4, 2)), torch.nn.Parameter(torch.randn(4)), torch.nn.Parameter(torch.randn(3))])   # base[2] stays unused
    ex = edist.GradExchange(lambda: [sets["env"], sets["base"]], average=True, algo="direct", overlap=True)
    out = []
    tune = None
    for step in range(5):
        if step == 4:
            tune = ex.autotune(reps=1)         # both exchange forms measured between two steps; every rank keeps the same (faster) one
        if step == 2:
            # "densification": every tensor of the base set is replaced by a fresh, LONGER nn.Parameter (gaussian2d_utils.py:526-621)
            sets["base"] = [torch.nn.Parameter(torch.cat([p.detach(), p.detach()[:1]])) for p in sets["base"]]
        env, base = sets["env"], sets["base"]
        ex.begin_step()
        x = float(rank + 1 + step)
        if step == 3:
            # two backward passes in one step (two views on this rank); on rank 1 the env set gets NO gradient from the last one, so its
            # bucket can only be launched by finish() -- the launch order must still be env, base on both ranks
            ex.hold()
            ((env[0] * x).sum() + (env[1] * env[1] * x).sum()).backward()
            ex.arm_last()
            l2 = (base[0] * 2 * x).sum() + (base[1] * x).sum()
            if rank == 0:
                l2 = l2 + (env[0] * base[0].sum()).sum()
            l2.backward()
        else:
            loss = (env[0] * x).sum() + (env[1] * env[1] * x).sum() + (base[0] * 2 * x).sum() + (base[1] * x).sum() + (env[0] * base[0].sum()).sum()
            loss.backward()
        nbytes = ex.finish()
        assert all(p.grad is not None and p.grad.data_ptr() >= B.flat.data_ptr() for B in ex.buckets for p in B.params)    # still views
        out.append((nbytes, [t.grad.tolist() for t in env + base], [t.detach().tolist() for t in env + base]))
    # the stateless helper with the direct algorithm equals the plain all-reduce
    a = torch.zeros(5, 3, requires_grad=True); a.grad = torch.full((5, 3), float(rank + 1))
    edist.allreduce_grads([a], average=False, algo="direct")
    ex.remove()
    q.put((rank, out, a.grad.tolist(), tune))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])
def test_grad_exchange_flat_views_overlap_densify_and_multi_backward(world):
    """GradExchange: persistent flat buffers the .grad tensors view, direct reduce-scatter + all-gather, hooks that launch in a fixed
    bucket order, parameters replaced by densification, several backward passes per step, grad-presence that differs between ranks.
    world = 8: the rank count of BASELINE configs[3], over gloo on the host -- chunking, padding to the world size and
    the fixed launch order with eight ranks, not just two."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:                                                                                                         # decided on the MAX over ranks: identical everywhere
        assert r[3] is not None and r[3] == res[0][3] and r[3]["chosen"] in ("direct", "allreduce") and r[3]["direct"] > 0 and r[3]["allreduce"] > 0
    pad = lambda n: n + (-n) % world
    for step in range(5):
        vals = [torch.tensor(v) for v in res[0][1][step][2]]             # same seed: identical parameters on every rank
        for r in res[1:]:
            assert all(torch.equal(torch.tensor(a), torch.tensor(b)) for a, b in zip(res[0][1][step][2], r[1][step][2]))
        xm = sum(r + 1.0 + step for r in range(world)) / world
        nb = vals[2].shape[0]
        cross_env = vals[2].sum() * (1.0 / world if step == 3 else 1.0)   # step 3: only rank 0 has the coupling term
        cross_base = vals[0].sum() * (1.0 / world if step == 3 else 1.0)
        exp = [torch.full((6, 3), xm) + cross_env, 2 * vals[1] * xm, torch.full((nb, 2), 2 * xm) + cross_base, torch.full((nb,), xm), torch.zeros(nb - 1)]
        for rank, out, _, _ in res:
            nbytes, grads, _ = out[step]
            assert nbytes == (pad(18 + 6) + pad(4 * nb - 1)) * 4                     # two flat buckets (padded to the world size)
            for g, e in zip(grads, exp):
                assert torch.allclose(torch.tensor(g), e, atol=1e-5), (step, rank, g, e)
    for rank, _, ga, _ in res:
        assert torch.allclose(torch.tensor(ga), torch.full((5, 3), world * (world + 1) / 2.0))


def test_single_process_is_a_noop():
    from envgs_amd import dist as edist
    a = torch.zeros(3, requires_grad=True); a.grad = torch.ones(3)
    assert edist.allreduce_grads([a]) == 0 and torch.equal(a.grad, torch.ones(3))
    assert edist.shard_views(8, 0, 1) == list(range(8))
