"""Multi-GPU data parallelism for the render-and-trace path: one process per GPU, camera batch sharded over
ranks, ONE all-reduce of the flat per-Gaussian gradient buffer per step (RCCL over xGMI; `nccl` backend on ROCm).

The reference's only parallelism is DDP (easyvolcap/scripts/main.py:240-275), which cannot follow the
nn.Parameter replacement done by densification (gaussian2d_utils.py:526-621), so the
exchange step is explicit here: every rank holds a full replica of the Gaussian sets, renders its share of
the view batch, and the summed gradients + densification statistics are made identical on all ranks so that
every rank takes the same densify / prune decisions (gaussian2d_utils.py:901-909).

Message size: 60 floats (240 B) per base Gaussian, 58 per env Gaussian -> 72 MB at P = 300 k: one flat bucket,
one collective (xGMI is point-to-point; fewer, larger messages are the cheap ones).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_views(num_views, rank, world):
    """Views of an `num_views`-camera batch owned by `rank` (round-robin, so any world size divides the work)."""
    return list(range(rank, num_views, world))


def allreduce_grads(tensors, average=True, group=None):
    """Sum (or average) the .grad of every tensor in `tensors` across ranks with ONE collective on a flat bucket.
    Tensors whose grad is None contribute zeros (a rank whose views saw nothing of a Gaussian)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return 0
    grads = [t.grad if t.grad is not None else torch.zeros_like(t) for t in tensors]
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(dist.get_world_size(group))
    off = 0
    for t, g in zip(tensors, grads):
        n = g.numel()
        new = flat[off:off + n].view_as(g).to(g.dtype)
        if t.grad is None:
            t.grad = new.clone()
        else:
            t.grad.copy_(new)
        off += n
    return flat.numel() * 4


def allreduce_densify_stats(grad_norm_accum, denom, weight_accum, max_radii, group=None):
    """Make the densification statistics identical on every rank: sums for the accumulators
    (gaussian2d_utils.py:901-909), max for the screen radii (gaussian2d_sampler.py:330-332)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([grad_norm_accum.reshape(-1).float(), denom.reshape(-1).float(), weight_accum.reshape(-1).float()])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    n = grad_norm_accum.numel()
    grad_norm_accum.copy_(flat[:n].view_as(grad_norm_accum))
    denom.copy_(flat[n:n + denom.numel()].view_as(denom))
    weight_accum.copy_(flat[n + denom.numel():].view_as(weight_accum))
    r = max_radii.float().contiguous()
    dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
    max_radii.copy_(r.to(max_radii.dtype))
