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
            backend = os.environ.get("ENVGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


class OverlappedGradReducer:
    """The gradient exchange of `allreduce_grads`, started from inside the backward pass.

    `buckets` = lists of parameters in the order their gradients become final (for EnvGS: the environment set first -- the tracer's
    backward runs before the base rasterizer's -- then the base set).  A post-accumulate-grad hook on every parameter launches the
    bucket's ONE flat all-reduce (async) as soon as its last gradient has been accumulated, so the environment bucket travels over xGMI
    while the base pass is still differentiating; `finish()` waits, averages and writes the results back into `.grad`.  Numerically
    identical to `allreduce_grads` bucket by bucket.  With a single process (or no process group) it does nothing."""

    def __init__(self, buckets, average=True, group=None):
        self.buckets = [[t for t in b if t is not None] for b in buckets]
        self.average, self.group = average, group
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self._handles = []
        self._reset()
        if self.enabled:
            for bi, b in enumerate(self.buckets):
                for t in b:
                    self._handles.append(t.register_post_accumulate_grad_hook(self._hook(bi)))

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._inflight = [None] * len(self.buckets)

    def _hook(self, bi):
        def fn(param):
            self._pending[bi] -= 1
            if self._pending[bi] == 0 and self._inflight[bi] is None:
                self._launch(bi)
        return fn

    def _launch(self, bi):
        b = self.buckets[bi]
        if not b:
            return
        grads = [t.grad if t.grad is not None else torch.zeros_like(t) for t in b]
        flat = torch.cat([g.reshape(-1).float() for g in grads])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight[bi] = (flat, grads, work)

    def finish(self):
        """Call after backward(): completes every bucket (launching those whose hooks did not all fire -- a parameter without a
        gradient this step) and returns the bytes exchanged."""
        if not self.enabled:
            return 0
        nbytes = 0
        world = dist.get_world_size(self.group)
        for bi, b in enumerate(self.buckets):
            if not b:
                continue
            if self._inflight[bi] is None:
                self._launch(bi)
            flat, grads, work = self._inflight[bi]
            work.wait()
            if self.average:
                flat.div_(world)
            off = 0
            for t, g in zip(b, grads):
                n = g.numel()
                new = flat[off:off + n].view_as(g).to(g.dtype)
                if t.grad is None:
                    t.grad = new.clone()
                else:
                    t.grad.copy_(new)
                off += n
            nbytes += flat.numel() * 4
        self._reset()
        return nbytes

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


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
