"""Multi-GPU data parallelism for the render-and-trace path: one process per GPU, camera batch sharded over
ranks, one exchange of the flat per-Gaussian gradient buffer per step (RCCL over xGMI; `nccl` backend on ROCm).

The reference's only parallelism is DDP (easyvolcap/scripts/main.py:240-275), which cannot follow the
nn.Parameter replacement done by densification (gaussian2d_utils.py:526-621), so the
exchange step is explicit here: every rank holds a full replica of the Gaussian sets, renders its share of
the view batch, and the summed gradients + densification statistics are made identical on all ranks so that
every rank takes the same densify / prune decisions (gaussian2d_utils.py:901-909).

Message size: 60 floats (240 B) per base Gaussian, 58 per env Gaussian -> 72 MB at P = 300 k.  xGMI is a full mesh of
point-to-point links (7 x ~153 GB/s per GPU), so the exchange is written as what the topology is good at
(SURVEY.md section 8e): a DIRECT reduce-scatter (every rank sends chunk j straight to rank j: 7 transfers on 7 different
links, one local sum) followed by a direct all-gather -- 2 x S/8 bytes per link instead of a ring's 2 x 7/8 x S through one.
`GradExchange` keeps ONE persistent flat fp32 buffer per bucket that the parameters' `.grad` are views of: autograd
accumulates straight into the message, nothing is packed or unpacked.
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
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (read when the HSA runtime starts: effective if no HIP call has been made yet)
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


def _active(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def shard_views(num_views, rank, world):
    """Views of an `num_views`-camera batch owned by `rank` (round-robin, so any world size divides the work)."""
    return list(range(rank, num_views, world))


# ------------------------------------------------------------------------------------------------------------------
# the collective itself

_SIDE_STREAMS = {}


def _side_stream(dev):
    """One communication stream per device: an exchange launched from a backward hook is ordered on it (after the gradient kernels already
    queued on the compute stream, before nothing of the compute stream), so the rest of backward() keeps running underneath."""
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        s = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


def exchange_flat(flat, average=True, group=None, algo="direct", async_op=False, recv=None, mine=None):
    """Sum (or average) the flat fp32 buffer `flat` over the ranks, in place.  len(flat) must be a multiple of the world size.

    algo = "direct": reduce-scatter + all-gather, each ONE step over the full xGMI mesh:
        all_to_all_single (chunk j of every rank -> rank j; 7 concurrent point-to-point transfers per GPU) -> local sum of the world
        received chunks (one torch kernel over S/world floats x world) -> all_gather_into_tensor of the owned chunk back into `flat`.
    algo = "allreduce": one `all_reduce` (the library picks the algorithm; a ring is per-link bound on xGMI).
    recv (len(flat)) / mine (len(flat) / world): optional persistent scratch (GradExchange keeps one pair per bucket; allocated here otherwise).
    Returns a callable that completes the exchange; with async_op=False it has already been called.

    GPU buffers, async_op=True: the WHOLE exchange -- all-to-all, local sum, scale, all-gather -- is queued at launch time on a communication
    stream that waits for the work already queued on the current stream, so bucket 0 (the environment set, final when the tracer's backward
    is done) is exchanged completely while the base rasterizer is still differentiating; the returned callable only makes the current stream
    wait for it.  (Round 2 queued the all-to-all alone and did the sum + all-gather in finish().)  CPU buffers (gloo): the collectives'
    own async work objects."""
    world = dist.get_world_size(group)
    n = flat.numel()
    assert flat.dtype == torch.float32 and flat.is_contiguous()
    if algo not in ("direct", "allreduce"):
        raise ValueError("algo must be 'direct' or 'allreduce', got %r" % (algo,))
    if algo == "direct" and n % world != 0:
        raise ValueError("exchange_flat('direct') needs a buffer length that is a multiple of the world size (%d %% %d != 0)" % (n, world))
    scale = (1.0 / world) if average else 1.0
    if algo == "direct":
        if recv is None or recv.numel() != n or recv.device != flat.device:
            recv = torch.empty_like(flat)
        if mine is None or mine.numel() != n // world or mine.device != flat.device:
            mine = torch.empty(n // world, dtype=torch.float32, device=flat.device)

    def whole():
        if algo == "allreduce":
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if scale != 1.0:
                flat.mul_(scale)
        else:
            dist.all_to_all_single(recv, flat, group=group)
            torch.sum(recv.view(world, n // world), dim=0, out=mine)   # chunk `rank` of every peer, summed: this rank's share of the result
            if scale != 1.0:
                mine.mul_(scale)
            dist.all_gather_into_tensor(flat, mine, group=group)

    if not async_op:
        whole()
        return lambda: None
    if flat.is_cuda:
        cur = torch.cuda.current_stream(flat.device)
        side = _side_stream(flat.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            whole()
        # scratch allocated HERE (direct API use; GradExchange passes persistent buffers) was allocated on the compute stream and is used on
        # the side stream: tell the caching allocator, and keep the tensors alive until the caller has waited
        for t in (recv, mine) if algo == "direct" else ():
            t.record_stream(side)
        flat.record_stream(side)
        held = (recv, mine) if algo == "direct" else ()

        def done(_held=held):
            torch.cuda.current_stream(flat.device).wait_stream(side)
        return done
    # CPU (gloo): asynchronous work objects; the arithmetic between the two collectives runs when the first completes
    if algo == "allreduce":
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)

        def done():
            work.wait()
            if scale != 1.0:
                flat.mul_(scale)
        return done
    w1 = dist.all_to_all_single(recv, flat, group=group, async_op=True)

    def done():
        w1.wait()
        torch.sum(recv.view(world, n // world), dim=0, out=mine)
        if scale != 1.0:
            mine.mul_(scale)
        dist.all_gather_into_tensor(flat, mine, group=group)
    return done


class _Bucket:
    __slots__ = ("params", "ids", "flat", "recv", "mine", "numel", "offsets", "pending", "done", "launched")


class GradExchange:
    """Persistent flat gradient buffers + the (optionally overlapped) exchange.

    `get_buckets` is a CALLABLE returning the current buckets (lists of parameters) in the order their gradients become final (for
    EnvGS: the environment set first -- the tracer's backward runs before the base rasterizer's -- then the base set).  It is called
    again at every `begin_step()`, so parameters replaced by densification / pruning (gaussian2d_utils.py:526-621 makes a fresh
    nn.Parameter for every tensor) are picked up: the layout is rebuilt whenever the identity or size of any parameter changed.

    Per step:
        ex.begin_step()              # (re)binds: p.grad = zeroed view of the bucket's flat buffer; arms the hooks
        loss.backward() [x n]        # autograd accumulates in place into the views; with `overlap`, the LAST backward of the step
                                     # (ex.arm_last() before it if there are several; default: the first) launches bucket i's exchange
                                     # from a post-accumulate hook as soon as its last gradient is final -- always in bucket order
        ex.finish()                  # launches what is left (in order), waits, averages; returns the bytes exchanged
    Every rank issues the collectives in the same order (bucket 0, 1, ...) regardless of which parameters received a gradient, so ranks
    whose views saw nothing of a set cannot dead-lock or mismatch sizes; a parameter without a gradient contributes the zeros of its view.
    With one process (or no process group) only the flat-buffer views are maintained (`p.grad` is still zeroed / valid)."""

    def __init__(self, get_buckets, average=True, group=None, algo="direct", overlap=True):
        self.get_buckets = get_buckets if callable(get_buckets) else (lambda b=get_buckets: b)
        if algo not in ("direct", "allreduce", "auto"):
            raise ValueError("algo must be 'direct', 'allreduce' or 'auto', got %r" % (algo,))
        self.auto = algo == "auto"                       # "auto": the direct form until autotune() has measured both on this machine
        self.tuned = None                                # autotune()'s result
        algo = "direct" if self.auto else algo
        self.average, self.group, self.algo, self.overlap = average, group, algo, overlap
        self.enabled = _active(group)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.buckets = []
        self._hooks = []
        self._armed = False
        self._in_step = False

    # -- layout ------------------------------------------------------------------------------------------------------
    def _signature(self, lists):
        return [[(id(p), tuple(p.shape), p.device) for p in b if p is not None] for b in lists]

    def _rebuild(self, lists):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.buckets = []
        for bi, plist in enumerate(lists):
            plist = [p for p in plist if p is not None]
            B = _Bucket()
            B.params, B.ids = plist, [id(p) for p in plist]
            n = sum(p.numel() for p in plist)
            B.numel = n
            pad = (-n) % self.world
            dev = plist[0].device if plist else torch.device("cpu")
            B.flat = torch.zeros(n + pad, dtype=torch.float32, device=dev)
            # persistent scratch of the direct exchange (round 2 allocated 72 - 162 MB per call): the all-to-all's receive buffer, this rank's reduced chunk
            direct = self.enabled and self.algo == "direct"
            B.recv = torch.empty(n + pad, dtype=torch.float32, device=dev) if direct else None
            B.mine = torch.empty((n + pad) // self.world, dtype=torch.float32, device=dev) if direct else None
            B.offsets, off = [], 0
            for p in plist:
                if p.dtype != torch.float32:
                    raise RuntimeError("GradExchange: fp32 parameters only (got %s)" % p.dtype)
                B.offsets.append(off)
                off += p.numel()
            self.buckets.append(B)
            if self.enabled and self.overlap:
                for p in plist:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._hook(bi)))
        self._sig = self._signature(lists)

    def begin_step(self):
        """Call before the first backward of a step (replaces `p.grad = None` / optimizer.zero_grad())."""
        lists = self.get_buckets()
        if not self.buckets or self._signature(lists) != self._sig:
            self._rebuild(lists)
        for B in self.buckets:
            B.flat.zero_()
            for p, off in zip(B.params, B.offsets):
                v = B.flat[off:off + p.numel()].view(p.shape)
                if p.grad is None or p.grad.data_ptr() != v.data_ptr() or p.grad.shape != v.shape:
                    p.grad = v                                       # autograd accumulates IN PLACE into an existing .grad
            B.pending, B.done, B.launched = len(B.params), None, False
        self._armed = True
        self._in_step = True

    def arm_last(self):
        """Several backward() calls per step (several views per rank): call this right before the LAST one.  Until then the hooks
        only count nothing and launch nothing."""
        for B in self.buckets:
            B.pending = len(B.params)
        self._armed = True

    def hold(self):
        """Disarm the hooks for the backward passes that are not the last one of the step."""
        self._armed = False

    # -- launching ---------------------------------------------------------------------------------------------------
    def _hook(self, bi):
        def fn(param):
            if not (self._armed and self._in_step):
                return
            B = self.buckets[bi]
            if B.launched:
                raise RuntimeError("GradExchange: a gradient of bucket %d arrived after its exchange was launched -- more than one "
                                   "backward() in this step?  Call hold() before the early ones and arm_last() before the last." % bi)
            g = param.grad
            if g is None or g.data_ptr() < B.flat.data_ptr() or g.data_ptr() >= B.flat.data_ptr() + B.flat.numel() * 4:
                raise RuntimeError("GradExchange: a parameter's .grad no longer views the flat buffer (was .grad replaced or set to None "
                                   "after begin_step()?)")
            B.pending -= 1
            self._launch_ready()
        return fn

    def _launch_ready(self):
        # fixed order on every rank: bucket i only after bucket i-1
        for B in self.buckets:
            if B.launched:
                continue
            if B.pending > 0:
                break
            self._launch(B)

    def _launch(self, B):
        B.launched = True
        if B.flat.numel():
            B.done = exchange_flat(B.flat, average=self.average, group=self.group, algo=self.algo, async_op=True, recv=B.recv, mine=B.mine)

    def finish(self):
        """Call after the last backward(): launches the remaining buckets in order, completes all of them, returns the bytes exchanged."""
        self._in_step = False
        if not self.enabled:
            return 0
        if self.auto and self.tuned is None and not self.__dict__.get("_warned_auto"):
            import warnings
            self._warned_auto = True
            warnings.warn("GradExchange(algo='auto'): autotune() has not been called -- using the direct reduce-scatter + all-gather form; call "
                          "autotune() between two steps to let the machine choose")
        nbytes = 0
        for B in self.buckets:
            if not B.launched:
                self._launch(B)
        for B in self.buckets:
            if B.done is not None:
                B.done()
                B.done = None
            nbytes += B.flat.numel() * 4
        return nbytes

    def autotune(self, reps=3):
        """Measure BOTH exchange forms on this exchange's own flat buffers and keep the faster one (collective: every rank calls it at the same
        point, BETWEEN steps -- after the optimizer consumed the gradients and before the next begin_step(), which zeroes the buffers anyway: their
        contents are scratch here; the decision is taken on the MAX over ranks, so all ranks switch together).  The direct form assumes that RCCL's
        all-to-all drives the seven xGMI links of a GPU concurrently -- an assumption no run had checked when this was
        written; `algo="auto"` callers (bench.py's default for N > 1) let the first steps on real hardware decide instead: "auto" means the direct
        form UNTIL this method has run (finish() warns once if it never does).  No second copy of the buckets is made: the direct
        form's receive / reduce scratch is the persistent one when it exists and becomes the persistent one if the direct form wins.  Returns
        {"direct": ms, "allreduce": ms, "chosen": name} (None with one process)."""
        import time
        if not self.enabled:
            return None
        reps = max(1, int(reps))
        if not self.buckets:
            self._rebuild(self.get_buckets())
        dev = self.buckets[0].flat.device if self.buckets else torch.device("cpu")

        def sync():
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            dist.barrier(group=self.group)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
        scratch = [((B.recv if B.recv is not None else torch.empty_like(B.flat)),
                    (B.mine if B.mine is not None else torch.empty(B.flat.numel() // self.world, dtype=torch.float32, device=B.flat.device))) for B in self.buckets]
        ms = {}
        for algo in ("direct", "allreduce"):
            t0 = None
            for it in range(reps + 1):
                if it == 1:
                    sync(); t0 = time.perf_counter()
                for B, (rv, mn) in zip(self.buckets, scratch):
                    if B.flat.numel():
                        exchange_flat(B.flat, average=self.average, group=self.group, algo=algo, recv=rv, mine=mn)
            sync()
            ms[algo] = (time.perf_counter() - t0) / reps * 1e3
        tt = torch.tensor([ms["direct"], ms["allreduce"]], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.group)
        ms = {"direct": float(tt[0]), "allreduce": float(tt[1])}
        self.algo = "direct" if ms["direct"] <= ms["allreduce"] else "allreduce"
        for B, (rv, mn) in zip(self.buckets, scratch):                                # the direct form's persistent scratch, if it won
            B.recv, B.mine = (rv, mn) if self.algo == "direct" else (None, None)
        ms["chosen"] = self.algo
        self.tuned = ms
        return ms

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


# ------------------------------------------------------------------------------------------------------------------
# one-shot helpers (no persistent state)

def allreduce_grads(tensors, average=True, group=None, algo="allreduce"):
    """Sum (or average) the .grad of every tensor in `tensors` across ranks with ONE exchange of a flat bucket (packed here: the
    stateless form, for callers that do not keep a `GradExchange`).  Tensors whose grad is None contribute zeros (a rank whose views
    saw nothing of a Gaussian).  Returns the bytes exchanged."""
    if not _active(group):
        return 0
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return 0
    world = dist.get_world_size(group)
    n = sum(t.numel() for t in tensors)
    flat = torch.zeros(n + (((-n) % world) if algo == "direct" else 0), dtype=torch.float32, device=tensors[0].device)
    off = 0
    for t in tensors:
        if t.grad is not None:
            flat[off:off + t.numel()].copy_(t.grad.reshape(-1))
        off += t.numel()
    exchange_flat(flat, average=average, group=group, algo=algo)
    off = 0
    for t in tensors:
        new = flat[off:off + t.numel()].view(t.shape)
        if t.grad is None:
            t.grad = new.to(t.dtype).clone()
        else:
            t.grad.copy_(new)
        off += t.numel()
    return flat.numel() * 4


_STATS_FLAT = {}


def allreduce_densify_stats(grad_norm_accum, denom, weight_accum, max_radii, group=None):
    """Make the densification statistics identical on every rank: sums for the accumulators
    (gaussian2d_utils.py:901-909), max for the screen radii (gaussian2d_sampler.py:330-332).  One persistent flat buffer per
    (device, size) for the three sums (copied in and out by slices: no torch.cat, no per-call allocation); the radii take their own MAX."""
    if not _active(group):
        return
    parts = (grad_norm_accum, denom, weight_accum)
    n = sum(p.numel() for p in parts)
    key = (grad_norm_accum.device, n)
    flat = _STATS_FLAT.get(key)
    if flat is None:
        if len(_STATS_FLAT) > 4:
            _STATS_FLAT.clear()                                   # P changed (densify / prune): drop the stale sizes
        flat = _STATS_FLAT[key] = torch.empty(n, dtype=torch.float32, device=grad_norm_accum.device)
    off = 0
    for p in parts:
        flat[off:off + p.numel()].copy_(p.reshape(-1))
        off += p.numel()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p in parts:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    r = max_radii.float().contiguous()
    dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
    max_radii.copy_(r.to(max_radii.dtype))
