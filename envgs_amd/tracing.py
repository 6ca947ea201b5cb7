"""Host-side mirror of the reference's `diff_surfel_tracing` interface, over the C-ABI of include/envgs_trace.h.

Same names, argument meaning and error behaviour as the extension the reference imports at
easyvolcap/utils/optix_utils.py:7 and drives at :24, :78, :104-119 and :188-201:

    SurfelTracingSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                          sh_degree, campos, prefiltered, debug, max_trace_depth, specular_threshold)
    tracer = SurfelTracer()
    tracer.build_acceleration_structure(vertices, faces, rebuild=True)
    rgb, dpt, acc, norm, dist, aux, mid, wet = tracer(ray_o, ray_d, v, means3D=..., grads3D=..., shs=..., colors_precomp=...,
                                                      others_precomp=..., opacities=..., scales=..., rotations=...,
                                                      cov3D_precomp=..., tracer_settings=..., start_from_first=...)

Outputs are channels-last with the ray tensor's leading shape: rgb (...,3), dpt (...,1), acc (...,1), norm (...,3),
dist (...,1), aux (...,2), mid (...,16*(max_trace_depth+1)), wet (P,1).  Differentiable inputs: ray_o, ray_d, means3D,
grads3D (gradient sink for the densification signal), shs | colors_precomp, others_precomp, opacities, scales, rotations.

OptiX is replaced by a hand-written HIP LBVH (Morton build every call with rebuild=True, like the reference rebuilds
its GAS every training iteration) and a persistent-wavefront traversal.  No fallback path exists.
"""
from typing import NamedTuple

import torch
from torch import nn

from . import _lib


class SurfelTracingSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    max_trace_depth: int
    specular_threshold: float


LAST_STATS = {}


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream(dev):
    return _lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def build_bvh(vertices, opacities=None, debug=False, refit=None):
    """LBVH over the (4P,3) quad vertices (leaf boxes tightened by `opacities` when given).  Returns (nodes: envgs_bvh_node_floats(P) floats, P).
    refit: the `nodes` tensor of an earlier build over the same number of surfels -- its topology is kept and only the boxes are recomputed, in
    place (envgs_bvh_refit; OptiX's "update")."""
    lib = _lib.load()
    v = _f32c(vertices.detach())
    if v.device.type != "cuda":
        raise RuntimeError("envgs_amd tracer needs tensors on the GPU (got %s); there is no CPU path" % v.device)
    if v.dim() != 2 or v.shape[1] != 3 or v.shape[0] % 4 != 0:
        raise RuntimeError("vertices must be (4P,3) in the get_disks layout, got %s" % (tuple(v.shape),))
    P = v.shape[0] // 4
    dev = v.device
    nf = lib.envgs_bvh_node_floats(P)
    if refit is not None and (refit.numel() != nf or refit.device != dev):
        refit = None                                                  # P changed (densify / prune) or another device: nothing to keep
    nodes = torch.empty(nf, dtype=torch.float32, device=dev)         # binary nodes, 4-wide nodes, sorted leaf order (a refit writes a FRESH buffer:
                                                                     # an earlier forward whose backward is outstanding may still hold the old one)
    tb = lib.envgs_bvh_temp_bytes(P)
    temp = torch.empty(max(tb, 1), dtype=torch.uint8, device=dev)
    op = None if opacities is None else _f32c(opacities.detach()).reshape(-1)
    if op is not None and op.numel() != P:
        raise RuntimeError("opacities (%d) do not match the %d surfels of the vertex buffer" % (op.numel(), P))
    if refit is not None:
        _lib.check(lib.envgs_bvh_refit(P, _lib.ptr(v), _lib.ptr(op), _lib.ptr(refit), _lib.ptr(nodes), _lib.ptr(temp), tb, 1 if debug else 0, _stream(dev)),
                   "envgs_bvh_refit")
    else:
        _lib.check(lib.envgs_bvh_build(P, _lib.ptr(v), _lib.ptr(op), _lib.ptr(nodes), _lib.ptr(temp), tb, 1 if debug else 0, _stream(dev)), "envgs_bvh_build")
    LAST_STATS["bvh"] = "refit" if refit is not None else "build"
    return nodes, P


def _cfg(settings, P, R, shs, others, start_from_first, ray_shape, f16=False):
    from .raster import sh_degree_of
    deg = sh_degree_of(settings.sh_degree)
    rh, rw = (int(ray_shape[0]), int(ray_shape[1])) if len(ray_shape) == 2 else (0, 0)
    bg_len = min(int(settings.bg.numel()), 3)
    return _lib.TraceCfg(P, R, deg, 0 if shs is None else int(shs.shape[1]), int(settings.max_trace_depth),
                         (2 if start_from_first == 2 else (1 if start_from_first else 0)), 0 if others is None else 1, bg_len, 1 if settings.debug else 0,
                         rh, rw, float(settings.scale_modifier), float(settings.specular_threshold), 1 if f16 else 0)


NCOPY = 8                    # must equal NCOPY in csrc/trace_common.h
ROW_CAP = {}                 # tests only: ROW_CAP["force_per_ray"] pins the rows per ray of the compact per-hit buffers; COMPACT["on"] = False keeps the (R, cap) layouts
COMPACT = {"on": True}
HIT_CAP = {}                 # tests only: HIT_CAP["force"] pins the list capacity of every tracer (e.g. tiny, to exercise the overflow hand-off)
SORT_RAYS = {"on": True}     # coherence-sort the rays (direction, origin) before tracing
USE_RECORDS = {"on": True}   # atomic-free backward (one record per (batch, surfel) entry, grouped by surfel); False = cooperative atomic flush


class CapState:
    """Adaptive capacity of the per-ray hit lists, per TRACER INSTANCE (and device): 20 % above the longest list of that tracer's previous
    call, read through a pinned host mirror that was copied asynchronously at the end of that call -- no host sync on the hot path.  Two
    tracers in one process (camera-ray tracing over the base surfels plus the environment trace, gaussian2d_sampler.py:413-426 next to
    envgs_sampler.py:548) see very different list lengths; a module-global capacity would make each re-size the other's scratch."""

    def __init__(self, cap=512):
        self.cap = int(cap)
        self.found_per_ray = None   # hits found per ray in the previous call: sizes the COMPACT per-hit buffers (rows) of the next one
        self.hits_per_entry = None  # composited hits per (batch, surfel) entry of the previous call that prepared a backward: how coherent this tracer's batches are
        self.colour_only = False    # SurfelTracer.set_colour_only_backward: the backward will see the colour's gradient only -> store plane 0 alone
        self.defer_reduce = False   # SurfelTracer.set_deferred_surfel_gradients: the backward finishes the surfel gradients off the caller's stream
        self._mirrors = {}          # device -> dict(host, event, valid)

    def mirror(self, dev):
        m = self._mirrors.get(dev)
        if m is None:
            m = self._mirrors[dev] = dict(host=torch.zeros(72, dtype=torch.int32).pin_memory(), event=torch.cuda.Event(), valid=False, rays=1)
        return m

    def next_cap(self, dev):
        if HIT_CAP.get("force"):                       # tests: pin the capacity (e.g. tiny, to exercise the overflow hand-off)
            return int(HIT_CAP["force"])
        m = self.mirror(dev)
        if m["valid"] and m["event"].query():
            mx = int(m["host"][1])
            want = ((int(mx * 1.2) + 8 + 63) // 64) * 64          # 20 % headroom, multiple of 64 entries (512 B)
            if want > 256 and int(mx * 1.1) + 8 <= 256:
                # 256 is a class boundary: a capacity above it adds the long-list launch to every forward segment (~0.04 ms per step even when it finds
                # nothing to do: 512 workgroups scheduled onto a busy chip) -- not crossed for less than 10 % of headroom.  A ray that then finds more
                # than 256 hits takes the K-buffer kernels, as with any capacity
                want = 256
            self.cap = max(64, min(want, 1024))
            found = (int(m["host"][8]) & 0xFFFFFFFF) | ((int(m["host"][9]) & 0xFFFFFFFF) << 32)
            self.found_per_ray = found / max(1, m["rays"])
            hits = (int(m["host"][2]) & 0xFFFFFFFF) | ((int(m["host"][3]) & 0xFFFFFFFF) << 32)
            ent = (int(m["host"][64]) & 0xFFFFFFFF) + (int(m["host"][65]) & 0xFFFFFFFF)       # hits filed per hit + entries of the batch kernel
            if ent > 0:
                self.hits_per_entry = hits / ent
        return self.cap

    def pinned_word(self):
        """One pinned int32 of a ring of 32 (the record count of a forward is read at its backward: normally a few forwards are outstanding)
        plus a TOKEN that identifies the claim: a forward whose slot was handed out again before its backward ran (more than 32 grad-enabled
        traced calls in between: eval renders without no_grad, many-view accumulation with retained graphs) finds a different token in the
        slot and reads its count from the device instead (trace_backward)."""
        ring = self.__dict__.setdefault("_ring", [None, 0, [None] * 32])
        if ring[0] is None:
            ring[0] = torch.zeros(32, dtype=torch.int32).pin_memory()
        ring[1] = (ring[1] + 1) % 32
        token = object()
        ring[2][ring[1]] = token
        return ring[0][ring[1]:ring[1] + 1], (ring[2], ring[1], token)

    def next_rows(self, R, cap):
        """Rows of the compact per-hit buffers (hit_state / entries / pairs; include/envgs_trace.h: compact_rows) for a call with R rays:
        15 % above the previous call's hits found per ray (a first call assumes 192 per ray), never more than the (R, cap) layout would take.
        Rays that do not fit fall back to the K-buffer kernels -- slower, never wrong -- and the next call has the right size."""
        if ROW_CAP.get("force_per_ray") is not None:                   # tests: pin the rows per ray (tight: exercises the fall-back)
            per = float(ROW_CAP["force_per_ray"])
        else:
            per = 192.0 if self.found_per_ray is None else 1.15 * self.found_per_ray + 4.0
        return int(min(R * cap, max(int(per * R) + 4096, 4096)))

    def publish(self, counters, dev, rays=1):
        """Queue the asynchronous read-back of this call's longest list (counters[1]) and of its total of hits found (counters[8:10])."""
        m = self.mirror(dev)
        m["host"].copy_(counters[0:72], non_blocking=True)                 # (one copy of the head of the counter block: words 1, 2..3, 8..9, 64..65 are read)
        m["event"].record(torch.cuda.current_stream(dev)); m["valid"] = True; m["rays"] = int(rays)


_DEFAULT_CAPS = CapState()       # direct callers of trace_forward (tests, diagnostics); every SurfelTracer owns its own


def _scratch(shape, dtype, dev):
    """Large per-call scratch (hit lists, per-hit state, entries, pairs, gradient records).  Their sizes follow the ray count and the
    adaptive list capacity, both of which change from call to call (bounce stages, views); asking torch's caching allocator for a
    different multi-GB size every time makes it go back to hipMalloc (~0.5 s for tens of GB, measured) and pile up reserved memory.  Sizes
    above 32 MB are therefore rounded up to {1, 1.25, 1.5, 1.75} x 2^k elements -- a handful of distinct sizes that the allocator's cache
    serves from then on -- and the tensor is a view of the front of that block."""
    n = 1
    for d in shape:
        n *= int(d)
    item = torch.empty(0, dtype=dtype).element_size()
    if n * item <= (32 << 20):
        return torch.empty(shape, dtype=dtype, device=dev)
    k = max(n - 1, 1).bit_length() - 1                   # 2^k < n <= 2^(k+1)
    for q in (5, 6, 7, 8):
        if (q << k) >> 2 >= n:
            return torch.empty((q << k) >> 2, dtype=dtype, device=dev)[:n].view(shape)
    raise AssertionError


def _carve_i32(dev, sizes):
    """One int32 allocation carved into named views (each starting on a 256 B boundary)."""
    offs, total = {}, 0
    for k, n in sizes.items():
        offs[k] = total
        total += ((int(n) + 63) // 64) * 64
    buf = torch.empty(max(total, 64), dtype=torch.int32, device=dev)
    out = {k: buf[o:o + int(sizes[k])] for k, o in offs.items()}
    out["_i32_block"] = buf
    return out


# record backward: entries with few hits filed per hit (envgs_trace.h: sparse_hits).  "auto": when this tracer state's previous call averaged fewer than
# `below` composited hits per entry; "on" / "off": tests
SPARSE = {"mode": "auto", "below": 6.0}
QUAD_SH = {"on": True}       # list path: four lanes share the fetch of a surfel's SH block (tests switch it off to cover the per-lane gathers)
KEEP_LISTS = {"on": False}   # tests: keep the last forward's per-ray hit lists reachable through last_hit_lists()


_DEFERRED = {"pending": False, "keep": None, "dev": None}


def join_deferred_gradients():
    """OPTIONAL, not part of the reference interface (include/envgs_trace.h: defer_reduce / envgs_trace_backward_join).  After a backward of a tracer
    with set_deferred_surfel_gradients(True), the gradients of the SURFEL parameters are still being finished on a stream of the library's own;
    this makes the current stream wait for them.  No-op when nothing is pending.  envgs_amd.optim.FusedAdam.step, GradExchange and every later
    traced call do it themselves; any other consumer of those gradients (a torch optimizer, a hook, .grad arithmetic) must call it first."""
    if not _DEFERRED["pending"]:
        return
    dev = _DEFERRED["dev"]
    with torch.cuda.device(dev):
        _lib.check(_lib.load().envgs_trace_backward_join(_stream(dev)), "envgs_trace_backward_join")
    _DEFERRED.update(pending=False, keep=None, dev=None)


class _DeferBarrier(torch.autograd.Function):
    """Identity in the forward; its backward joins the deferred surfel gradients before handing them on."""

    @staticmethod
    def forward(ctx, *ts):
        return tuple(t.view_as(t) for t in ts)

    @staticmethod
    def backward(ctx, *gs):
        join_deferred_gradients()
        return gs


def defer_barrier(*tensors):
    """OPTIONAL, for a training loop whose tracer inputs are NOT leaves (EasyVolcap feeds the tracer sigmoid / exp / normalize of its raw parameters)
    and that wants set_deferred_surfel_gradients all the same:

        act = [torch.sigmoid(raw_opacity), torch.exp(raw_scale), ...]          # at the START of the step, before the base pass
        opacities, scales, ... = envgs_amd.tracing.defer_barrier(*act)         # identity; pass THESE (and only to) the tracer

    The tracer's backward returns its (still unfinished) surfel gradients into this node, whose backward makes the current stream wait for them and
    only then hands them to the activations' backward.  Autograd runs ready nodes latest-created first, so a barrier created BEFORE the base pass
    comes up after the base pass's backward has been queued -- which is the overlap the deferral is for; created later it is merely correct.
    The returned tensors must feed the tracer and nothing else (a second consumer makes autograd ADD the two gradients on arrival, before the
    join), and each parameter must still receive its tracer gradient from one traced call or bounce chain per backward pass."""
    outs = _DeferBarrier.apply(*tensors)
    return outs if len(tensors) != 1 else outs[0]


def _behind_barrier(t):
    fn = t.grad_fn
    return fn is not None and type(fn).__name__ == "_DeferBarrierBackward" and not getattr(t, "_backward_hooks", None)


def trace_forward(nodes, ray_o, ray_d, means3D, shs, colors_precomp, others_precomp, opacities, scales, rotations, settings,
                  start_from_first, use_lists=True, need_grad=True, caps=None):
    lib = _lib.load()
    join_deferred_gradients()                  # (the deferred tail of the previous backward reads scratch this call rewrites)
    caps = _DEFAULT_CAPS if caps is None else caps
    dev = means3D.device
    lead = tuple(ray_o.shape[:-1])
    ro = _f32c(ray_o).reshape(-1, 3); rd = _f32c(ray_d).reshape(-1, 3)
    R, P = ro.shape[0], means3D.shape[0]
    means3D = _f32c(means3D); opacities = _f32c(opacities); scales = _f32c(scales); rotations = _f32c(rotations)
    from .raster import _featc
    shs = _featc(shs); colors_precomp = _featc(colors_precomp); others_precomp = _f32c(others_precomp)       # features may stay in fp16 storage
    f16 = any(t is not None and t.dtype == torch.float16 for t in (shs, colors_precomp))
    bg = _f32c(settings.bg).reshape(-1).to(dev)
    cfg = _cfg(settings, P, R, shs, others_precomp, start_from_first, lead, f16)
    ND = cfg.max_trace_depth + 1
    f32 = dict(dtype=torch.float32, device=dev)
    srec = torch.empty(max(P, 1), 16, **f32)
    counters = torch.empty(96, dtype=torch.int32, device=dev)
    rgb = torch.empty(R, 3, **f32); dpt = torch.empty(R, 1, **f32); acc = torch.empty(R, 1, **f32)
    norm = torch.empty(R, 3, **f32); dist = torch.empty(R, 1, **f32); aux = torch.empty(R, 2, **f32)
    mid = torch.empty(R, 16 * ND, **f32); wet = torch.empty(P, 1, **f32); final_T = torch.empty(R, **f32)
    cap = caps.next_cap(dev) if (use_lists and ND == 1 and P > 0 and R > 0) else 0
    rows = 0
    lists = None
    keep = {}
    if cap:
        i32 = dict(dtype=torch.int32, device=dev)
        sb = lib.envgs_raster_scan_temp_bytes(NCOPY * P)
        rb = lib.envgs_trace_ray_sort_temp_bytes(R)
        nbatch = (R + 63) // 64
        # the small int32 scratch of a call comes out of ONE allocation (a dozen torch.empty calls are ~0.1 ms of host time, and this code runs
        # right behind the rasterizer's one host sync, where the GPU has nothing queued)
        keep = _carve_i32(dev, dict(hit_cnt=R, n_used=R, spill=lib.envgs_trace_stack_spill_ints(R), surf_acc=2 * P * NCOPY, surf_cnt=P * NCOPY,
                                    surf_off=P * NCOPY, scan_temp=(max(sb, 1) + 3) // 4, ray_keys=2 * R, ray_order=2 * R, ray_sort_temp=(max(rb, 1) + 3) // 4,
                                    n_entries=2 * nbatch, row_off=R, batch_rows=2 * nbatch, row_blk=nbatch + 16))
        keep["n_entries"] = keep["n_entries"].view(nbatch, 2)
        keep["hit_lists"] = _scratch((R, cap, 2), torch.int32, dev)
        srt = SORT_RAYS["on"]
        if shs is not None and shs.shape[1] == 16 and QUAD_SH["on"]:
            keep["sh_perm"] = torch.empty(P, 48, dtype=shs.dtype, device=dev)      # quad-permuted SH copy (envgs_trace.h: sh_perm)
        if need_grad and USE_RECORDS["on"]:
            # what the record backward needs from the forward: per-hit state, and the (batch, surfel) entries with their (lane, k) pairs.
            # COMPACT: rows follow the hits the rays actually have (a prefix sum of the hit counts, taken on the device between the collection
            # and the sort) instead of rays x capacity -- a ray uses a third of its capacity (42 -> 19 GB for a 1.92 M-ray stage)
            colour_only = bool(getattr(caps, "colour_only", False))
            sw = 4 if colour_only else (10 if others_precomp is not None else 8)     # floats per row: two 16 B planes (+ one 8 B plane with `others`); colour only: plane 0
            if COMPACT["on"] and srt:
                rows = caps.next_rows(R, cap)
                keep.update(hit_state=_scratch((rows, sw), torch.float32, dev), entries=_scratch((rows,), torch.int64, dev), pairs=_scratch((rows,), torch.int32, dev))
            else:
                keep.update(hit_state=_scratch((R, cap, sw), torch.float32, dev), entries=_scratch((nbatch, 64 * cap), torch.int64, dev),
                            pairs=_scratch((nbatch, 64 * cap), torch.int32, dev))
        sparse = SPARSE["mode"] == "on" or (SPARSE["mode"] == "auto" and caps.hits_per_entry is not None and caps.hits_per_entry < SPARSE["below"])
        if "hit_state" in keep and sparse:
            # sparse entries (envgs_trace.h: sparse_hits): the previous call of this tracer state was INCOHERENT (few hits per (batch, surfel) entry:
            # bounce rays off rough geometry) -- entries of at most four hits are filed per hit; room for every row (most hits of such a call are
            # filed; what finds no room takes the batch kernel -- slower, never wrong)
            n_rows = rows if rows else R * cap
            keep["sparse_hits"] = _scratch((max(4096, n_rows), 4), torch.int32, dev)
        lists = _lib.TraceLists(keep["hit_lists"].data_ptr(), keep["hit_cnt"].data_ptr(), keep["n_used"].data_ptr(), cap,
                                keep["spill"].data_ptr(), keep["surf_acc"].data_ptr(), keep["surf_cnt"].data_ptr(), keep["surf_off"].data_ptr(),
                                keep["scan_temp"].data_ptr(), sb, keep["ray_keys"].data_ptr() if srt else None,
                                keep["ray_order"].data_ptr() if srt else None, keep["ray_sort_temp"].data_ptr() if srt else None, rb, None, 0,
                                *[(keep[k].data_ptr() if k in keep else None) for k in ("hit_state", "entries", "pairs")],
                                keep["n_entries"].data_ptr() if "hit_state" in keep else None, rows,
                                *[(keep[k].data_ptr() if (k in keep and rows) else None) for k in ("row_off", "batch_rows", "row_blk")],
                                keep["sh_perm"].data_ptr() if "sh_perm" in keep else None, 0,
                                keep["sparse_hits"].data_ptr() if "sparse_hits" in keep else None, keep["sparse_hits"].shape[0] if "sparse_hits" in keep else 0)
        if "hit_state" in keep and getattr(caps, "colour_only", False):
            lists.state_planes = 1
            keep["colour_only"] = True
        if "hit_state" in keep and getattr(caps, "defer_reduce", False):
            keep["defer_reduce"] = True
    p = _lib.ptr
    _lib.check(lib.envgs_trace_forward(cfg, p(nodes), p(ro), p(rd), p(means3D), p(scales), p(rotations), p(opacities), p(shs),
                                       p(colors_precomp), p(others_precomp), p(bg), p(srec), p(counters), p(rgb), p(dpt), p(acc),
                                       p(norm), p(dist), p(aux), p(mid), p(wet), p(final_T), lists, _stream(dev)), "envgs_trace_forward")
    LAST_STATS.update(P=P, R=R, caps=caps, rows=rows, counters=counters, n_entries=(keep.get("n_entries") if "hit_state" in keep else None), cap=cap,
                      lists=((keep["hit_lists"], keep["n_used"], keep["hit_cnt"]) if (cap and KEEP_LISTS["on"]) else None))
    if cap:
        if "hit_state" in keep and not KEEP_LISTS["on"]:
            # the record backward reads the hit counts, the per-hit state and the entries / pairs -- not the lists themselves (rays x capacity
            # x 8 B: the largest buffer of the call, 4.9 GB per stage of the 1200x1600 two-bounce configuration): released here, so that the
            # next stage / the next call re-uses the block instead of three stages holding one each until their backward
            keep.pop("hit_lists")
            lists.hit_lists = None
        # asynchronous read-backs for later: the longest list (sizes the next call's cap) and the number of gradient records
        caps.publish(counters, dev, rays=R)
        if "hit_state" in keep:                                   # (a pinned word from the tracer's ring: pinning host memory per call is ~0.1 ms)
            keep["n_rec_host"], keep["n_rec_claim"] = caps.pinned_word()
            keep["n_rec_host"].copy_(keep["surf_off"].view(-1)[NCOPY * P - 1:NCOPY * P], non_blocking=True)
            keep["n_rec_event"] = torch.cuda.Event(); keep["n_rec_event"].record(torch.cuda.current_stream(dev))
    saved = dict(cfg=cfg, nodes=nodes, ro=ro, rd=rd, means3D=means3D, scales=scales, rotations=rotations, opacities=opacities,
                 shs=shs, colors_precomp=colors_precomp, others=others_precomp, bg=bg, srec=srec, counters=counters,
                 rgb=(rgb if ND == 1 else mid[:, 13:16].contiguous()),      # (C-ABI in-kernel bounces, forward use only: stage 0's own colour)
                 dpt=dpt, acc=acc, norm=norm, aux=aux, final_T=final_T, lead=lead, lists=lists, keep=keep, cap=cap)
    outs = (rgb.reshape(lead + (3,)), dpt.reshape(lead + (1,)), acc.reshape(lead + (1,)), norm.reshape(lead + (3,)),
            dist.reshape(lead + (1,)), aux.reshape(lead + (2,)), mid.reshape(lead + (16 * ND,)), wet)
    return outs, saved


class _BounceChain:
    """The traced calls of ONE multi-bounce forward (SurfelTracer._forward_bounces) differentiate the same surfels.  With deferred surfel
    gradients their backwards share the accumulators (include/envgs_trace.h: ENVGS_TRACE_ACCUMULATE / _NO_FINISH): the first backward to run
    allocates and zeroes them, stage 0 -- the last: every later stage's rays hang off its outputs -- converts them and returns the gradients;
    the stages in between return none for the surfel parameters."""

    def __init__(self):
        self.acc = None
        self.plain = False                               # a stage of the chain ran the stream-ordered backward (no record path): all of them must


def trace_backward(saved, g_rgb, g_dpt, g_acc, g_norm, g_aux, chain=None, allow_defer=True):
    """chain = (a _BounceChain, this call's stage index) for the stages of a multi-bounce forward whose tracer defers its surfel gradients;
    allow_defer = False: the autograd node found a surfel input that is not a plain leaf without a gradient and without hooks -- stream-ordered then."""
    lib = _lib.load()
    cfg = saved["cfg"]
    P, R = cfg.P, cfg.num_rays
    dev = saved["ro"].device
    f32 = dict(dtype=torch.float32, device=dev)
    z = lambda g, c: None if g is None else _f32c(g).reshape(R, c)          # an output the loss does not use: NULL = zero upstream gradient
    g_rgb, g_dpt, g_acc, g_norm, g_aux = z(g_rgb, 3), z(g_dpt, 1), z(g_acc, 1), z(g_norm, 3), z(g_aux, 2)
    shs, others = saved["shs"], saved["others"]
    p = _lib.ptr
    s = saved
    lists = s["lists"]
    defer = bool(allow_defer and lists is not None and USE_RECORDS["on"] and "hit_state" in s["keep"] and s["keep"].get("defer_reduce"))
    chained = defer and chain is not None
    if chain is not None:
        if (not defer and chain[0].acc is not None) or (defer and chain[0].plain):
            raise RuntimeError("SurfelTracer: the stages of one multi-bounce call disagree about the record backward (deferred surfel gradients need it in every stage)")
        chain[0].plain = not defer
    first = not chained or chain[0].acc is None          # (of the chain: zeroes the accumulators)
    last = not chained or chain[1] == 0                  # converts them
    if first:
        join_deferred_gradients()
    if chained and not first:
        geo_rec, dshs, dcolors = chain[0].acc
    else:
        geo_rec = torch.empty(max(P, 1), 16, **f32)
        dshs = torch.empty(shs.shape, **f32) if shs is not None else None            # gradients are fp32 whatever the feature storage
        dcolors = torch.empty(P, 3, **f32) if shs is None else None
        if chained:
            chain[0].acc = (geo_rec, dshs, dcolors)
    if chained and last:
        chain[0].acc = None                              # (a second backward pass through a retained graph starts over)
    dmeans = dgrads3D = dscales = drots = dopac = None
    if last:
        dmeans = torch.empty(P, 3, **f32); dgrads3D = torch.empty(P, 3, **f32); dscales = torch.empty(P, 2, **f32)
        drots = torch.empty(P, 4, **f32); dopac = torch.empty(P, 1, **f32)
    dothers = torch.empty(P, 2, **f32) if others is not None else None
    dro = torch.empty(R, 3, **f32); drd = torch.empty(R, 3, **f32)
    records = None
    if lists is not None and s["keep"].get("colour_only") and any(g_ is not None for g_ in (g_dpt, g_acc, g_norm, g_aux)):
        raise RuntimeError("SurfelTracer: set_colour_only_backward(True) promised that only the colour output would be differentiated, but a gradient "
                           "arrived for dpt / acc / norm / aux -- the forward kept the colour's per-hit state only; switch the promise off")
    if lists is not None and USE_RECORDS["on"] and "hit_state" in s["keep"]:
        # atomic-free backward: one 256 B record per (batch, surfel) entry, grouped by surfel.  The count is known on the device
        # (inclusive scan of the per-surfel entry counts, done at the end of the forward); reading it is the one host sync here.
        s["keep"]["n_rec_event"].synchronize()         # copied at the end of the forward; long since complete
        slots, slot, token = s["keep"]["n_rec_claim"]
        if slots[slot] is token:
            n_rec = int(s["keep"]["n_rec_host"][0]) & 0xFFFFFFFF if P > 0 else 0
        else:                                           # the pinned slot was re-used by a later forward: synchronous read of this forward's own scan
            n_rec = int(s["keep"]["surf_off"].view(-1)[NCOPY * P - 1].item()) & 0xFFFFFFFF if P > 0 else 0
        if n_rec > 0:
            records = _scratch((n_rec, 64), torch.float32, dev)
            lists.records = records.data_ptr()
            lists.num_records = n_rec
    if lists is not None:
        lists.defer_reduce = (1 if defer else 0) | (0 if first else 2) | (0 if last else 4)
    _lib.check(lib.envgs_trace_backward(cfg, p(s["nodes"]), p(s["ro"]), p(s["rd"]), p(s["means3D"]), p(s["scales"]), p(s["rotations"]),
                                        p(s["opacities"]), p(shs), p(s["colors_precomp"]), p(others), p(s["bg"]), p(s["srec"]),
                                        p(s["counters"]), p(s["rgb"]), p(s["dpt"]), p(s["acc"]), p(s["norm"]), p(s["aux"]), p(s["final_T"]),
                                        p(g_rgb), p(g_dpt), p(g_acc), p(g_norm), p(g_aux), p(geo_rec), p(dmeans), p(dgrads3D), p(dscales),
                                        p(drots), p(dopac), p(dshs), p(dcolors), p(dothers), p(dro), p(drd), lists, _stream(dev)),
               "envgs_trace_backward")
    if defer:
        # the tail (record sums -> surfel gradients) runs on the library's stream: what it reads and writes stays referenced until someone joins
        # (the forward's scratch through `saved`: autograd drops its own reference as soon as this node returns) -- the gradients through their
        # STORAGES, so that autograd still sees the returned tensors as unshared and moves them into .grad instead of copying them
        held = [saved] + [t.untyped_storage() for t in (geo_rec, records, dmeans, dgrads3D, dscales, drots, dopac, dshs, dcolors) if t is not None]
        _DEFERRED.update(pending=True, dev=dev, keep=(_DEFERRED["keep"] or []) + held)
    lead = s["lead"]
    if not last:
        dshs = dcolors = None                            # (the accumulators: stage 0 returns them)
    return dict(ray_o=dro.reshape(lead + (3,)), ray_d=drd.reshape(lead + (3,)), means3D=dmeans, grads3D=dgrads3D, shs=dshs,
                colors_precomp=dcolors, others_precomp=dothers, opacities=dopac, scales=dscales, rotations=drots)


def frame_from_transmat(cov3D_precomp, settings):
    """Invert the python transMat of optix_utils.py:143-154 / gaussian2d_utils.py:1050-1061:

        T = (splat2world[:, [0,1,3]] @ world2pix[:, [0,1,3]]).permute(0,2,1).reshape(-1,9),   world2pix = full_proj @ ndc2pix

    Row r of splat2world[:, [0,1,3]] is (s_u a, 0), (s_v b, 0), (p, 1), so with PM = world2pix[:, [0,1,3]] (4x3) and PM3 = PM[:3]:
    s_u a = M_u PM3^-1, s_v b = M_v PM3^-1 where M = T.reshape(3,3)^T.  Returns (scales (P,2) with scale_modifier divided out,
    rotations (P,4) unit quaternions (r,x,y,z) of the frame [a, b, a x b]); plain torch expressions, so autograd carries the tracer's
    gradients back to cov3D_precomp."""
    T = cov3D_precomp
    if T.dtype != torch.float32:
        T = T.float()
    dev = T.device
    W, H = float(settings.image_width), float(settings.image_height)
    fp = settings.projmatrix.to(dev).float()
    PM = torch.stack([0.5 * W * fp[:, 0] + 0.5 * (W - 1.0) * fp[:, 3], 0.5 * H * fp[:, 1] + 0.5 * (H - 1.0) * fp[:, 3], fp[:, 3]], dim=1)   # (4,3)
    inv = torch.linalg.inv(PM[:3].double()).float()                     # (3,3)
    M = T.reshape(-1, 3, 3).transpose(1, 2)                              # rows: u, v, 1 ; columns: x*w, y*w, w
    A = M[:, 0] @ inv; B = M[:, 1] @ inv                                # s_u a, s_v b   (P,3)
    su = A.norm(dim=-1, keepdim=True); sv = B.norm(dim=-1, keepdim=True)
    a = A / su; b = B / sv
    n = torch.cross(a, b, dim=-1)
    # rotation matrix with columns a, b, n -> quaternion; all four candidate forms, the numerically largest one selected
    m00, m10, m20 = a[:, 0], a[:, 1], a[:, 2]
    m01, m11, m21 = b[:, 0], b[:, 1], b[:, 2]
    m02, m12, m22 = n[:, 0], n[:, 1], n[:, 2]
    q_abs = torch.sqrt(torch.clamp_min(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1), 1e-12))
    cand = torch.stack([
        torch.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], dim=-1)], dim=1)          # (P,4 candidates,4)
    cand = cand / (2.0 * q_abs[..., None])
    best = q_abs.detach().argmax(dim=-1)
    q = cand[torch.arange(cand.shape[0], device=dev), best]
    q = q / q.norm(dim=-1, keepdim=True)
    mod = float(settings.scale_modifier)
    return torch.cat([su, sv], dim=-1) / mod, q


_BUILD_STREAMS = {}
_SCENE_EPOCH = [0]


def invalidate_all_structures():
    """OPTIONAL: every SurfelTracer of the process answers its next `rebuild=True` request with a FULL build -- called by the code that knows the
    surfel set changed wholesale while keeping its size (envgs_amd.densify.SurfelSet: densify / prune / opacity reset), so that the adaptive
    build-or-refit policy never serves such a step from a stale topology (the quality guard only sees it one trace later)."""
    _SCENE_EPOCH[0] += 1



def _build_stream(dev):
    s = _BUILD_STREAMS.get(dev)
    if s is None:
        s = _BUILD_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


class _TraceSurfels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray_o, ray_d, v, means3D, grads3D, shs, colors_precomp, others_precomp, opacities, scales, rotations,
                cov3D_precomp, tracer_settings, start_from_first, nodes, caps=None, chain=None):
        ctx.set_materialize_grads(False)          # outputs the loss does not use arrive as None (= NULL upstream pointer), not as buffers of zeros
        none = lambda t: None if (t is None or t.numel() == 0) else t
        from .raster import _store                       # fp16 feature storage selected (envgs_amd.set_feature_storage): half copies made HERE, fp32 gradients
        outs, saved = trace_forward(nodes, ray_o, ray_d, means3D, _store(none(shs)), _store(none(colors_precomp)), none(others_precomp), opacities,
                                    scales, rotations, tracer_settings, start_from_first, need_grad=any(ctx.needs_input_grad), caps=caps)
        ctx.saved = saved
        ctx.chain = chain
        # deferred surfel gradients (set_deferred_surfel_gradients) are only safe when autograd does nothing with them but MOVE them into .grad: every
        # differentiated surfel input must be a leaf (a non-leaf's gradient is fed to the next backward node on the current stream at once); whether
        # it has a .grad to add into, or hooks that would read it, is looked at when the backward runs
        ctx.surfel_inputs = [t for t in (means3D, grads3D, shs, colors_precomp, opacities, scales, rotations) if t is not None and t.requires_grad and t.is_leaf]
        ctx.surfel_leaves = all(t.is_leaf or _behind_barrier(t) for t in (means3D, grads3D, shs, colors_precomp, opacities, scales, rotations)
                                if t is not None and t.requires_grad)           # (or the output of defer_barrier: that node joins before anyone else sees the gradient)
        ctx.in_dtypes = tuple(None if t is None else t.dtype for t in (ray_o, ray_d, means3D, grads3D, shs, colors_precomp,
                                                                        others_precomp, opacities, scales, rotations))
        rgb, dpt, acc, norm, dist, aux, mid, wet = outs
        ctx.mark_non_differentiable(dist, mid, wet)
        return rgb, dpt, acc, norm, dist, aux, mid, wet

    @staticmethod
    def backward(ctx, g_rgb, g_dpt, g_acc, g_norm, g_dist, g_aux, g_mid, g_wet):
        quiet = lambda t: t.grad is None and not getattr(t, "_backward_hooks", None) and not getattr(t, "_post_accumulate_grad_hooks", None)
        g = trace_backward(ctx.saved, g_rgb, g_dpt, g_acc, g_norm, g_aux, chain=ctx.chain,
                           allow_defer=bool(ctx.surfel_leaves and all(quiet(t) for t in ctx.surfel_inputs)))
        order = ("ray_o", "ray_d", "means3D", "grads3D", "shs", "colors_precomp", "others_precomp", "opacities", "scales", "rotations")
        vals = [None if (g[k] is None or dt is None) else g[k].to(dt) for k, dt in zip(order, ctx.in_dtypes)]
        ro, rd, m3, g3, sh, col, oth, op, sc, rot = vals
        return ro, rd, None, m3, g3, sh, col, oth, op, sc, rot, None, None, None, None, None, None


class SurfelTracer(nn.Module):
    def __init__(self):
        super().__init__()
        _lib.load()                       # fail at construction (optix_utils.py:24 creates the OptiX context here)
        self.nodes = None
        self.num_surfels = 0
        self._pending = None
        self._keep = None                 # the structure a refit request (rebuild=False) updates in place
        self.caps = CapState()            # this tracer's adaptive hit-list capacity (not shared with other tracers of the process)
        # build-or-refit policy for rebuild=True requests (set_structure_policy): the reference's caller asks for a rebuild on EVERY training step
        # (optix_utils.py:73-78).  A refit is exact, so such a request is answered with one while the tree is young and has not degraded.
        self._policy = dict(mode="adaptive", max_age=16, max_growth=1.25)
        self._age = 0                     # structures derived from the last full build by refits
        self._q = None                    # quality read-backs: dict(host=pinned (2,2) floats, build=(event, row), last=(event, row))

    def set_structure_policy(self, mode=None, max_age=None, max_growth=None):
        """OPTIONAL, not part of the reference interface.  How a `rebuild=True` request is served when the surfel count is unchanged:
        "adaptive" (default): by a REFIT of the existing topology (envgs_bvh_refit: 6 launches instead of 15; hit sets identical to a fresh build's)
                     unless `max_age` structures have been derived from the last full build, or the tree's surface-area cost measured on the
                     device after the previous refit (envgs_bvh_quality, read back asynchronously -- never a host sync) has grown by more than
                     `max_growth` x since that build (surfels moved far: opacity reset, a prune + densify that kept the count, a scene change);
        "rebuild":   always by a full build -- the literal behaviour of the reference's OptiX GAS.
        `rebuild=False` requests are always refits (OptiX's update), a changed surfel count always a full build.
        Only the arguments that are given change (None = keep the current value; defaults: "adaptive", 16, 1.25) -- a caller that tunes one
        field does not reset the others."""
        if mode is not None and mode not in ("adaptive", "rebuild"):
            raise ValueError("structure policy must be 'adaptive' or 'rebuild', got %r" % (mode,))
        if mode is not None: self._policy["mode"] = mode
        if max_age is not None: self._policy["max_age"] = int(max_age)
        if max_growth is not None: self._policy["max_growth"] = float(max_growth)

    def invalidate_structure(self):
        """OPTIONAL: forget the existing structure (the next request is a full build whatever the policy) -- for a caller that knows the surfel
        set was re-indexed or moved wholesale without changing its size (prune + densify, opacity reset).  envgs_amd.densify.SurfelSet says so
        for every tracer of the process through `invalidate_all_structures()`; the unchanged EasyVolcap GaussianModel does not, and pays ONE
        badly fitted refit before the quality guard rebuilds."""
        self._keep = None
        self.nodes = None
        self._age = 0
        self._epoch = _SCENE_EPOCH[0]

    def _refit_allowed(self):
        pol = self._policy
        if pol["mode"] != "adaptive" or self._age >= pol["max_age"]:
            return False
        q = self._q
        if q is None or q.get("build") is None:
            return True                                   # (nothing measured yet: the age bound alone)
        ev_b, ev_l = q["build"], q.get("last")
        if ev_l is None or not ev_b.query() or not ev_l.query():
            return True                                   # the read-back of the previous structure has not landed: decide on age, never wait
        h = q["host"]
        b = float(h[0, 0]) / max(float(h[0, 1]), 1e-30)
        l = float(h[1, 0]) / max(float(h[1, 1]), 1e-30)
        LAST_STATS["bvh_growth"] = l / max(b, 1e-30)
        return l <= pol["max_growth"] * b

    def _measure(self, nodes, refit):
        """Queue the quality measurement of the structure just built (row 0) or refitted (row 1) and its read-back into pinned memory."""
        lib = _lib.load()
        dev = nodes.device
        q = self._q
        if q is None or q["dev"].device != dev:
            q = self._q = dict(host=torch.zeros(2, 2, dtype=torch.float32).pin_memory(), dev=torch.zeros(2, 2, dtype=torch.float32, device=dev), build=None, last=None)
        row = 1 if refit else 0
        _lib.check(lib.envgs_bvh_quality(self.num_surfels, _lib.ptr(nodes), _lib.ptr(q["dev"][row]), _stream(dev)), "envgs_bvh_quality")
        q["host"][row].copy_(q["dev"][row], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(dev))
        if refit:
            q["last"] = ev
        else:
            q["build"], q["last"] = ev, None

    def build_acceleration_structure(self, vertices, faces=None, rebuild=True):
        """optix_utils.py:78.  `faces` must be the get_disks layout (2 triangles per 4 consecutive vertices).
        The LBVH itself is built lazily by the next traced call, which knows the opacities and can bound every surfel by
        the region where it can still contribute (tighter boxes, identical results)."""
        if faces is not None and faces.shape[0] * 2 != vertices.shape[0]:
            raise RuntimeError("faces (%d,3) do not match vertices (%d,3): expected 2 triangles per 4 vertices" % (faces.shape[0], vertices.shape[0]))
        if vertices.dim() != 2 or vertices.shape[1] != 3 or vertices.shape[0] % 4 != 0:
            raise RuntimeError("vertices must be (4P,3) in the get_disks layout, got %s" % (tuple(vertices.shape),))
        if vertices.device.type != "cuda":
            raise RuntimeError("envgs_amd tracer needs tensors on the GPU (got %s); there is no CPU path" % vertices.device)
        # rebuild=False is OptiX's "update": the topology of the existing structure is kept and its boxes are refitted to the new vertices
        # (envgs_bvh_refit) -- exact, the hit sets are those of a fresh build.  Nothing to update (first call, or P changed): a full build.
        # The reference itself only ever passes rebuild=True (optix_utils.py:78), which rebuilds, like its OptiX GAS.
        ev = self.__dict__.pop("_build_event", None)
        if ev is not None and self.nodes is not None:
            torch.cuda.current_stream(self.nodes.device).wait_event(ev)       # (a build started by prepare() that no trace has consumed)
        if self.__dict__.get("_epoch", 0) != _SCENE_EPOCH[0]:      # invalidate_all_structures() since this tracer's last build: topology is stale
            self.invalidate_structure()
        have = self.nodes if self.nodes is not None else self._keep
        same = have is not None and vertices.shape[0] // 4 == self.num_surfels
        self._keep = have if (same and (not rebuild or self._refit_allowed())) else None
        self._pending = vertices.detach()
        self.nodes = None
        self.num_surfels = vertices.shape[0] // 4

    def _built(self, was_refit):
        self._age = self._age + 1 if was_refit else 0
        if self._policy["mode"] == "adaptive" and self.num_surfels > 1:
            self._measure(self.nodes, was_refit)

    def set_deferred_surfel_gradients(self, on=True):
        """OPTIONAL, not part of the reference interface (include/envgs_trace.h: defer_reduce).  The backward of this tracer's bounce-free calls
        returns as soon as the RAY gradients are complete on the current stream; the gradients of the surfel parameters are finished on a stream
        of the library's own, beside whatever the caller runs next (the base pass's backward: 0.2 ms of the EnvGS step).  That is only sound when
        autograd does nothing with those gradients but move them into `.grad`, which the autograd node CHECKS per call -- every differentiated
        surfel input a leaf, without a `.grad` to add into and without tensor / post-accumulate hooks; anything else (activated parameters as in
        the unchanged EasyVolcap caller, gradient accumulation over several backward passes, GradExchange's flat `.grad` views) takes the
        stream-ordered backward -- and when each parameter receives its gradient from ONE traced call (or bounce chain) per backward pass.  What
        the caller still promises: to call envgs_amd.tracing.join_deferred_gradients() before anything reads the gradients -- FusedAdam.step and
        the next traced call do."""
        self.caps.defer_reduce = bool(on)

    def set_colour_only_backward(self, on=True):
        """OPTIONAL, not part of the reference interface (include/envgs_trace.h: state_planes): a promise that the backward of this tracer's
        bounce-free calls will receive a gradient for the COLOUR output only -- the EnvGS training step (the env pass's depth / accumulation /
        normal maps are not supervised).  The forward then stores a quarter to a half of the per-hit state.  A gradient for another output
        raises instead of being dropped."""
        self.caps.colour_only = bool(on)

    def prepare(self, opacities=None):
        """OPTIONAL, not part of the reference interface: start the structure build requested by build_acceleration_structure() NOW, on a side
        stream, instead of inside the next traced call.  A caller that knows the environment set before it renders the base pass (the fused
        caller of envgs_amd/envgs_step.py) overlaps the 0.2 ms build with the rasterizer's forward; the traced call waits for it.  The
        unchanged EasyVolcap caller never calls this and builds at trace time, as before."""
        if self._pending is None or self.nodes is not None:
            return
        dev = self._pending.device
        side = _build_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))           # the vertex buffer is produced on the caller's stream
        self._pending.record_stream(side)
        if opacities is not None:
            opacities = opacities.detach()
            opacities.record_stream(side)
        if self._keep is not None:
            self._keep.record_stream(side)
        with torch.cuda.stream(side):
            self.nodes, self.num_surfels = build_bvh(self._pending, opacities, refit=self._keep)
            self._built(LAST_STATS["bvh"] == "refit")
        self._keep = None
        self._build_event = torch.cuda.Event()
        self._build_event.record(side)
        self._pending = None

    def forward(self, ray_o, ray_d, v=None, *, means3D, grads3D=None, shs=None, colors_precomp=None, others_precomp=None,
                opacities=None, scales=None, rotations=None, cov3D_precomp=None, tracer_settings=None, start_from_first=True):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if cov3D_precomp is not None:
            # pipe.compute_cov3D_python (optix_utils.py:143-154): the caller hands over the screen-space transMat.  The tracer intersects
            # surfels in WORLD space, so the tangent frame is recovered from it (differentiably: the gradient reaches cov3D_precomp)
            scales, rotations = frame_from_transmat(cov3D_precomp, tracer_settings)
        if self.nodes is None and self._pending is None:
            if v is None:
                raise RuntimeError("SurfelTracer: no acceleration structure; call build_acceleration_structure first")
            self.build_acceleration_structure(v)
        if means3D.shape[0] != self.num_surfels:
            raise RuntimeError("SurfelTracer: acceleration structure holds %d surfels, call has %d (rebuild after densification)"
                               % (self.num_surfels, means3D.shape[0]))
        if self.nodes is None:
            if self._pending.shape[0] != 4 * means3D.shape[0]:
                raise RuntimeError("SurfelTracer: acceleration structure was requested for %d surfels, call has %d" % (self._pending.shape[0] // 4, means3D.shape[0]))
            self.nodes, self.num_surfels = build_bvh(self._pending, opacities, refit=self._keep)
            self._built(LAST_STATS["bvh"] == "refit")
            self._pending = None
            self._keep = None
        ev = self.__dict__.pop("_build_event", None)
        if ev is not None:                                             # built ahead on the side stream (prepare()): order this stream behind it
            cur = torch.cuda.current_stream(self.nodes.device)
            cur.wait_event(ev)
            self.nodes.record_stream(cur)
        if grads3D is None:
            grads3D = torch.zeros_like(means3D)
        e = torch.Tensor([])
        args = (v, means3D, grads3D, e if shs is None else shs, e if colors_precomp is None else colors_precomp,
                e if others_precomp is None else others_precomp, opacities, scales, rotations, None)
        depth = int(tracer_settings.max_trace_depth)
        if depth == 0 or ray_o.numel() == 0 or means3D.shape[0] == 0:
            return _TraceSurfels.apply(ray_o, ray_d, *args, tracer_settings._replace(max_trace_depth=0), bool(start_from_first), self.nodes, self.caps)
        return self._forward_bounces(ray_o, ray_d, args, tracer_settings, bool(start_from_first))

    def bounce_caps(self, k):
        """Bounce stage k traces a different ray population (fewer, less coherent rays) than stage 0: its own capacity state."""
        st = self.__dict__.setdefault("_bounce_caps", {})
        if k not in st:
            st[k] = CapState(self.caps.cap)
        return st[k]

    def _forward_bounces(self, ray_o, ray_d, args, settings, start_from_first):
        """max_trace_depth > 0 (gaussian2d_sampler.py:413-426, optix_utils.py:117-118): every stage is one bounce-free traced call
        (the differentiable `_TraceSurfels` node, list path), glued by two differentiable kernels of envgs_amd.fused (bounce_rays: the next
        stage's rays; bounce_blend: the next stage's colour blended back -- one launch each way; round 4, torch expressions until then):

            stage k bounces where aux_k[0] > specular_threshold, acc_k > 0.5 and |norm_k| > 0
            o_{k+1} = o_k + d_k * dpt_k / acc_k          d_{k+1} = d_k - 2 (d_k . n) n,  n = norm_k / |norm_k|       (t_min = 1e-3)
            rgb = (1 - s_0) rgb_0 + s_0 ((1 - s_1) rgb_1 + s_1 (...)),   s_k = aux_k[0]

        so the backward IS the derivative of the returned `rgb`: through every stage's colour, through the blend weights s_k (into the
        `others_precomp` channel and the opacities / geometry that composite it), and through the reflected-ray construction into the
        previous stage's depth, accumulation and normal.  dpt / acc / norm / dist / aux are stage 0's, as in the single-stage call; `wet` is summed
        over all stages;
        `mid` holds the 16 channels of every stage (non-differentiable)."""
        depth = int(settings.max_trace_depth)
        thr = float(settings.specular_threshold)
        s0 = settings._replace(max_trace_depth=0)
        lead = tuple(ray_o.shape[:-1])
        from . import fused
        o = ray_o.reshape(-1, 3); d = ray_d.reshape(-1, 3)
        R = o.shape[0]
        dev = o.device
        was = self.caps.colour_only
        self.caps.colour_only = False                  # (its depth / accumulation / normal / specular outputs build the next stage's rays)
        chain = _BounceChain() if self.caps.defer_reduce else None        # set_deferred_surfel_gradients: the stages' backwards share their accumulators
        try:
            out0 = _TraceSurfels.apply(o, d, *args, s0, start_from_first, self.nodes, self.caps, (chain, 0) if chain else None)
        finally:
            self.caps.colour_only = was
        stages = [dict(o=o, d=d, out=out0, idx=torch.arange(R, device=dev), sel=None)]
        for k in range(1, depth + 1):
            p = stages[-1]
            rgb, dpt, acc, norm = p["out"][0], p["out"][1], p["out"][2], p["out"][3]
            aux = p["out"][5]
            with torch.no_grad():
                nl = norm.norm(dim=-1, keepdim=True)
                go = ((aux[:, 0:1] > thr) & (acc > 0.5) & (nl > 0.0))[:, 0]
                sel = go.nonzero(as_tuple=False)[:, 0]
            if sel.numel() == 0:
                break
            # one launch each way (envgs_amd.fused.bounce_rays) instead of five gathers, a norm, two divisions and the reflection -- whose
            # backward, through advanced indexing, was a SORTED index_put: 4 ms of radix sorts per 1200x1600 step
            o2, d2 = fused.bounce_rays(p["o"], p["d"], dpt, acc, norm, sel)
            bc = self.bounce_caps(k)
            was_k = bc.colour_only, bc.defer_reduce
            bc.colour_only = k == depth                 # the last stage's other outputs go into `mid` (no gradient) and nowhere else
            bc.defer_reduce = chain is not None
            try:
                out = _TraceSurfels.apply(o2, d2, *args, s0, 2, self.nodes, bc, (chain, k) if chain else None)
            finally:
                bc.colour_only, bc.defer_reduce = was_k  # (the promises are this call's, not the stage state's: ADVICE r4)
            stages.append(dict(o=o2, d=d2, out=out, idx=p["idx"].index_select(0, sel), sel=sel))
        col = stages[-1]["out"][0]
        for k in range(len(stages) - 2, -1, -1):
            p, c = stages[k], stages[k + 1]
            prgb = p["out"][0]
            col = fused.bounce_blend(prgb, p["out"][5], col, c["sel"])
        with torch.no_grad():
            mid = torch.zeros(R, 16 * (depth + 1), dtype=torch.float32, device=dev)
            for k, st in enumerate(stages):
                r_, dp_, ac_, no_, _, au_ = st["out"][:6]
                fused.bounce_pack_mid(mid, k, depth + 1, (st["idx"] if k else None), st["o"], st["d"], dp_, ac_, no_, au_, r_)
        rgb0, dpt0, acc0, norm0, dist0, aux0, mid0, wet = out0
        with torch.no_grad():
            # wet = the blend weights every surfel received over ALL stages: a surfel that only bounce rays blend contributes to the returned
            # colour and receives gradients, so the caller's visibility filter (wet > 0, optix_utils.py:203-213), its densification statistics
            # and the sparse Adam must see it (the OptiX ray-gen loop adds the weight wherever it composites a hit)
            for st in stages[1:]:
                wet = wet + st["out"][7]
        return (col.reshape(lead + (3,)), dpt0.reshape(lead + (1,)), acc0.reshape(lead + (1,)), norm0.reshape(lead + (3,)),
                dist0.reshape(lead + (1,)), aux0.reshape(lead + (2,)), mid.reshape(lead + (16 * (depth + 1),)), wet)


def last_hit_lists():
    """Tests (KEEP_LISTS on): the sorted per-ray hit lists of the most recent list-path forward -- (ids (R,cap) int32 front to back,
    tbits (R,cap) int32 = float bits of the hit distances, n_used (R,) hits composited before termination, hit_cnt (R,) hits found;
    > cap = the ray took the K-buffer path)."""
    L = LAST_STATS.get("lists")
    if L is None:
        return None
    hl, n_used, hit_cnt = L
    return hl[:, :, 1].contiguous(), hl[:, :, 0].contiguous(), n_used, hit_cnt


def last_entry_counts():
    """Diagnostics of the last list-path forward that prepared a record backward: (entries merged in the per-batch tables,
    single entries that found no room in a table).  Host sync."""
    ne = LAST_STATS.get("n_entries")
    if ne is None:
        return 0, 0
    v = ne.sum(0).cpu()
    return int(v[0]), int(v[1])


def last_trace_counts():
    """(composited hits, BVH node visits, traversal rounds) of the most recent forward (synchronises)."""
    c = LAST_STATS.get("counters")
    if c is None:
        return None
    w = c.cpu()
    v = w[2:20].view(torch.int64)
    return dict(coop_cycles=dict(expand=int(v[6]), walk=int(v[7]), wait=int(v[8])), hits=int(v[0]), node_visits=int(v[1]), rounds=int(v[2]), found=int(v[3]), packet_nodes=int(v[4]), packet_leaves=int(v[5]),
                max_list=int(w[1]), sparse_hits=int(w[64]), cap=LAST_STATS["caps"].cap, compact_rows=LAST_STATS.get("rows", 0), rays_without_rows=int(w[21]), rays=LAST_STATS["R"], stack_overflows=int(w[20]))
