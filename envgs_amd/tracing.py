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


def build_bvh(vertices, opacities=None, debug=False):
    """LBVH over the (4P,3) quad vertices (leaf boxes tightened by `opacities` when given).  Returns (nodes: envgs_bvh_node_floats(P) floats, P)."""
    lib = _lib.load()
    v = _f32c(vertices.detach())
    if v.device.type != "cuda":
        raise RuntimeError("envgs_amd tracer needs tensors on the GPU (got %s); there is no CPU path" % v.device)
    if v.dim() != 2 or v.shape[1] != 3 or v.shape[0] % 4 != 0:
        raise RuntimeError("vertices must be (4P,3) in the get_disks layout, got %s" % (tuple(v.shape),))
    P = v.shape[0] // 4
    dev = v.device
    nodes = torch.empty(lib.envgs_bvh_node_floats(P), dtype=torch.float32, device=dev)    # binary nodes, then the 4-wide nodes
    tb = lib.envgs_bvh_temp_bytes(P)
    temp = torch.empty(max(tb, 1), dtype=torch.uint8, device=dev)
    op = None if opacities is None else _f32c(opacities.detach()).reshape(-1)
    if op is not None and op.numel() != P:
        raise RuntimeError("opacities (%d) do not match the %d surfels of the vertex buffer" % (op.numel(), P))
    _lib.check(lib.envgs_bvh_build(P, _lib.ptr(v), _lib.ptr(op), _lib.ptr(nodes), _lib.ptr(temp), tb, 1 if debug else 0, _stream(dev)),
               "envgs_bvh_build")
    return nodes, P


def _cfg(settings, P, R, shs, others, start_from_first, ray_shape):
    from .raster import sh_degree_of
    deg = sh_degree_of(settings.sh_degree)
    rh, rw = (int(ray_shape[0]), int(ray_shape[1])) if len(ray_shape) == 2 else (0, 0)
    bg_len = min(int(settings.bg.numel()), 3)
    return _lib.TraceCfg(P, R, deg, 0 if shs is None else int(shs.shape[1]), int(settings.max_trace_depth),
                         (2 if start_from_first == 2 else (1 if start_from_first else 0)), 0 if others is None else 1, bg_len, 1 if settings.debug else 0,
                         rh, rw, float(settings.scale_modifier), float(settings.specular_threshold))


NCOPY = 8                    # must equal NCOPY in csrc/trace_common.h
HIT_CAP = {"cap": 512}
SORT_RAYS = {"on": True}     # coherence-sort the rays (direction, origin) before tracing
USE_RECORDS = {"on": True}   # atomic-free backward (one record per (batch, surfel) entry, grouped by surfel); False = cooperative atomic flush


_ASYNC = {}      # pinned host mirrors of two device counters + the events that say when they are valid


def _mirror(name, dev):
    m = _ASYNC.get((name, dev))
    if m is None:
        m = _ASYNC[(name, dev)] = dict(host=torch.zeros(1, dtype=torch.int32).pin_memory(), event=torch.cuda.Event(), valid=False)
    return m


def _next_cap(dev):
    """Capacity of the per-ray hit lists: 20 % above the longest list of the PREVIOUS call, read through a pinned mirror that was
    copied asynchronously at the end of that call -- no host sync on the hot path."""
    if HIT_CAP.get("force"):                       # tests: pin the capacity (e.g. tiny, to exercise the overflow hand-off)
        return int(HIT_CAP["force"])
    m = _mirror("max_list", dev)
    if m["valid"] and m["event"].query():
        mx = int(m["host"][0])
        want = ((int(mx * 1.2) + 8 + 63) // 64) * 64          # 20 % headroom, multiple of 64 entries (512 B)
        HIT_CAP["cap"] = max(64, min(want, 1024))
    return HIT_CAP["cap"]


BOUNCE_LISTS = {"on": True}  # specular bounces as one list-path trace per stage (False: all stages inside the K-buffer kernel)


def _trace_forward_bounces(nodes, ray_o, ray_d, means3D, shs, colors_precomp, others_precomp, opacities, scales, rotations, settings,
                           start_from_first, need_grad):
    """max_trace_depth > 0 on the list path: stage 0 is the ordinary bounce-free trace (it alone is differentiated and it alone feeds
    `wet`), every further stage is a forward-only list-path trace of the rays that bounce -- o + d * dpt / acc along d - 2 (d.n) n when
    aux[0] > specular_threshold and acc > 0.5 -- and the stage colours are blended back to front, (1 - s_k) rgb_k + s_k rgb_{k+1}.
    Same semantics as the in-kernel stages of the K-buffer path (and the oracle); ~20x faster on the bench scene."""
    depth = int(settings.max_trace_depth)
    s0 = settings._replace(max_trace_depth=0)
    lead = tuple(ray_o.shape[:-1])
    outs, saved = trace_forward(nodes, ray_o, ray_d, means3D, shs, colors_precomp, others_precomp, opacities, scales, rotations, s0,
                                start_from_first, need_grad=need_grad)
    rgb0, dpt0, acc0, norm0, dist0, aux0, mid0, wet = outs
    R = saved["ro"].shape[0]
    dev = means3D.device
    thr = float(settings.specular_threshold)
    flat = lambda t, c: t.reshape(R, c)
    stages = [dict(o=saved["ro"], d=saved["rd"], rgb=flat(rgb0, 3), dpt=flat(dpt0, 1), acc=flat(acc0, 1), norm=flat(norm0, 3), aux=flat(aux0, 2),
                   idx=torch.arange(R, device=dev))]
    for k in range(1, depth + 1):
        p = stages[-1]
        nl = p["norm"].norm(dim=-1, keepdim=True)
        go = ((p["aux"][:, 0:1] > thr) & (p["acc"] > 0.5) & (nl > 0.0))[:, 0]
        sel = go.nonzero(as_tuple=False)[:, 0]
        if sel.numel() == 0:
            break
        o, d = p["o"][sel], p["d"][sel]
        nh = p["norm"][sel] / nl[sel]
        tdep = p["dpt"][sel] / p["acc"][sel]
        dn = (d * nh).sum(-1, keepdim=True)
        o2 = (o + d * tdep).contiguous()
        d2 = (d - 2.0 * dn * nh).contiguous()
        with torch.no_grad():
            (r2, dp2, ac2, no2, _, au2, _, _), _ = trace_forward(nodes, o2, d2, means3D, shs, colors_precomp, others_precomp, opacities, scales, rotations,
                                                                 s0, 2, need_grad=False)
        stages.append(dict(o=o2, d=d2, rgb=r2, dpt=dp2, acc=ac2, norm=no2, aux=au2, idx=p["idx"][sel], sel=sel))
    # blend back to front (each stage lives on the subset of its parent's rays that bounced)
    col = stages[-1]["rgb"]
    for k in range(len(stages) - 2, -1, -1):
        p, c = stages[k], stages[k + 1]
        out = p["rgb"].clone()
        s = p["aux"][c["sel"], 0:1]
        out[c["sel"]] = (1.0 - s) * p["rgb"][c["sel"]] + s * col
        col = out
    mid = torch.zeros(R, 16 * (depth + 1), dtype=torch.float32, device=dev)
    for k, st in enumerate(stages):
        mid[st["idx"], 16 * k:16 * k + 16] = torch.cat([st["o"], st["d"], st["dpt"], st["acc"], st["norm"], st["aux"], st["rgb"]], dim=1)
    outs = (col.reshape(lead + (3,)), dpt0, acc0, norm0, dist0, aux0, mid.reshape(lead + (16 * (depth + 1),)), wet)
    return outs, saved


def trace_forward(nodes, ray_o, ray_d, means3D, shs, colors_precomp, others_precomp, opacities, scales, rotations, settings,
                  start_from_first, use_lists=True, need_grad=True):
    if (int(settings.max_trace_depth) > 0 and use_lists and BOUNCE_LISTS["on"] and HIT_CAP.get("force", 1) != 0 and means3D.shape[0] > 0
            and ray_o.numel() > 0):
        return _trace_forward_bounces(nodes, ray_o, ray_d, means3D, shs, colors_precomp, others_precomp, opacities, scales, rotations, settings,
                                      start_from_first, need_grad)
    lib = _lib.load()
    dev = means3D.device
    lead = tuple(ray_o.shape[:-1])
    ro = _f32c(ray_o).reshape(-1, 3); rd = _f32c(ray_d).reshape(-1, 3)
    R, P = ro.shape[0], means3D.shape[0]
    means3D = _f32c(means3D); opacities = _f32c(opacities); scales = _f32c(scales); rotations = _f32c(rotations)
    shs = _f32c(shs); colors_precomp = _f32c(colors_precomp); others_precomp = _f32c(others_precomp)
    bg = _f32c(settings.bg).reshape(-1).to(dev)
    cfg = _cfg(settings, P, R, shs, others_precomp, start_from_first, lead)
    ND = cfg.max_trace_depth + 1
    f32 = dict(dtype=torch.float32, device=dev)
    srec = torch.empty(max(P, 1), 16, **f32)
    counters = torch.empty(96, dtype=torch.int32, device=dev)
    rgb = torch.empty(R, 3, **f32); dpt = torch.empty(R, 1, **f32); acc = torch.empty(R, 1, **f32)
    norm = torch.empty(R, 3, **f32); dist = torch.empty(R, 1, **f32); aux = torch.empty(R, 2, **f32)
    mid = torch.empty(R, 16 * ND, **f32); wet = torch.empty(P, 1, **f32); final_T = torch.empty(R, **f32)
    cap = _next_cap(dev) if (use_lists and ND == 1 and P > 0 and R > 0) else 0
    lists = None
    keep = {}
    if cap:
        i32 = dict(dtype=torch.int32, device=dev)
        keep = dict(hit_lists=torch.empty(R, cap, 2, **i32), hit_cnt=torch.empty(R, **i32), n_used=torch.empty(R, **i32),
                    spill=torch.empty(lib.envgs_trace_stack_spill_ints(R), **i32), surf_acc=torch.empty(P, NCOPY, dtype=torch.int64, device=dev),
                    surf_cnt=torch.empty(P, NCOPY, **i32),
                    surf_off=torch.empty(P, NCOPY, **i32))
        sb = lib.envgs_raster_scan_temp_bytes(NCOPY * P)
        keep["scan_temp"] = torch.empty(max(sb, 1), dtype=torch.uint8, device=dev)
        rb = lib.envgs_trace_ray_sort_temp_bytes(R)
        keep.update(ray_keys=torch.empty(2 * R, **i32), ray_order=torch.empty(2 * R, **i32),
                    ray_sort_temp=torch.empty(max(rb, 1), dtype=torch.uint8, device=dev))
        srt = SORT_RAYS["on"]
        if need_grad and USE_RECORDS["on"]:
            # what the record backward needs from the forward: per-hit state, and the (batch, surfel) entries with their (lane, k) pairs
            nbatch = (R + 63) // 64
            keep.update(hit_state=torch.empty(R, cap, 12 if others_precomp is not None else 8, **f32), entries=torch.empty(nbatch, 64 * cap, dtype=torch.int64, device=dev),
                        pairs=torch.empty(nbatch, 64 * cap, **i32), n_entries=torch.empty(nbatch, 2, **i32))
        lists = _lib.TraceLists(keep["hit_lists"].data_ptr(), keep["hit_cnt"].data_ptr(), keep["n_used"].data_ptr(), cap,
                                keep["spill"].data_ptr(), keep["surf_acc"].data_ptr(), keep["surf_cnt"].data_ptr(), keep["surf_off"].data_ptr(),
                                keep["scan_temp"].data_ptr(), sb, keep["ray_keys"].data_ptr() if srt else None,
                                keep["ray_order"].data_ptr() if srt else None, keep["ray_sort_temp"].data_ptr() if srt else None, rb, None, 0,
                                *[(keep[k].data_ptr() if k in keep else None) for k in ("hit_state", "entries", "pairs", "n_entries")])
    p = _lib.ptr
    _lib.check(lib.envgs_trace_forward(cfg, p(nodes), p(ro), p(rd), p(means3D), p(scales), p(rotations), p(opacities), p(shs),
                                       p(colors_precomp), p(others_precomp), p(bg), p(srec), p(counters), p(rgb), p(dpt), p(acc),
                                       p(norm), p(dist), p(aux), p(mid), p(wet), p(final_T), lists, _stream(dev)), "envgs_trace_forward")
    LAST_STATS.update(P=P, R=R, counters=counters, n_entries=keep.get("n_entries"))
    if cap:
        # asynchronous read-backs for later: the longest list (sizes the next call's cap) and the number of gradient records
        m = _mirror("max_list", dev)
        m["host"].copy_(counters[1:2], non_blocking=True); m["event"].record(torch.cuda.current_stream(dev)); m["valid"] = True
        keep["n_rec_host"] = torch.zeros(1, dtype=torch.int32).pin_memory()
        keep["n_rec_host"].copy_(keep["surf_off"].view(-1)[NCOPY * P - 1:NCOPY * P], non_blocking=True)
        keep["n_rec_event"] = torch.cuda.Event(); keep["n_rec_event"].record(torch.cuda.current_stream(dev))
    saved = dict(cfg=cfg, nodes=nodes, ro=ro, rd=rd, means3D=means3D, scales=scales, rotations=rotations, opacities=opacities,
                 shs=shs, colors_precomp=colors_precomp, others=others_precomp, bg=bg, srec=srec, counters=counters,
                 rgb=(rgb if ND == 1 else mid[:, 13:16].contiguous()),      # the backward differentiates STAGE 0: its own colour, not the blend
                 dpt=dpt, acc=acc, norm=norm, aux=aux, final_T=final_T, lead=lead, lists=lists, keep=keep, cap=cap)
    outs = (rgb.reshape(lead + (3,)), dpt.reshape(lead + (1,)), acc.reshape(lead + (1,)), norm.reshape(lead + (3,)),
            dist.reshape(lead + (1,)), aux.reshape(lead + (2,)), mid.reshape(lead + (16 * ND,)), wet)
    return outs, saved


def trace_backward(saved, g_rgb, g_dpt, g_acc, g_norm, g_aux):
    lib = _lib.load()
    cfg = saved["cfg"]
    P, R = cfg.P, cfg.num_rays
    dev = saved["ro"].device
    f32 = dict(dtype=torch.float32, device=dev)
    z = lambda g, c: torch.zeros(R, c, **f32) if g is None else _f32c(g).reshape(R, c)
    g_rgb, g_dpt, g_acc, g_norm, g_aux = z(g_rgb, 3), z(g_dpt, 1), z(g_acc, 1), z(g_norm, 3), z(g_aux, 2)
    shs, others = saved["shs"], saved["others"]
    geo_rec = torch.empty(max(P, 1), 16, **f32)
    dmeans = torch.empty(P, 3, **f32); dgrads3D = torch.empty(P, 3, **f32); dscales = torch.empty(P, 2, **f32)
    drots = torch.empty(P, 4, **f32); dopac = torch.empty(P, 1, **f32)
    dshs = torch.empty_like(shs) if shs is not None else None
    dcolors = torch.empty(P, 3, **f32) if shs is None else None
    dothers = torch.empty(P, 2, **f32) if others is not None else None
    dro = torch.empty(R, 3, **f32); drd = torch.empty(R, 3, **f32)
    p = _lib.ptr
    s = saved
    lists = s["lists"]
    records = None
    if lists is not None and USE_RECORDS["on"] and "hit_state" in s["keep"]:
        # atomic-free backward: one 256 B record per (batch, surfel) entry, grouped by surfel.  The count is known on the device
        # (inclusive scan of the per-surfel entry counts, done at the end of the forward); reading it is the one host sync here.
        s["keep"]["n_rec_event"].synchronize()         # copied at the end of the forward; long since complete
        n_rec = int(s["keep"]["n_rec_host"][0]) & 0xFFFFFFFF if P > 0 else 0
        if n_rec > 0:
            records = torch.empty(n_rec, 64, **f32)
            lists.records = records.data_ptr()
            lists.num_records = n_rec
    _lib.check(lib.envgs_trace_backward(cfg, p(s["nodes"]), p(s["ro"]), p(s["rd"]), p(s["means3D"]), p(s["scales"]), p(s["rotations"]),
                                        p(s["opacities"]), p(shs), p(s["colors_precomp"]), p(others), p(s["bg"]), p(s["srec"]),
                                        p(s["counters"]), p(s["rgb"]), p(s["dpt"]), p(s["acc"]), p(s["norm"]), p(s["aux"]), p(s["final_T"]),
                                        p(g_rgb), p(g_dpt), p(g_acc), p(g_norm), p(g_aux), p(geo_rec), p(dmeans), p(dgrads3D), p(dscales),
                                        p(drots), p(dopac), p(dshs), p(dcolors), p(dothers), p(dro), p(drd), lists, _stream(dev)),
               "envgs_trace_backward")
    lead = s["lead"]
    return dict(ray_o=dro.reshape(lead + (3,)), ray_d=drd.reshape(lead + (3,)), means3D=dmeans, grads3D=dgrads3D, shs=dshs,
                colors_precomp=dcolors, others_precomp=dothers, opacities=dopac, scales=dscales, rotations=drots)


class _TraceSurfels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray_o, ray_d, v, means3D, grads3D, shs, colors_precomp, others_precomp, opacities, scales, rotations,
                cov3D_precomp, tracer_settings, start_from_first, nodes):
        none = lambda t: None if (t is None or t.numel() == 0) else t
        outs, saved = trace_forward(nodes, ray_o, ray_d, means3D, none(shs), none(colors_precomp), none(others_precomp), opacities,
                                    scales, rotations, tracer_settings, start_from_first, need_grad=any(ctx.needs_input_grad))
        ctx.saved = saved
        ctx.in_dtypes = tuple(None if t is None else t.dtype for t in (ray_o, ray_d, means3D, grads3D, shs, colors_precomp,
                                                                        others_precomp, opacities, scales, rotations))
        rgb, dpt, acc, norm, dist, aux, mid, wet = outs
        ctx.mark_non_differentiable(dist, mid, wet)
        return rgb, dpt, acc, norm, dist, aux, mid, wet

    @staticmethod
    def backward(ctx, g_rgb, g_dpt, g_acc, g_norm, g_dist, g_aux, g_mid, g_wet):
        g = trace_backward(ctx.saved, g_rgb, g_dpt, g_acc, g_norm, g_aux)
        order = ("ray_o", "ray_d", "means3D", "grads3D", "shs", "colors_precomp", "others_precomp", "opacities", "scales", "rotations")
        vals = [None if (g[k] is None or dt is None) else g[k].to(dt) for k, dt in zip(order, ctx.in_dtypes)]
        ro, rd, m3, g3, sh, col, oth, op, sc, rot = vals
        return ro, rd, None, m3, g3, sh, col, oth, op, sc, rot, None, None, None, None


class SurfelTracer(nn.Module):
    def __init__(self):
        super().__init__()
        _lib.load()                       # fail at construction (optix_utils.py:24 creates the OptiX context here)
        self.nodes = None
        self.num_surfels = 0
        self._pending = None

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
        if rebuild or (self.nodes is None and self._pending is None):
            self._pending = vertices.detach()
            self.nodes = None
            self.num_surfels = vertices.shape[0] // 4

    def forward(self, ray_o, ray_d, v=None, *, means3D, grads3D=None, shs=None, colors_precomp=None, others_precomp=None,
                opacities=None, scales=None, rotations=None, cov3D_precomp=None, tracer_settings=None, start_from_first=True):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if cov3D_precomp is not None:
            raise Exception('The HIP surfel tracer intersects surfels analytically in world space and needs scales / rotations; '
                            'a precomputed screen-space transMat (cov3D_precomp) cannot be traced.')
        if scales is None or rotations is None:
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
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
            self.nodes, self.num_surfels = build_bvh(self._pending, opacities)
            self._pending = None
        if grads3D is None:
            grads3D = torch.zeros_like(means3D)
        e = torch.Tensor([])
        return _TraceSurfels.apply(ray_o, ray_d, v, means3D, grads3D, e if shs is None else shs,
                                   e if colors_precomp is None else colors_precomp, e if others_precomp is None else others_precomp,
                                   opacities, scales, rotations, None, tracer_settings, bool(start_from_first), self.nodes)


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
    v = w[2:10].view(torch.int64)
    return dict(hits=int(v[0]), node_visits=int(v[1]), rounds=int(v[2]), found=int(v[3]), max_list=int(w[1]), cap=HIT_CAP["cap"],
                rays=LAST_STATS["R"], stack_overflows=int(w[10]))
