"""Host-side mirror of the reference's surfel rasterizer interface, over the C-ABI of include/envgs_raster.h.

Same names, argument meaning and error behaviour as the packages the reference imports at
easyvolcap/utils/gaussian2d_utils.py:1013-1015 and calls at :1025-1038 / :1089-1099:

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None)
        -> rendered_image (C,H,W), radii (P,) int32, allmap (7,H,W), weight (P,1)

The channel count C (3 / 5 / 7) is the only thing that differs between the three packages; it is bound
by `make_package(C)`.  PyTorch is plumbing here (device memory, streams, autograd graph); every stage
runs in hand-written HIP.  No fallback: a missing library raises at first use.
"""
from typing import NamedTuple

import torch
from torch import nn

from . import _lib


LAST_STATS = {}     # N (tile instances) etc. of the most recent forward; read by bench.py for the roofline figures


class GaussianRasterizationSettings(NamedTuple):
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


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _featc(t):
    """Per-surfel FEATURE arrays (shs, colors_precomp) keep fp16 storage when the caller passes half tensors (cfg.feature_f16: converted
    on load inside the kernels, fp32 arithmetic and fp32 gradient buffers); anything else is fp32."""
    if t is None:
        return None
    if t.dtype != torch.float16:
        return _f32c(t)
    return t.contiguous()


def _stream(dev):
    return _lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


import weakref

_DEG_CACHE = {}      # id(tensor) -> (weakref to the tensor, in-place version, value)


def sh_degree_of(deg):
    """settings.sh_degree is a 1-element tensor in the reference (a registered buffer on the GPU, gaussian2d_utils.py:294) and
    reading it is a host sync.  The value is cached per tensor OBJECT and in-place version (the reference passes the same buffer
    every iteration and bumps it in place once every 1000 iterations); a fresh tensor is always read.  The cache is keyed by id()
    and validated through a weak reference -- no Tensor.__eq__ (a kernel launch + sync on a GPU buffer) is ever evaluated."""
    if not torch.is_tensor(deg):
        return int(deg)
    key = id(deg)
    hit = _DEG_CACHE.get(key)
    if hit is not None and hit[0]() is deg and hit[1] == deg._version:
        return hit[2]
    v = int(deg.reshape(-1)[0].item())
    if len(_DEG_CACHE) > 64:                                  # drop entries whose tensor died (ids are recycled)
        for k in [k for k, h in _DEG_CACHE.items() if h[0]() is None]:
            del _DEG_CACHE[k]
    _DEG_CACHE[key] = (weakref.ref(deg), deg._version, v)
    return v


SPECULATIVE_N = {"on": True}  # size the binning buffers from the previous call's instance count instead of waiting for this call's (see rasterize_forward)
_N_GUESS = {}                  # (device, P, H, W) -> capacity for the next call
_N_MIRROR = {}


def _n_mirror(dev):
    m = _N_MIRROR.get(dev.index)
    if m is None:
        m = _N_MIRROR[dev.index] = dict(host=torch.empty(1, dtype=torch.int32).pin_memory(), event=torch.cuda.Event())
    return m


# fp16 FEATURE STORAGE without fp16 GRADIENTS (BASELINE configs[4]).  A caller that passes half shs / colors_precomp gets the gradient back in
# half -- autograd casts an input's gradient to the input's dtype -- and at training magnitudes (1e-6 and below) that is fp16's subnormal
# range: whole blocks of dL/dSH flush to zero.  With this switch on, fp32 feature tensors are accepted as usual and the HALF COPY is made
# inside the autograd node: the kernels read 2 B per value, the gradient leaves the node in fp32.  Shared by the three raster packages and
# the tracer (envgs_amd.set_feature_storage).
FEATURE_STORAGE = {"f16": False}


def _store(t):
    """The tensor the kernels read for a feature input: a half copy when fp16 feature storage is selected, else the input itself."""
    if t is not None and FEATURE_STORAGE["f16"] and t.dtype == torch.float32 and t.numel() > 0:
        return t.detach().half()
    return t


CONTRIB_MASK = {"on": True}   # forward records which pixel quadrants blended each tile instance; the backward walks exactly those (tests switch it off to cover the geometric fallback)


def _cfg(settings, P, C, sh_coeffs, bg_len, f16=False):
    deg = sh_degree_of(settings.sh_degree)
    return _lib.RasterCfg(P, deg, sh_coeffs, C, int(settings.image_width), int(settings.image_height), bg_len,
                          1 if settings.debug else 0, float(settings.scale_modifier), float(settings.tanfovx),
                          float(settings.tanfovy), 1 if f16 else 0)


def rasterize_forward(C, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings,
                      keep_binning=False):
    """R1..R6 through the C-ABI.  Returns (outputs, saved-state dict).  All tensors live on means3D.device."""
    lib = _lib.load()
    dev = means3D.device
    if dev.type != "cuda":
        raise RuntimeError("envgs_amd rasterizer needs tensors on the GPU (got %s); there is no CPU path" % dev)
    P = means3D.shape[0]
    H, W = int(settings.image_height), int(settings.image_width)
    means3D = _f32c(means3D); opacities = _f32c(opacities)
    shs = _featc(shs); colors_precomp = _featc(colors_precomp)
    f16 = any(t is not None and t.dtype == torch.float16 for t in (shs, colors_precomp))
    scales = _f32c(scales); rotations = _f32c(rotations); cov3D_precomp = _f32c(cov3D_precomp)
    if shs is not None and C != 3:
        raise RuntimeError("in-kernel SH evaluation produces 3 channels; the %d-channel rasterizer needs colors_precomp" % C)
    if colors_precomp is not None and colors_precomp.shape[-1] != C:
        raise RuntimeError("colors_precomp has %d channels, this rasterizer composites %d" % (colors_precomp.shape[-1], C))
    bg = _f32c(settings.bg).reshape(-1).to(dev)
    view = _f32c(settings.viewmatrix).to(dev); proj = _f32c(settings.projmatrix).to(dev)
    campos = _f32c(settings.campos).reshape(-1).to(dev)
    sh_coeffs = 0 if shs is None else int(shs.shape[1])
    cfg = _cfg(settings, P, C, sh_coeffs, min(int(bg.numel()), C), f16)
    stream = _stream(dev)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)

    geom = torch.empty(P, 16, **f32)
    radii = torch.empty(P, **i32)
    tiles = torch.empty(P, **i32)
    offsets = torch.empty(P, **i32)
    rgb = torch.empty(P, 3, **f32) if shs is not None else None
    clamped = torch.empty(P, 3, dtype=torch.uint8, device=dev) if shs is not None else None
    scan_bytes = lib.envgs_raster_scan_temp_bytes(P)
    scan_temp = torch.empty(max(scan_bytes, 1), dtype=torch.uint8, device=dev)
    p = _lib.ptr
    tiles_n = ((W + 15) // 16) * ((H + 15) // 16)
    colors = rgb if shs is not None else colors_precomp
    out_color = torch.empty(C, H, W, **f32)
    allmap = torch.empty(7, H, W, **f32)
    final_T = torch.empty(3, H, W, **f32)
    n_contrib = torch.empty(2, H, W, **i32)
    weight = torch.empty(P, 1, **f32)
    ranges = torch.empty(tiles_n, 2, **i32)

    def bin_and_render(cap):
        """R3-R6 with N-sized buffers of `cap` entries (the instance count, or a guess: the kernels re-derive the count themselves)."""
        pairs = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
        keys_s = torch.empty(max(cap, 1), dtype=torch.int64, device=dev) if keep_binning else None
        point_list = torch.empty(max(cap, 1), **i32)
        bin_bytes = lib.envgs_raster_sort_temp_bytes(max(cap, 1), W, H)
        bin_temp = torch.empty(max(bin_bytes, 1), dtype=torch.uint8, device=dev)
        # per tile instance: the pixel quadrants that blended it (the backward walks exactly those)
        cmask = torch.empty(max(cap, 1), dtype=torch.uint8, device=dev) if CONTRIB_MASK["on"] else None
        _lib.check(lib.envgs_raster_bin_and_render(cfg, cap, p(geom), p(radii), p(colors), p(bg), p(pairs), p(keys_s), p(point_list),
                                                   p(bin_temp), bin_bytes, p(ranges), p(out_color), p(allmap), p(final_T),
                                                   p(n_contrib), p(weight), p(cmask), stream),
                   "envgs_raster_bin_and_render")
        return pairs, keys_s, point_list, cmask

    # The number of tile instances N sizes the binning buffers, and it is known only after the projection.  Waiting for it drains the
    # GPU's queue at the start of every step (and every launch after it is exposed until the host is ahead again: ~0.5 ms of idle GPU per
    # EnvGS step, measured from the kernel trace).  So: enqueue the projection, start an asynchronous read-back of N, enqueue R3-R6 with a
    # CAPACITY guessed from the previous calls of this shape -- and only then look at N, which has long arrived while the GPU still has the
    # binning and compositing queued.  A guess that was too small (never out of bounds, see envgs_raster.h) repeats R3-R6 with the exact size.
    key = (dev.index, P, H, W)
    guess = _N_GUESS.get(key) if SPECULATIVE_N["on"] else None
    mirror = _n_mirror(dev)
    if guess is None or P == 0:
        n_host = _lib.c_uint32(0)
        _lib.check(lib.envgs_raster_project(cfg, p(means3D), p(scales), p(rotations), p(opacities), p(shs), p(cov3D_precomp),
                                            p(view), p(proj), p(campos), p(geom), p(rgb), p(clamped), p(radii), p(tiles),
                                            p(offsets), p(scan_temp), scan_bytes, n_host, stream), "envgs_raster_project")
        N = int(n_host.value)
        bufs = bin_and_render(N)
    else:
        _lib.check(lib.envgs_raster_project(cfg, p(means3D), p(scales), p(rotations), p(opacities), p(shs), p(cov3D_precomp),
                                            p(view), p(proj), p(campos), p(geom), p(rgb), p(clamped), p(radii), p(tiles),
                                            p(offsets), p(scan_temp), scan_bytes, None, stream), "envgs_raster_project")
        mirror["host"].copy_(offsets[P - 1:P], non_blocking=True)
        mirror["event"].record(torch.cuda.current_stream(dev))
        cap = guess
        bufs = bin_and_render(cap)
        mirror["event"].synchronize()                        # (the copy was queued BEFORE R3-R6: this does not wait for them)
        N = int(mirror["host"][0]) & 0xFFFFFFFF
        if N > cap:
            LAST_STATS["n_guess_misses"] = LAST_STATS.get("n_guess_misses", 0) + 1
            bufs = bin_and_render(N)
    pairs, keys_s, point_list, cmask = bufs
    # next capacity: 15 % above this count (64 k granularity), and not below 97 % of the previous capacity -- views alternate, scenes change slowly
    _N_GUESS[key] = max(((max(int(N * 1.15), N + 4096) + 65535) // 65536) * 65536, int(0.97 * (guess or 0)))
    LAST_STATS.update(N=N, P=P, H=H, W=W, C=C)
    saved = dict(cfg=cfg, N=N, geom=geom, colors=colors, bg=bg, point_list=point_list, ranges=ranges, final_T=final_T,
                 n_contrib=n_contrib, contrib_mask=cmask, means3D=means3D, scales=scales, rotations=rotations, shs=shs, clamped=clamped,
                 cov3D_precomp=cov3D_precomp, radii=radii, view=view, proj=proj, campos=campos)
    if keep_binning:
        saved.update(tiles_touched=tiles, offsets=offsets, tile_pairs=pairs, keys_sorted=keys_s)
    return (out_color, radii, allmap, weight), saved


def render_audit(saved, lmax, skip_px=None, want_weight=False):
    """Parity audit (tests): re-runs the compositing kernel in its AUDIT instantiation on the saved binning state and returns
    (contrib (H*W, lmax) uint8, n_contrib (2,H,W), out_color) -- see envgs_raster_render_audit.  skip_px: (H,W) bool / uint8 tensor of pixels
    to leave out of the per-surfel weight sums; want_weight: additionally return that weight (P,)."""
    lib = _lib.load()
    cfg = saved["cfg"]
    dev = saved["geom"].device
    H, W, C, P = cfg.height, cfg.width, cfg.channels, cfg.P
    f32 = dict(dtype=torch.float32, device=dev)
    out_color = torch.empty(C, H, W, **f32); allmap = torch.empty(7, H, W, **f32); final_T = torch.empty(3, H, W, **f32)
    n_contrib = torch.empty(2, H, W, dtype=torch.int32, device=dev); weight = torch.empty(max(P, 1), **f32)
    contrib = torch.empty(H * W, int(lmax), dtype=torch.uint8, device=dev)
    skip = None if skip_px is None else skip_px.to(device=dev, dtype=torch.uint8).reshape(-1).contiguous()
    if skip is not None and skip.numel() != H * W:
        raise RuntimeError("skip_px must have H*W elements")
    p = _lib.ptr
    _lib.check(lib.envgs_raster_render_audit(cfg, p(saved["geom"]), p(saved["colors"]), p(saved["bg"]), p(saved["point_list"]),
                                             p(saved["ranges"]), p(out_color), p(allmap), p(final_T), p(n_contrib), p(weight),
                                             p(contrib), int(lmax), p(skip), _stream(dev)), "envgs_raster_render_audit")
    if want_weight:
        return contrib, n_contrib, out_color, weight[:P]
    return contrib, n_contrib, out_color


def rasterize_backward(saved, dL_dcolor, dL_dallmap):
    """R7+R8 through the C-ABI.  Returns dict of parameter gradients (None where not applicable)."""
    lib = _lib.load()
    cfg = saved["cfg"]
    P, C = cfg.P, cfg.channels
    dev = saved["geom"].device
    f32 = dict(dtype=torch.float32, device=dev)
    dL_dcolor = _f32c(dL_dcolor); dL_dallmap = _f32c(dL_dallmap)          # None = that output has no upstream gradient (NULL in the C-ABI)
    shs, cov = saved["shs"], saved["cov3D_precomp"]
    grad_rec = torch.empty(P, 32, **f32)
    dmeans3D = torch.empty(P, 3, **f32)
    dmeans2D = torch.empty(P, 3, **f32)
    dopac = torch.empty(P, 1, **f32)
    dscales = torch.empty(P, 2, **f32) if cov is None else None
    drots = torch.empty(P, 4, **f32) if cov is None else None
    dcov = torch.empty(P, 9, **f32) if cov is not None else None
    dshs = torch.empty(shs.shape, **f32) if shs is not None else None            # gradients are fp32 whatever the feature storage
    dcolors = torch.empty(P, C, **f32) if shs is None else None
    p = _lib.ptr
    _lib.check(lib.envgs_raster_backward(cfg, saved["N"], p(saved["geom"]), p(saved["colors"]), p(saved["bg"]),
                                         p(saved["point_list"]), p(saved["ranges"]), p(saved["final_T"]), p(saved["n_contrib"]),
                                         p(saved.get("contrib_mask")), p(dL_dcolor), p(dL_dallmap), p(saved["means3D"]), p(saved["scales"]), p(saved["rotations"]),
                                         p(shs), p(saved["clamped"]), p(cov), p(saved["radii"]), p(saved["view"]), p(saved["proj"]),
                                         p(saved["campos"]), p(grad_rec), p(dmeans3D), p(dmeans2D), p(dscales), p(drots), p(dshs),
                                         p(dcolors), p(dopac), p(dcov), _stream(dev)), "envgs_raster_backward")
    return dict(means3D=dmeans3D, means2D=dmeans2D, shs=dshs, colors_precomp=dcolors, opacities=dopac, scales=dscales,
                rotations=drots, cov3D_precomp=dcov, grad_rec=grad_rec)


def make_package(C):
    """Bind the channel count: returns (GaussianRasterizationSettings, GaussianRasterizer) for one drop-in package."""

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
            ctx.set_materialize_grads(False)          # outputs the loss does not use arrive as None (= NULL upstream pointer), not as buffers of zeros
            none = lambda t: None if (t is None or t.numel() == 0) else t
            outs, saved = rasterize_forward(C, means3D, _store(none(sh)), _store(none(colors_precomp)), opacities, none(scales),
                                            none(rotations), none(cov3Ds_precomp), raster_settings)
            ctx.saved = saved
            ctx.in_dtypes = tuple(None if t is None else t.dtype for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))
            color, radii, allmap, weight = outs
            ctx.mark_non_differentiable(radii, weight)
            return color, radii, allmap, weight

        @staticmethod
        def backward(ctx, grad_color, grad_radii, grad_allmap, grad_weight):
            saved = ctx.saved
            cfg = saved["cfg"]
            dev = saved["geom"].device
            g = rasterize_backward(saved, grad_color, grad_allmap)
            outs = (g["means3D"] if saved["cov3D_precomp"] is None or saved["shs"] is not None else None, g["means2D"],
                    g["shs"], g["colors_precomp"], g["opacities"], g["scales"], g["rotations"], g["cov3D_precomp"])
            outs = tuple(None if (o is None or dt is None) else o.to(dt) for o, dt in zip(outs, ctx.in_dtypes))
            return outs + (None,)

    class GaussianRasterizer(nn.Module):
        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def markVisible(self, positions):
            """Frustum test used by some 3DGS-lineage callers: view depth > 0.2."""
            with torch.no_grad():
                V = self.raster_settings.viewmatrix.to(positions.device).float()
                z = positions.float() @ V[:3, 2] + V[3, 2]
                return z > 0.2

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            rs = self.raster_settings
            if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
                raise Exception('Please provide excatly one of either SHs or precomputed colors!')
            if ((scales is None or rotations is None) and cov3D_precomp is None) or \
               ((scales is not None or rotations is not None) and cov3D_precomp is not None):
                raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
            e = torch.Tensor([])
            return _RasterizeGaussians.apply(means3D, means2D, e if shs is None else shs,
                                             e if colors_precomp is None else colors_precomp, opacities,
                                             e if scales is None else scales, e if rotations is None else rotations,
                                             e if cov3D_precomp is None else cov3D_precomp, rs)

    GaussianRasterizer.channels = C
    GaussianRasterizer._function = _RasterizeGaussians
    return GaussianRasterizationSettings, GaussianRasterizer
