"""numpy/ctypes front-end of oracle/surfel_raster_oracle.c (test infrastructure only).

Mirrors the call boundary of GaussianRasterizer (reference: easyvolcap/utils/gaussian2d_utils.py:1089-1099)
but returns every intermediate (R1..R6) so tests can compare stage by stage.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
TILE = 16


class _Cfg(ctypes.Structure):
    _fields_ = [("P", ctypes.c_int), ("D", ctypes.c_int), ("M", ctypes.c_int), ("C", ctypes.c_int),
                ("W", ctypes.c_int), ("H", ctypes.c_int), ("scale_modifier", ctypes.c_float)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("surfel_raster_oracle.c", "surfel_trace_oracle.c", "adam_oracle.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_scan.restype = ctypes.c_uint32
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def raster_forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, *, scales=None, rotations=None,
                   transmat_precomp=None, shs=None, colors_precomp=None, sh_degree=0, bg=None,
                   scale_modifier=1.0):
    """Returns a dict with every stage output.  Arrays are numpy, layouts as in the C header comments."""
    L = lib()
    means3D = _f32(means3D); opacities = _f32(opacities).reshape(-1)
    scales = _f32(scales); rotations = _f32(rotations); transmat_precomp = _f32(transmat_precomp)
    shs = _f32(shs); colors_precomp = _f32(colors_precomp)
    viewmatrix = _f32(viewmatrix); projmatrix = _f32(projmatrix); campos = _f32(campos)
    P = means3D.shape[0]
    if shs is not None:
        M, C = shs.shape[1], 3
    else:
        M, C = 0, colors_precomp.shape[1]
    bg = np.zeros(C, np.float32) if bg is None else _f32(bg).reshape(-1)
    cfg = _Cfg(P, int(sh_degree), M, C, int(W), int(H), float(scale_modifier))
    out = dict(cfg=cfg, W=W, H=H, C=C, M=M, P=P)
    transmat = np.zeros((P, 9), np.float32); normal_opacity = np.zeros((P, 4), np.float32)
    xy = np.zeros((P, 2), np.float32); depth = np.zeros(P, np.float32)
    radii = np.zeros(P, np.int32); tiles = np.zeros(P, np.uint32)
    rgb = np.zeros((P, C), np.float32); clamped = np.zeros((P, 3), np.uint8)
    L.orc_preprocess(ctypes.byref(cfg), _p(means3D), _p(scales), _p(rotations), _p(opacities), _p(shs),
                     _p(transmat_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), _p(transmat),
                     _p(normal_opacity), _p(xy), _p(depth), _p(radii), _p(tiles), _p(rgb), _p(clamped))
    colors = rgb if shs is not None else colors_precomp
    offsets = np.zeros(P, np.uint32)
    N = int(L.orc_scan(P, _p(tiles), _p(offsets)))
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ku = np.zeros(max(N, 1), np.uint64); vu = np.zeros(max(N, 1), np.uint32)
    ks = np.zeros(max(N, 1), np.uint64); pl = np.zeros(max(N, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    L.orc_bin(ctypes.byref(cfg), _p(xy), _p(depth), _p(radii), _p(offsets), ctypes.c_uint32(N), _p(ku), _p(vu),
              _p(ks), _p(pl), _p(ranges))
    out_color = np.zeros((C, H, W), np.float32); allmap = np.zeros((7, H, W), np.float32)
    final_T = np.zeros((3, H, W), np.float32); n_contrib = np.zeros((2, H, W), np.int32)
    weight = np.zeros(P, np.float64)
    L.orc_render_fwd(ctypes.byref(cfg), _p(ranges), _p(pl), _p(transmat), _p(xy), _p(normal_opacity), _p(colors),
                     _p(bg), ctypes.c_int(len(bg)), _p(out_color), _p(allmap), _p(final_T), _p(n_contrib), _p(weight))
    out.update(transmat=transmat, normal_opacity=normal_opacity, xy=xy, depth=depth, radii=radii,
               tiles_touched=tiles, rgb=rgb, clamped=clamped, colors=colors, offsets=offsets, N=N,
               keys_unsorted=ku[:N], vals_unsorted=vu[:N], keys_sorted=ks[:N], point_list=pl[:N], ranges=ranges,
               out_color=out_color, allmap=allmap, final_T=final_T, n_contrib=n_contrib, weight=weight, bg=bg,
               inputs=dict(means3D=means3D, opacities=opacities, scales=scales, rotations=rotations,
                           transmat_precomp=transmat_precomp, shs=shs, colors_precomp=colors_precomp,
                           viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos))
    return out


def raster_audit(fwd, want_contrib=False, canonical=True):
    """Fragility audit of the forward `fwd` (orc_render_audit).  canonical=True (round 4): `fragile` (H,W) bool = the pixels where a decision of
    the float code lies within the A-PRIORI few-ulp band that still separates two implementations of the canonical operation sequence;
    `legacy_fragile` = the round-1..3 definition (K x the float code's own distance from a double shadow, plus ill-conditioned pixels), reported
    only.  canonical=False: `fragile` is the legacy definition (attribution runs).  Also: illcond / relbad statistics, tainted (P,) bool,
    contrib (H*W, lmax) uint8 or None, lmax.  See the C comment for the definitions."""
    L = lib()
    cfg = fwd["cfg"]; H, W, P = fwd["H"], fwd["W"], fwd["P"]
    pl = fwd["point_list"] if fwd["N"] > 0 else np.zeros(1, np.uint32)
    r = fwd["ranges"].astype(np.int64)
    lmax = int((r[:, 1] - r[:, 0]).max()) if r.size else 0
    fragile = np.zeros(H * W, np.uint8); tainted = np.zeros(max(P, 1), np.uint8)
    contrib = np.zeros((H * W, max(lmax, 1)), np.uint8) if want_contrib else None
    L.orc_render_audit(ctypes.byref(cfg), _p(fwd["ranges"]), _p(pl), _p(fwd["transmat"]), _p(fwd["xy"]), _p(fwd["normal_opacity"]),
                       _p(fragile), _p(contrib), ctypes.c_int(max(lmax, 1)), _p(tainted), ctypes.c_int(1 if canonical else 0))
    legacy = ((fragile & 16) != 0).reshape(H, W)
    flips = ((fragile & 1) != 0).reshape(H, W)
    illcond = ((fragile & 2) != 0).reshape(H, W)
    return dict(fragile=(flips if canonical else legacy), flips=flips, illcond=illcond, relbad=((fragile & 4) != 0).reshape(H, W),
                legacy_fragile=legacy, canonical=bool(canonical), tainted=tainted[:P].astype(bool), contrib=contrib, lmax=max(lmax, 1))


def raster_dist64(fwd):
    """Distortion shadow of the forward `fwd` (orc_render_dist64): (dist64 (H,W) float64 = the distortion evaluated in double from the float code's own
    alphas and depths, bound (H,W) float64 = the a-priori fp32 rounding bound of the moment form, see the C comment)."""
    L = lib()
    cfg = fwd["cfg"]; H, W = fwd["H"], fwd["W"]
    pl = fwd["point_list"] if fwd["N"] > 0 else np.zeros(1, np.uint32)
    d64 = np.zeros(H * W, np.float64); bd = np.zeros(H * W, np.float64)
    L.orc_render_dist64(ctypes.byref(cfg), _p(fwd["ranges"]), _p(pl), _p(fwd["transmat"]), _p(fwd["xy"]), _p(fwd["normal_opacity"]), _p(d64), _p(bd))
    return d64.reshape(H, W), bd.reshape(H, W)


def sh_clamp_audit(fwd):
    """(P,) bool: surfels whose SH colour lies within fp32 rounding of the clamp at 0 in some channel (orc_sh_clamp_audit); None without SH."""
    inp = fwd["inputs"]
    if inp["shs"] is None:
        return None
    L = lib()
    frag = np.zeros(max(fwd["P"], 1), np.uint8)
    L.orc_sh_clamp_audit(ctypes.byref(fwd["cfg"]), _p(inp["means3D"]), _p(inp["campos"]), _p(inp["shs"]), _p(frag))
    return frag[:fwd["P"]].astype(bool)


def raster_weight(fwd, skip_px=None):
    """Per-surfel weight of the forward `fwd` with the pixels of skip_px ((H,W) bool) left out (orc_render_weight).
    Returns (weight (P,) float64, unc (P,) float64 = sum over the surfel's pixels of |w_f32 - w_f64|, the float code's own error)."""
    L = lib()
    cfg = fwd["cfg"]; P = fwd["P"]
    pl = fwd["point_list"] if fwd["N"] > 0 else np.zeros(1, np.uint32)
    skip = None if skip_px is None else np.ascontiguousarray(np.asarray(skip_px, np.uint8).reshape(-1))
    w = np.zeros(max(P, 1), np.float64); e = np.zeros(max(P, 1), np.float64)
    L.orc_render_weight(ctypes.byref(cfg), _p(fwd["ranges"]), _p(pl), _p(fwd["transmat"]), _p(fwd["xy"]), _p(fwd["normal_opacity"]),
                        _p(skip), _p(w), _p(e))
    return w[:P], e[:P]


def raster_backward(fwd, dL_dcolor, dL_dallmap, want_cond=False):
    """Gradients for the forward `fwd` (dict from raster_forward).  Returns dict of float32 arrays + raw records.
    want_cond: additionally `cond` = dict of sum |term| per gradient element (same keys): R7's records accumulated in absolute value and
    pushed through R8 with the absolute value of its (linear) map -- the magnitude of what each element is a sum of."""
    L = lib()
    cfg = fwd["cfg"]; P, C, M = fwd["P"], fwd["C"], fwd["M"]
    inp = fwd["inputs"]
    dL_dcolor = _f32(dL_dcolor); dL_dallmap = _f32(dL_dallmap)
    pl = fwd["point_list"] if fwd["N"] > 0 else np.zeros(1, np.uint32)
    dT = np.zeros((P, 9)); dn = np.zeros((P, 3)); dop = np.zeros(P); dcol = np.zeros((P, C)); dm2 = np.zeros((P, 2))
    ab = [np.zeros_like(x) for x in (dT, dn, dop, dcol, dm2)] if want_cond else [None] * 5
    L.orc_render_bwd(ctypes.byref(cfg), _p(fwd["ranges"]), _p(pl), _p(fwd["transmat"]), _p(fwd["xy"]),
                     _p(fwd["normal_opacity"]), _p(fwd["colors"]), _p(fwd["bg"]), ctypes.c_int(len(fwd["bg"])),
                     _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(dL_dcolor), _p(dL_dallmap), _p(dT), _p(dn), _p(dop),
                     _p(dcol), _p(dm2), *[_p(x) for x in ab])

    def r8(dT_, dn_, dm2_, dcol_):
        dmeans3D = np.zeros((P, 3), np.float32); dmeans2D = np.zeros((P, 3), np.float32)
        dscales = np.zeros((P, 2), np.float32); drots = np.zeros((P, 4), np.float32)
        dshs = np.zeros((P, max(M, 1), 3), np.float32); dtm = np.zeros((P, 9), np.float32)
        L.orc_preprocess_bwd(ctypes.byref(cfg), _p(inp["means3D"]), _p(inp["scales"]), _p(inp["rotations"]), _p(inp["shs"]),
                             _p(fwd["clamped"]), _p(inp["transmat_precomp"]), _p(fwd["transmat"]), _p(fwd["radii"]),
                             _p(inp["viewmatrix"]), _p(inp["projmatrix"]), _p(inp["campos"]), _p(dT_), _p(dn_), _p(dm2_),
                             _p(dcol_), _p(dmeans3D), _p(dmeans2D), _p(dscales), _p(drots), _p(dshs), _p(dtm))
        return dict(dmeans3D=dmeans3D, dmeans2D=dmeans2D, dscales=dscales, drots=drots,
                    dshs=dshs if inp["shs"] is not None else None,
                    dtransmat_precomp=dtm if inp["transmat_precomp"] is not None else None)

    out = r8(dT, dn, dm2, dcol)
    out.update(dcolors=dcol.astype(np.float32) if inp["shs"] is None else None, dopacities=dop.astype(np.float32),
               rec_dT=dT, rec_dnormal=dn, rec_dcolor=dcol, rec_dmean2D=dm2, rec_dopacity=dop)
    def through_r8(aT, an, aop, acol, am2):
        # R8 is linear in the records: sum_k |J_ik| * x_k = sum over the record words of |R8(e_k * x_k)|   (x >= 0)
        z = lambda a: np.zeros_like(a)
        res = None
        probes = [(0, k) for k in range(9)] + [(1, k) for k in range(3)] + [(2, k) for k in range(2)] + [(3, k) for k in range(C)]
        for which, k in probes:
            args = [z(aT), z(an), z(am2), z(acol)]
            src = (aT, an, am2, acol)[which]
            args[which][:, k] = src[:, k]
            o = r8(*args)
            if res is None:
                res = {kk: (np.abs(v).astype(np.float64) if v is not None else None) for kk, v in o.items()}
            else:
                for kk, v in o.items():
                    if v is not None: res[kk] += np.abs(v)
        res["dcolors"] = acol if inp["shs"] is None else None
        res["dopacities"] = aop
        return res

    if want_cond:
        out["cond"] = through_r8(*ab)
        # the oracle's own fp32 uncertainty, term by term (orc_render_bwd_unc), pushed through |R8| the same way
        un = [np.zeros_like(x) for x in (dT, dn, dop, dcol, dm2)]
        L.orc_render_bwd_unc(ctypes.byref(cfg), _p(fwd["ranges"]), _p(pl), _p(fwd["transmat"]), _p(fwd["xy"]),
                             _p(fwd["normal_opacity"]), _p(fwd["colors"]), _p(fwd["bg"]), ctypes.c_int(len(fwd["bg"])),
                             _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(dL_dcolor), _p(dL_dallmap), *[_p(x) for x in un])
        out["unc"] = through_r8(*un)
    return out
