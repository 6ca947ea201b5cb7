"""numpy/ctypes front-end of oracle/surfel_trace_oracle.c (test infrastructure only).
Mirrors SurfelTracer.__call__ (reference boundary: easyvolcap/utils/optix_utils.py:188-201)."""
import ctypes

import numpy as np

from .raster import lib, _p, _f32


class _TCfg(ctypes.Structure):
    _fields_ = [("P", ctypes.c_int), ("R", ctypes.c_int), ("D", ctypes.c_int), ("M", ctypes.c_int),
                ("max_trace_depth", ctypes.c_int), ("start_from_first", ctypes.c_int), ("has_others", ctypes.c_int),
                ("bg_len", ctypes.c_int), ("scale_modifier", ctypes.c_float), ("specular_threshold", ctypes.c_float)]


def trace_forward(ray_o, ray_d, means3D, scales, rotations, opacities, *, shs=None, colors_precomp=None, others=None,
                  sh_degree=0, bg=None, max_trace_depth=0, specular_threshold=0.0, start_from_first=True, scale_modifier=1.0):
    L = lib()
    ray_o = _f32(ray_o).reshape(-1, 3); ray_d = _f32(ray_d).reshape(-1, 3)
    means3D = _f32(means3D); scales = _f32(scales); rotations = _f32(rotations); opacities = _f32(opacities).reshape(-1)
    shs = _f32(shs); colors_precomp = _f32(colors_precomp); others = _f32(others)
    P, R = means3D.shape[0], ray_o.shape[0]
    M = 0 if shs is None else shs.shape[1]
    bg = np.zeros(3, np.float32) if bg is None else _f32(bg).reshape(-1)
    cfg = _TCfg(P, R, int(sh_degree), M, int(max_trace_depth), (2 if start_from_first == 2 else int(bool(start_from_first))), int(others is not None),
                len(bg), float(scale_modifier), float(specular_threshold))
    ND = max_trace_depth + 1
    rgb = np.zeros((R, 3), np.float32); dpt = np.zeros(R, np.float32); acc = np.zeros(R, np.float32)
    norm = np.zeros((R, 3), np.float32); dist = np.zeros(R, np.float32); aux = np.zeros((R, 2), np.float32)
    mid = np.zeros((R, 16 * ND), np.float32); wet = np.zeros(P, np.float64)
    final_T = np.zeros(R, np.float32); nhits = np.zeros(R, np.int32)
    # the distortion shadow (trc_set_dist_shadow): stage 0's distortion in double from the float code's own alphas / distances, and the a-priori
    # fp32 rounding bound of its moment form -- what a HIP value is asserted against
    dist64 = np.zeros(R, np.float64); dist_bound = np.zeros(R, np.float64)
    L.trc_set_dist_shadow(_p(dist64), _p(dist_bound))
    L.trc_forward(ctypes.byref(cfg), _p(ray_o), _p(ray_d), _p(means3D), _p(scales), _p(rotations), _p(opacities), _p(shs),
                  _p(colors_precomp), _p(others), _p(bg), _p(rgb), _p(dpt), _p(acc), _p(norm), _p(dist), _p(aux), _p(mid),
                  _p(wet), _p(final_T), _p(nhits))
    return dict(cfg=cfg, rgb=rgb, dpt=dpt, acc=acc, norm=norm, dist=dist, aux=aux, mid=mid, wet=wet, final_T=final_T,
                nhits=nhits, bg=bg, dist64=dist64, dist_bound=dist_bound,
                inputs=dict(ray_o=ray_o, ray_d=ray_d, means3D=means3D, scales=scales, rotations=rotations,
                            opacities=opacities, shs=shs, colors_precomp=colors_precomp, others=others))


def trace_audit(ray_o, ray_d, means3D, scales, rotations, opacities, *, others=None, start_from_first=True, tmin=None,
                bounce_thr=None, scale_modifier=1.0, lcap=None, shs=None, sh_degree=0):
    """Fragility audit of one trace stage (trc_audit): dict(fragile (R,) bool, ids / tbits (R,lcap) front-to-back composited surfel ids and
    float bits of their hit distance, nhit (R,)).  tmin overrides the start_from_first rule (bounce stages: 1e-3); bounce_thr adds the bounce decisions."""
    L = lib()
    ray_o = _f32(ray_o).reshape(-1, 3); ray_d = _f32(ray_d).reshape(-1, 3)
    means3D = _f32(means3D); scales = _f32(scales); rotations = _f32(rotations); opacities = _f32(opacities).reshape(-1)
    others = _f32(others); shs = _f32(shs)          # shs (P,M,3) + sh_degree: audit the colour clamp as well (see trc_audit)
    P, R = means3D.shape[0], ray_o.shape[0]
    cfg = _TCfg(P, R, int(sh_degree) if shs is not None else 0, int(shs.shape[1]) if shs is not None else 0, 0, int(bool(start_from_first)),
                int(others is not None), 3, float(scale_modifier), 0.0)
    lcap = int(lcap or max(P, 1))
    fragile = np.zeros(max(R, 1), np.uint8); ids = np.full((max(R, 1), lcap), -1, np.int32); nhit = np.zeros(max(R, 1), np.int32)
    tbits = np.zeros((max(R, 1), lcap), np.uint32)
    L.trc_audit(ctypes.byref(cfg), _p(ray_o), _p(ray_d), _p(means3D), _p(scales), _p(rotations), _p(opacities), _p(others), _p(shs),
                ctypes.c_float(-1.0 if tmin is None else float(tmin)), ctypes.c_float(-1.0 if bounce_thr is None else float(bounce_thr)),
                _p(fragile), _p(ids), _p(tbits), ctypes.c_int(lcap), _p(nhit))
    return dict(fragile=fragile[:R].astype(bool), ids=ids[:R], tbits=tbits[:R], nhit=nhit[:R])


def trace_backward(fwd, dL_drgb, dL_ddpt, dL_dacc, dL_dnorm, dL_daux, want_cond=False):
    """want_cond: additionally `cond` = the noise scale of every gradient element (see trc_backward's comment), same keys."""
    L = lib()
    cfg = fwd["cfg"]; i = fwd["inputs"]
    P, R, M = cfg.P, cfg.R, cfg.M
    g = [_f32(x) for x in (dL_drgb, dL_ddpt, dL_dacc, dL_dnorm, dL_daux)]
    dmeans = np.zeros((P, 3)); dscales = np.zeros((P, 2)); drots = np.zeros((P, 4)); dopac = np.zeros(P)
    dshs = np.zeros((P, max(M, 1), 3)); dcolors = np.zeros((P, 3)); dothers = np.zeros((P, 2))
    dro = np.zeros((R, 3)); drd = np.zeros((R, 3))
    cnd = [np.zeros_like(x) for x in (dmeans, dscales, drots, dopac, (dshs if M > 0 else dcolors), dothers, dro, drd)]
    unc = [np.zeros_like(x) for x in cnd]
    L.trc_backward(ctypes.byref(cfg), _p(i["ray_o"]), _p(i["ray_d"]), _p(i["means3D"]), _p(i["scales"]), _p(i["rotations"]),
                   _p(i["opacities"]), _p(i["shs"]), _p(i["colors_precomp"]), _p(i["others"]), _p(fwd["bg"]), _p(g[0]), _p(g[1]),
                   _p(g[2]), _p(g[3]), _p(g[4]), _p(dmeans), _p(dscales), _p(drots), _p(dopac), _p(dshs), _p(dcolors), _p(dothers),
                   _p(dro), _p(drd), *[_p(x) for x in ((cnd + unc) if want_cond else [None] * 16)])
    out = dict(dmeans3D=dmeans, dscales=dscales, drots=drots, dopacities=dopac, dshs=dshs if M > 0 else None,
               dcolors=dcolors if M == 0 else None, dothers=dothers if i["others"] is not None else None,
               dray_o=dro, dray_d=drd)
    if want_cond:
        out["cond"] = dict(dmeans3D=cnd[0], dscales=cnd[1], drots=cnd[2], dopacities=cnd[3], dshs=cnd[4] if M > 0 else None,
                           dcolors=cnd[4] if M == 0 else None, dothers=cnd[5] if i["others"] is not None else None, dray_o=cnd[6], dray_d=cnd[7])
        out["unc"] = dict(dmeans3D=unc[0], dscales=unc[1], drots=unc[2], dopacities=unc[3], dshs=unc[4] if M > 0 else None,
                          dcolors=unc[4] if M == 0 else None, dothers=unc[5] if i["others"] is not None else None, dray_o=unc[6], dray_d=unc[7])
    return out
