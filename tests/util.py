"""Shared helpers for the test-suite: small seeded scenes in the layout the reference call sites use."""
import os

import numpy as np
import torch

from envgs_amd import synth


def small_scene(P=400, H=64, W=80, seed=0, view=1, C=3, sh=True, spread=1.0, scale_mul=4.0):
    """A small scene whose surfels are big enough on a tiny image to overlap heavily."""
    g = synth.base_gaussians(P, seed=seed)
    g["means3D"] = g["means3D"] * spread
    g["scales"] = g["scales"] * scale_mul
    cam = synth.orbit_camera(view, H=H, W=W, fx=1111.1 * W / 800.0)
    if not sh:
        gen = torch.Generator().manual_seed(seed + 7)
        g["colors_precomp"] = torch.rand(P, C, generator=gen)
    return g, cam


def cam_args(cam):
    return dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
                W=cam.image_width, H=cam.image_height)


def npy(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


def rel_err(a, b, eps=1e-8):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + eps))


def assert_close_frac(a, b, tol, max_bad_frac=1e-4, flip_bound=None, what=""):
    """|a-b| <= tol * max|b| everywhere except a vanishing fraction of elements.

    The compositing has hard thresholds (alpha >= 1/255, T*(1-alpha) < 1e-4, T > 0.5, rho3d <= rho2d).  At the
    BASELINE size ~3e8 (pixel, splat) evaluations are made, and a handful land inside the fp noise between the
    GPU's v_exp_f32 / v_rcp_f32 and glibc's expf / IEEE division, flipping one splat in one pixel (an O(1/255)
    change).  Those are not kernel errors; they are bounded in count (max_bad_frac) and size (flip_bound)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-12
    err = np.abs(a - b) / scale
    bad = err > tol
    assert bad.mean() <= max_bad_frac, "%s: %.3g of elements beyond %.1e (max %.3g)" % (what, bad.mean(), tol, err.max())
    if flip_bound is not None:
        assert err.max() <= flip_bound, "%s: max rel err %.3g > flip bound %.3g" % (what, err.max(), flip_bound)


# ---------------------------------------------------------------------------------------------------------------------
# The 1e-4 contract (BASELINE.json north_star: "within 1e-4 rel on rendered pixels and gradients").
#
# Error of element i:   err_i = |a_i - b_i| / (|b_i| + floor_i)
#   values   : floor = mean |b| over the compared elements -- an elementwise relative error with an absolute floor at the tensor's own
#              typical magnitude;
#   gradients: floor_i = KAPPA * cond_i, where cond_i = sum |term| is the magnitude of what element i is a sum of, accumulated by the
#              oracle itself next to the gradient (oracle/raster.py, oracle/trace.py: want_cond).  A gradient element is a sum of
#              thousands of signed per-pixel / per-hit terms that largely cancel; any fp32 implementation carries a few ulp of EACH TERM
#              into the sum, so the element's error is judged against what was summed, not against the remainder:
#              |a - b| <= 1e-4 |b| + 2e-6 sum|term|   (KAPPA = 0.02; 2e-6 ~ 16 fp32 ulp per term).
#              Rasterizer gradients additionally get the oracle's MEASURED fp32 uncertainty unc_i = sum |term_f32 - term_f64|
#              (oracle/surfel_raster_oracle.c:orc_render_bwd_unc: the ray/splat intersection is ill-conditioned for edge-on splats and
#              loses ~2 digits to px*Tw - Tu at 800 px, in any fp32 implementation; the tracer's k-th blend weight carries the rounding
#              of k transmittance factors):  ... + K_UNC * unc_i, K_UNC = 16 (unc is the REALISED error of one fp32 evaluation; another
#              evaluation's error is of the same size in expectation, not term by term).  Measured on the 300 k / 800x800 case:
#              19 305 of 900 000 position-gradient elements are beyond 1e-4 without the unc term, 393 with K_UNC = 1, 33 with 4, 6 with 16.
#              A last floor of 1e-2 * mean|b| (absolute error below 1e-6 of the tensor's typical magnitude) covers the rounding of the
#              per-surfel chain R8 itself (e.g. an SH basis function near one of its zeros), which neither cond nor unc sees.
# NO element may exceed the tolerance: threshold flips are not absorbed here, they are separated beforehand by the oracle's audit (fragile pixels / rays are
# excluded from the comparison -- and counted).  Every comparison is recorded and printed at the end of the pytest run
# (tests/conftest.py), so the measured errors are part of the GPU test log.
TOL = 1e-4
KAPPA = 0.02
K_UNC = 1.0     # rounds 1-3: 16 (plus, at full size, a tail allowance); round 4: 4.  Round 5: 1 everywhere, and 0 (NO measured-uncertainty term: the plain
                # 1e-4 |b| + 2e-6 sum|term| bound) on the full-size raster tensors (k_unc=0.0 at those call sites) -- what profiles/r04_parity_errors.txt
                # measured for every comparison of the suite.  The one exception is the test whose `unc` is an a-priori ulp bound, not a realised error
                # (test_raster_parity.py: sparse_distortion_gradient, k_unc=16).
FRAGILE_PX_MAX = 2e-3       # stated bounds on what the oracle's audit may exclude from a comparison (fail, do not only count): pixels whose
FRAGILE_RAYS_MAX = 4e-2     # outcome hangs on a threshold inside fp32 noise (measured: <= 7.2e-4 of the pixels, <= 3.3e-2 of the rays of the deep-list cases;
                            # round 6: 5e-2 -> 4e-2 = the measured maximum + margin, and what the excluded rays do is asserted too:
                            # tests/test_trace_parity.py::test_fragile_rays_differ_from_the_oracle_by_threshold_hits_only)
TIMINGS = []                # device times recorded by tests (printed in the summary; never asserted under -m gpu)
ERROR_TABLE = []


def floor_rel_err(a, b, floor=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if b.size == 0:
        return np.zeros(0), 0.0
    if floor is None:
        floor = float(np.abs(b).mean())
    floor = max(floor, 1e-30)
    return np.abs(a - b) / (np.abs(b) + floor), floor


def check_close(test, name, a, b, tol=TOL, keep=None, floor=None, excluded=0, cond=None, unc=None, k_unc=None):
    """Assert the contract on (a, b) restricted to `keep` (boolean mask broadcastable to the leading dims, or None); record the result.
    cond / unc: per-element sum |term| and measured fp32 uncertainty from the oracle (gradients), see the comment above.
    k_unc: multiple of `unc` in the floor (default K_UNC; the one test whose `unc` is an a-priori ulp bound instead of a realised error says 16).
    (Rounds 2-3 had a `tail=` allowance for the full-size raster comparisons; gone with the canonical operation order, round 4.)"""
    k_assert = K_UNC if k_unc is None else float(k_unc)
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if cond is not None:
        cond = np.asarray(cond, np.float64).reshape(b.shape)
    if unc is not None:
        unc = np.asarray(unc, np.float64).reshape(b.shape)
    if keep is not None:
        keep = np.asarray(keep, bool)
        a = a[keep]; b = b[keep]
        if cond is not None: cond = cond[keep]
        if unc is not None: unc = unc[keep]
    if cond is not None:
        fl_i = 0.01 * float(np.abs(b).mean() if b.size else 0.0) + KAPPA * cond + ((k_assert / tol) * unc if unc is not None else 0.0)
        err = np.abs(a - b) / (np.abs(b) + fl_i + 1e-300)
        fl = float(np.mean(fl_i)) if cond.size else 0.0
    else:
        err, fl = floor_rel_err(a, b, floor)
    mx = float(err.max()) if err.size else 0.0
    plain = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300)) if err.size else 0.0      # the plain reading: max|a-b| / max|b| over the tensor
    note = "" if mx <= tol else "%d elements beyond tol" % int((err > tol).sum())
    if cond is not None and unc is not None and err.size:
        # sensitivity of the verdict to K_UNC: the same comparison with the measured-uncertainty term at other multiples,
        # down to NONE (K_UNC=0: the plain 1e-4 |b| + 2e-6 sum|term| bound)
        sens = ["asserted at K_UNC=%g" % k_assert]
        for k in [kk for kk in (4.0, 1.0, 0.0) if kk != k_assert]:
            fl_k = 0.01 * float(np.abs(b).mean()) + KAPPA * cond + (k / tol) * unc
            e_k = np.abs(a - b) / (np.abs(b) + fl_k + 1e-300)
            sens.append("K_UNC=%g: max %.2e, %d beyond" % (k, float(e_k.max()), int((e_k > tol).sum())))
        note = (note + "  " if note else "") + "[" + "; ".join(sens) + "]"
    ERROR_TABLE.append(dict(test=test, tensor=name, max_err=mx, plain=plain, tol=tol, n=int(err.size), excluded=int(excluded), floor=fl, note=note))
    if os.environ.get("ENVGS_PARITY_COLLECT"):        # diagnosis runs: record everything, assert nothing -- and the session FAILS at the end (conftest.py)
        return mx
    assert mx <= tol, "%s / %s: max elementwise error %.3g > %.1e (floor %.3g, %d elements, %d excluded as fragile)" % (test, name, mx, tol, fl, err.size, excluded)
    return mx


def record(test, name, value, note=""):
    ERROR_TABLE.append(dict(test=test, tensor=name, max_err=float(value), tol=None, n=0, excluded=0, floor=0.0, note=note))


def record_fragile(test, name, mask, bound, note=""):
    """Record the fraction of pixels / rays the oracle's audit excludes from a comparison and FAIL when it exceeds the stated bound."""
    mask = np.asarray(mask)
    frac = float(mask.mean()) if mask.size else 0.0
    ERROR_TABLE.append(dict(test=test, tensor=name, max_err=frac, tol=None, n=0, excluded=int(mask.sum()), floor=0.0,
                            note=(note + " " if note else "") + "(%d of %d; bound %.0e)" % (int(mask.sum()), mask.size, bound)))
    assert frac <= bound, "%s / %s: %.3g of the elements are fragile (bound %.1e): the comparison would not mean anything" % (test, name, frac, bound)
    return frac
