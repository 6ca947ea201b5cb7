"""Stage-wise parity of COMPOSED pipelines (the bounce chain of SurfelTracer, the full EnvGS step) at the 1e-4 contract.

A composed chain has no per-element noise scale of its own: stage k+1 sees rays that stage k produced, so two implementations that differ by
fp32 rounding in stage k trace DIFFERENT rays in stage k+1, and a whole-chain comparison can only be made with a widened bound (round 2 used
3e-3).  Instead every link is checked on bit-identical inputs:

  * every traced call the HIP chain makes is tapped (TraceTap): its ray tensors, its outputs, the upstream gradients autograd delivers to
    those outputs and the ray gradients the kernel returns are recorded;
  * the oracle is run on exactly those rays (the HIP chain's own fp32 rays) with exactly those upstream gradients, and outputs, ray gradients
    and parameter gradients are compared with the elementwise 1e-4 contract and the oracle's cond / unc noise floors (tests/util.py) -- the
    parameter gradients as the sum over the chain's calls, which is what the leaves receive;
  * the torch glue between the calls (reflected rays, blends) is checked against the same expressions evaluated in float64 on the recorded
    tensors (the callers of this module do that part, the expressions being theirs).

Test infrastructure only."""
import numpy as np
import torch

from tests.util import check_close, record


class TraceTap:
    """Records every _TraceSurfels.apply call made inside the `with` block.  Each record: o_in / d_in (the tensors handed to the kernel; their
    .grad after backward is exactly the kernel's ray gradient), outs (the 8 outputs), up (upstream gradients of outputs 0..5, None = unused),
    sff (start_from_first as passed: False / True / 2), settings."""

    def __enter__(self):
        from envgs_amd import tracing
        self.tracing = tracing
        self.calls = []
        self.orig = tracing._TraceSurfels.apply
        tap = self

        def tapped(ray_o, ray_d, *rest):
            need = torch.is_grad_enabled() and (ray_o.requires_grad or ray_d.requires_grad)
            o_in = ray_o.clone() if need else ray_o
            d_in = ray_d.clone() if need else ray_d
            if need:
                o_in.retain_grad(); d_in.retain_grad()
            outs = tap.orig(o_in, d_in, *rest)
            # rest = (v, means3D, grads3D, shs, colors, others, opacities, scales, rotations, cov3D, settings, start_from_first, nodes[, caps])
            rec = dict(o_in=o_in, d_in=d_in, outs=outs, up=[None] * 6, settings=rest[10], sff=rest[11])
            for i in (0, 1, 2, 3, 5):
                if outs[i].requires_grad:
                    outs[i].register_hook(lambda g, i=i, rec=rec: rec["up"].__setitem__(i, None if g is None else g.detach().clone()))
            tap.calls.append(rec)
            return outs

        tracing._TraceSurfels.apply = staticmethod(tapped)
        return self

    def __exit__(self, *a):
        del self.tracing._TraceSurfels.apply          # back to the inherited classmethod


class RasterTap:
    """Records every rasterize_backward call (the function the autograd node of the three raster packages calls): the saved forward state,
    the upstream gradients (dL_dcolor (C,H,W), dL_dallmap (7,H,W)) and the gradients the kernels returned."""

    def __enter__(self):
        from envgs_amd import raster
        self.raster = raster
        self.calls = []
        self.orig = raster.rasterize_backward
        tap = self

        def tapped(saved, dL_dcolor, dL_dallmap):
            g = tap.orig(saved, dL_dcolor, dL_dallmap)
            tap.calls.append(dict(saved=saved, dL_dcolor=dL_dcolor.detach().clone(), dL_dallmap=dL_dallmap.detach().clone(), grads=g))
            return g

        raster.rasterize_backward = tapped
        return self

    def __exit__(self, *a):
        self.raster.rasterize_backward = self.orig


def _n(t):
    return t.detach().cpu().numpy()


def oracle_trace_call(test, name, rec, g, bg, deg, use_sh=True, others=True, check_rays=True, nfr=0, keep=None):
    """One tapped call against the oracle on the SAME rays and the SAME upstream gradients.  Compares the five differentiable outputs and the
    ray gradients here; returns the oracle's backward (with cond / unc) so that the caller can sum the parameter gradients over the calls.
    keep: (R,) bool -- the per-ray comparisons are restricted to these rays (the caller zeroed the upstream gradient of the others)."""
    from oracle import trace as otr
    o = _n(rec["o_in"]).reshape(-1, 3); d = _n(rec["d_in"]).reshape(-1, 3)
    R = o.shape[0]
    ckw = dict(shs=g["shs"].numpy(), sh_degree=deg) if use_sh else dict(colors_precomp=g["colors_precomp"].numpy())
    ref = otr.trace_forward(o, d, g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), g["opacities"].numpy(),
                            others=(g["others"].numpy() if others else None), bg=np.asarray(bg, np.float32), start_from_first=rec["sff"], **ckw)
    outs = rec["outs"]
    for i, nm, c in ((0, "rgb", 3), (1, "dpt", 1), (2, "acc", 1), (3, "norm", 3), (5, "aux", 2)):
        if nm == "aux" and not others: continue
        check_close(test, "%s.%s" % (name, nm), _n(outs[i]).reshape(R, c), ref[nm].reshape(R, c), excluded=nfr, keep=keep)
    up = []
    for i, c in ((0, 3), (1, 1), (2, 1), (3, 3), (5, 2)):
        u = rec["up"][i]
        up.append(np.zeros((R, c), np.float32) if u is None else _n(u).reshape(R, c).astype(np.float32))
    rb = otr.trace_backward(ref, up[0], up[1][:, 0], up[2][:, 0], up[3], up[4], want_cond=True)
    if check_rays and rec["o_in"].grad is not None:
        check_close(test, "%s.dray_o" % name, _n(rec["o_in"].grad).reshape(R, 3), rb["dray_o"], excluded=nfr, cond=rb["cond"]["dray_o"], unc=rb["unc"]["dray_o"], keep=keep)
        check_close(test, "%s.dray_d" % name, _n(rec["d_in"].grad).reshape(R, 3), rb["dray_d"], excluded=nfr, cond=rb["cond"]["dray_d"], unc=rb["unc"]["dray_d"], keep=keep)
    return ref, rb


PARAM_KEYS = (("means3D", "dmeans3D"), ("scales", "dscales"), ("rotations", "drots"), ("opacities", "dopacities"), ("others", "dothers"),
              ("shs", "dshs"), ("colors_precomp", "dcolors"))


def check_summed_param_grads(test, name, leaves, backs, nfr=0, k_unc_by_key=None):
    """leaves: dict of HIP leaf tensors (their .grad = the sum over the chain's calls); backs: the oracle backward dict of every call.
    k_unc_by_key: {oracle key: multiple of the oracle's measured fp32 uncertainty} for tensors asserted at another multiple than the suite's K_UNC
    (the caller says why); the table's sensitivity columns carry the other multiples either way."""
    for k_hip, k_ref in PARAM_KEYS:
        if k_hip not in leaves or leaves[k_hip].grad is None or backs[0].get(k_ref) is None:
            continue
        want = sum(np.asarray(b[k_ref], np.float64) for b in backs)
        cond = sum(np.asarray(b["cond"][k_ref], np.float64) for b in backs)
        unc = sum(np.asarray(b["unc"][k_ref], np.float64) for b in backs)
        got = _n(leaves[k_hip].grad).astype(np.float64).reshape(want.shape)
        check_close(test, "%s.%s" % (name, k_ref), got, want, excluded=nfr, cond=cond, unc=unc, k_unc=(k_unc_by_key or {}).get(k_ref))


def glue_check(test, name, got, want, floor=None, cond=None):
    """A glue value / gradient (torch autograd in fp32 on the GPU, or the fused HIP glue) against the same expression in float64 on the same
    recorded tensors.  Elements that are not finite on either side (0/0 where a pixel has no coverage: nan_to_num's backward) are left out and
    counted; everything else must be within 1e-4 elementwise."""
    a = _n(got).astype(np.float64)
    b = _n(want).astype(np.float64).reshape(a.shape)
    fin = np.isfinite(a) & np.isfinite(b)
    if not fin.all():
        record(test, name + ".nonfinite", float((~fin).mean()), "(%d elements not finite on one side, left out)" % int((~fin).sum()))
        assert (~fin).mean() < 0.2
    if cond is not None:        # per-element magnitude of what the element is a (cancelling) sum of: floor 2e-6 * cond, like the kernels' gradients
        check_close(test, name, a[fin], b[fin], cond=_n(cond).astype(np.float64).reshape(a.shape)[fin])
    else:
        check_close(test, name, a[fin], b[fin], floor=floor)
