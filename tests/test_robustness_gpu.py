"""GPU: degenerate and non-finite inputs at the package boundary -- the counterpart of the reference's crash-repro scripts
(tests/cuda_illegal_memory_access_tests.py:8-33 feeds random non-PSD covariances to its rasterizer "to catch illegal-memory crashes"; SURVEY.md
section 4).  A training run produces such values now and then (a scale that underflows, a quaternion that collapses, an opacity logit that
overflows, a reflected ray built from a zero normal); the extensions must neither crash nor hang nor let ONE bad surfel / ray poison what the
others contribute.  Measured behaviour (scratch/robust_probe.py), asserted here:

  rasterizer  surfels with NaN / infinite means or scales, a zero quaternion, NaN or negative opacity, or behind the camera are CULLED: the image
              is bit-identical to the render of the remaining surfels and every gradient is finite (zero for the culled ones);
              zero scales, huge scales, opacities above 1: defined (finite) results;
  tracer      the same surfel corruptions: finite outputs and gradients; rays with a NaN / zero direction or an infinite origin composite nothing
              and leave every OTHER ray's result bit-identical and every parameter gradient finite (round 5: a directionless ray used to put a NaN
              row into the MFMA operand that sums its batch's colour gradients -> NaN dL/dshs for every surfel the other 63 rays blended; a zero
              quaternion used to give 0 * inf in its own rotation gradient)."""
import pytest
import torch

from envgs_amd import synth

pytestmark = pytest.mark.gpu

H = W = 96
BAD = slice(0, 300)
NAN, INF = float("nan"), float("inf")


def _raster(gd, dev, cam):
    import diff_surfel_rasterization_wet as pkg
    st = pkg.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                           viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=torch.tensor([3], device=dev),
                                           campos=cam.camera_center, prefiltered=False, debug=False)
    L = {k: gd[k].to(dev).clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    m2 = torch.zeros_like(L["means3D"], requires_grad=True)
    color, radii, allmap, weight = pkg.GaussianRasterizer(raster_settings=st)(means3D=L["means3D"], means2D=m2, shs=L["shs"], colors_precomp=None, opacities=L["opacities"],
                                                                             scales=L["scales"], rotations=L["rotations"], cov3D_precomp=None)
    (color.sum() + allmap[:5].sum()).backward()
    torch.cuda.synchronize()
    return color.detach(), radii, allmap.detach(), {k: v.grad for k, v in L.items()}, m2.grad


CULLED = {"nan_means": ("means3D", NAN), "inf_means": ("means3D", INF), "nan_scales": ("scales", NAN), "zero_quaternions": ("rotations", 0.0),
          "nan_opacity": ("opacities", NAN), "negative_opacity": ("opacities", -1.0)}
DEFINED = {"zero_scales": ("scales", 0.0), "huge_scales": ("scales", 1e6), "opacity_5": ("opacities", 5.0)}


@pytest.mark.parametrize("kind", sorted(CULLED) + ["behind_camera"] + sorted(DEFINED))
def test_rasterizer_survives_degenerate_surfels(kind):
    dev = torch.device("cuda:0")
    g = synth.base_gaussians(3000, seed=3); g["scales"] = g["scales"] * 4
    cam = synth.orbit_camera(1, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
    d = {k: v.clone() for k, v in g.items()}
    if kind == "behind_camera":
        d["means3D"][BAD] = torch.tensor([50.0, 50.0, 50.0])
    else:
        key, val = (CULLED.get(kind) or DEFINED[kind])
        d[key][BAD] = val
    color, radii, allmap, grads, m2g = _raster(d, dev, cam)
    assert bool(torch.isfinite(color).all()) and bool(torch.isfinite(allmap).all())
    for k, v in grads.items():
        assert bool(torch.isfinite(v).all()), k
    assert bool(torch.isfinite(m2g).all())
    if kind in DEFINED:
        return
    # culled: nothing of the bad surfels reaches the image -- bit-identical to the render of the others
    sane = {k: v[300:].clone() for k, v in g.items()}
    c2, r2, a2, g2, _ = _raster(sane, dev, cam)
    assert torch.equal(color, c2) and torch.equal(allmap, a2)
    assert torch.equal(radii[300:], r2)
    if kind not in ("nan_opacity", "negative_opacity"):           # (those are dropped by the alpha threshold in the compositing, not by the projection)
        assert int((radii[BAD] > 0).sum()) == 0
    for k in grads:
        sc = float(g2[k].abs().max()) + 1e-30                       # and the others' gradients are what they are without them (R7 sums with float atomics:
        assert float((grads[k][300:] - g2[k]).abs().max()) <= 1e-5 * sc, k      # the order of a sum, not its terms, may differ between two runs)
        assert float(grads[k][BAD].abs().max()) == 0.0, k


def _trace(e, ro, rd, dev):
    import diff_surfel_tracing as tpkg
    ts = tpkg.SurfelTracingSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=torch.eye(4, device=dev),
                                    projmatrix=torch.eye(4, device=dev), sh_degree=torch.tensor([3], device=dev), campos=torch.zeros(3, device=dev), prefiltered=False, debug=False,
                                    max_trace_depth=0, specular_threshold=0.0)
    L = {k: e[k].to(dev).clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    o = ro.to(dev).clone().requires_grad_(True); d = rd.to(dev).clone().requires_grad_(True)
    v, f = synth.get_disks(L["means3D"].detach(), L["scales"].detach(), L["rotations"].detach())
    t = tpkg.SurfelTracer(); t.build_acceleration_structure(v, f, rebuild=True)
    outs = t(o, d, v, means3D=L["means3D"], grads3D=None, shs=L["shs"], colors_precomp=None, others_precomp=None, opacities=L["opacities"], scales=L["scales"],
             rotations=L["rotations"], cov3D_precomp=None, tracer_settings=ts, start_from_first=False)
    (outs[0].sum() + outs[1].sum() + outs[3].sum()).backward()
    torch.cuda.synchronize()
    return [x.detach() for x in outs], {k: x.grad for k, x in L.items()}, o.grad, d.grad


def _rays(R=4096):
    gen = torch.Generator().manual_seed(1)
    return (torch.rand(1, R, 3, generator=gen) * 2 - 1), torch.randn(1, R, 3, generator=gen)


@pytest.mark.parametrize("kind", ["nan_means", "inf_means", "zero_scales", "huge_scales", "nan_scales", "zero_quaternions", "nan_opacity", "negative_opacity"])
def test_tracer_survives_degenerate_surfels(kind):
    dev = torch.device("cuda:0")
    e = synth.env_gaussians(2000, seed=4, bound=12.0)
    key, val = {"nan_means": ("means3D", NAN), "inf_means": ("means3D", INF), "zero_scales": ("scales", 0.0), "huge_scales": ("scales", 1e6), "nan_scales": ("scales", NAN),
                "zero_quaternions": ("rotations", 0.0), "nan_opacity": ("opacities", NAN), "negative_opacity": ("opacities", -1.0)}[kind]
    d = {k: v.clone() for k, v in e.items()}
    d[key][BAD] = val
    ro, rd = _rays()
    outs, grads, go, gd = _trace(d, ro, rd, dev)
    for i in (0, 1, 2, 3, 5, 7):
        assert bool(torch.isfinite(outs[i]).all()), i
    for k, v in grads.items():
        assert bool(torch.isfinite(v).all()), k
    assert bool(torch.isfinite(go).all()) and bool(torch.isfinite(gd).all())
    if kind != "huge_scales":
        assert float(outs[7][BAD].abs().max()) == 0.0               # the bad surfels received no weight ...
        sane_o, _, _, _ = _trace({k: v[300:].clone() for k, v in e.items()}, ro, rd, dev)
        assert torch.equal(outs[0], sane_o[0]) and torch.equal(outs[2], sane_o[2])      # ... and the rays see exactly the other surfels


@pytest.mark.parametrize("kind", ["nan_directions", "zero_directions", "infinite_origins", "nan_origins", "axis_aligned_directions"])
def test_tracer_bad_rays_do_not_poison_the_others(kind):
    dev = torch.device("cuda:0")
    e = synth.env_gaussians(2000, seed=4, bound=12.0)
    ro, rd = _rays()
    ref, gref, _, _ = _trace(e, ro, rd, dev)
    ro2, rd2 = ro.clone(), rd.clone()
    bad = slice(0, 100)
    if kind == "nan_directions": rd2[0, bad] = NAN
    elif kind == "zero_directions": rd2[0, bad] = 0.0
    elif kind == "infinite_origins": ro2[0, bad] = INF
    elif kind == "nan_origins": ro2[0, bad] = NAN
    else: rd2[0, bad] = torch.tensor([1.0, 0.0, 0.0])              # (zero components: 1/d is infinite on two axes; a legitimate ray)
    outs, grads, go, gd = _trace(e, ro2, rd2, dev)
    for i in (0, 1, 2, 3):
        assert torch.equal(outs[i][0, 100:], ref[i][0, 100:]), i    # a ray's result does not depend on the rays it is traced with
        assert bool(torch.isfinite(outs[i]).all()), i
    for k, v in grads.items():
        assert bool(torch.isfinite(v).all()), k                     # no parameter gradient is poisoned by the bad rays
    assert bool(torch.isfinite(go[0, 100:]).all()) and bool(torch.isfinite(gd[0, 100:]).all())
    if kind == "axis_aligned_directions":
        assert float(outs[2][0, bad].mean()) > 0.5                  # ... and an axis-aligned ray is traced like any other
        assert bool(torch.isfinite(go).all()) and bool(torch.isfinite(gd).all())
    else:
        assert float(outs[2][0, bad].abs().max()) == 0.0            # the bad rays composite nothing
