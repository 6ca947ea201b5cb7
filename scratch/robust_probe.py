"""Exploratory (round 5): what do the two packages do with non-finite / degenerate inputs?  (-> tests/test_robustness_gpu.py)"""
import torch, sys
sys.path.insert(0, '.')
from envgs_amd import synth
import diff_surfel_rasterization_wet as pkg, diff_surfel_rasterization_wet_ch05 as pkg5
import diff_surfel_tracing as tpkg
dev = torch.device("cuda:0")
H = W = 96
P = 3000
g = synth.base_gaussians(P, seed=3); g["scales"] = g["scales"] * 4
cam = synth.orbit_camera(1, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
st = pkg.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
      viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=torch.tensor([3], device=dev), campos=cam.camera_center, prefiltered=False, debug=False)
def run(gd, tag):
    L = {k: gd[k].to(dev).clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    m2 = torch.zeros_like(L["means3D"], requires_grad=True)
    try:
        color, radii, allmap, weight = pkg.GaussianRasterizer(raster_settings=st)(means3D=L["means3D"], means2D=m2, shs=L["shs"], colors_precomp=None, opacities=L["opacities"], scales=L["scales"], rotations=L["rotations"], cov3D_precomp=None)
        (color.sum() + allmap[:5].sum()).backward()
        torch.cuda.synchronize()
        print("%-28s img finite %s (nan px %d) radii>0 %d grads finite %s" % (tag, bool(torch.isfinite(color).all()), int((~torch.isfinite(color)).any(0).sum()), int((radii > 0).sum()),
              {k: bool(torch.isfinite(v.grad).all()) for k, v in L.items()}))
        return color.detach()
    except Exception as e:
        print(tag, "EXC", type(e).__name__, str(e)[:100])
ref = run(g, "sane")
bad = slice(0, 300)
for tag, f in (("nan means", lambda d: d["means3D"].__setitem__(bad, float("nan"))), ("inf means", lambda d: d["means3D"].__setitem__(bad, float("inf"))),
               ("zero scales", lambda d: d["scales"].__setitem__(bad, 0.0)), ("huge scales", lambda d: d["scales"].__setitem__(bad, 1e6)), ("nan scales", lambda d: d["scales"].__setitem__(bad, float("nan"))),
               ("zero quats", lambda d: d["rotations"].__setitem__(bad, 0.0)), ("nan opac", lambda d: d["opacities"].__setitem__(bad, float("nan"))), ("neg opac", lambda d: d["opacities"].__setitem__(bad, -1.0)),
               ("opac 5", lambda d: d["opacities"].__setitem__(bad, 5.0)), ("nan shs", lambda d: d["shs"].__setitem__(bad, float("nan"))), ("behind camera", lambda d: d["means3D"].__setitem__(bad, d["means3D"][bad] * 0 + torch.tensor([50., 50., 50.])))):
    d = {k: v.clone() for k, v in g.items()}
    f(d)
    out = run(d, tag)
    if out is not None and ref is not None:
        sane = {k: v[300:].clone() for k, v in g.items()}
        exp = run(sane, "   (sane subset)")
        print("      == sane-subset render:", bool(torch.equal(torch.nan_to_num(out), torch.nan_to_num(exp))), float((torch.nan_to_num(out) - exp).abs().max()))
# tracer
e = synth.env_gaussians(2000, seed=4, bound=12.0)
ts = tpkg.SurfelTracingSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=torch.eye(4, device=dev), projmatrix=torch.eye(4, device=dev),
      sh_degree=torch.tensor([3], device=dev), campos=torch.zeros(3, device=dev), prefiltered=False, debug=False, max_trace_depth=0, specular_threshold=0.0)
R = 4096
gen = torch.Generator().manual_seed(1)
ro0 = (torch.rand(1, R, 3, generator=gen) * 2 - 1); rd0 = torch.randn(1, R, 3, generator=gen)
def trace(ed, ro, rd, tag):
    L = {k: ed[k].to(dev).clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    o = ro.to(dev).clone().requires_grad_(True); d = rd.to(dev).clone().requires_grad_(True)
    try:
        v, f = synth.get_disks(L["means3D"].detach(), L["scales"].detach(), L["rotations"].detach())
        t = tpkg.SurfelTracer(); t.build_acceleration_structure(v, f, rebuild=True)
        outs = t(o, d, v, means3D=L["means3D"], grads3D=None, shs=L["shs"], colors_precomp=None, others_precomp=None, opacities=L["opacities"], scales=L["scales"], rotations=L["rotations"], cov3D_precomp=None, tracer_settings=ts, start_from_first=False)
        (outs[0].sum() + outs[1].sum()).backward()
        torch.cuda.synchronize()
        print("%-28s rgb finite rays %d/%d acc mean %.3f grads finite %s ray grads finite %s" % (tag, int(torch.isfinite(outs[0]).all(-1).sum()), R, float(torch.nan_to_num(outs[2]).mean()),
              {k: bool(torch.isfinite(x.grad).all()) for k, x in L.items()}, bool(torch.isfinite(o.grad).all() and torch.isfinite(d.grad).all())))
        return outs[0].detach()
    except Exception as ex:
        print(tag, "EXC", type(ex).__name__, str(ex)[:120])
tref = trace(e, ro0, rd0, "tracer sane")
for tag, f in (("nan means", lambda d: d["means3D"].__setitem__(bad, float("nan"))), ("inf means", lambda d: d["means3D"].__setitem__(bad, float("inf"))), ("zero scales", lambda d: d["scales"].__setitem__(bad, 0.0)),
               ("huge scales", lambda d: d["scales"].__setitem__(bad, 1e6)), ("zero quats", lambda d: d["rotations"].__setitem__(bad, 0.0)), ("nan opac", lambda d: d["opacities"].__setitem__(bad, float("nan")))):
    d = {k: v.clone() for k, v in e.items()}
    f(d)
    trace(d, ro0, rd0, "tracer " + tag)
for tag, fo, fd in (("nan dirs", None, lambda x: x.__setitem__((0, slice(0, 100)), float("nan"))), ("zero dirs", None, lambda x: x.__setitem__((0, slice(0, 100)), 0.0)), ("inf origins", lambda x: x.__setitem__((0, slice(0, 100)), float("inf")), None),
                    ("axis-aligned dirs", None, lambda x: x.__setitem__((0, slice(0, 100)), torch.tensor([1.0, 0.0, 0.0])))):
    ro, rd = ro0.clone(), rd0.clone()
    if fo: fo(ro)
    if fd: fd(rd)
    out = trace(e, ro, rd, "tracer " + tag)
    if out is not None:
        print("      other rays unchanged:", bool(torch.equal(out[0, 100:], tref[0, 100:])))
