"""fp16 FEATURE storage (BASELINE configs[4]; cfg.feature_f16): the kernels read shs / colors_precomp as half -- either because the caller passes half
tensors (its gradient then comes back in half, autograd's rule) or because envgs_amd.set_feature_storage("f16") is on (fp32 in, fp32 gradients out) -- and convert on
load, arithmetic / accumulation / gradient buffers stay fp32.  Converting half -> float is exact, so the fp16-storage path must reproduce the
fp32 path run on the SAME (fp16-rounded) values to fp32 rounding -- a much sharper statement than an "fp16 tolerance" -- and, against the
original fp32 features, stay within the quantisation error of the features themselves."""
import numpy as np
import pytest
import torch

from envgs_amd import synth
from tests.test_oracle_trace import trace_scene
from tests.util import small_scene, check_close, record

pytestmark = pytest.mark.gpu


def _raster(mod, C, g, cam, deg, feats, dev, seed=5):
    from tests.test_raster_parity import _settings
    bg = torch.tensor([0.2, 0.5, 0.9])
    st = _settings(mod, cam, bg, deg, dev)
    L = {k: g[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations")}
    f = feats.to(dev).requires_grad_(True)
    m2 = torch.zeros_like(L["means3D"], requires_grad=True)
    kw = dict(shs=f, colors_precomp=None) if feats.dim() == 3 else dict(shs=None, colors_precomp=f)
    color, radii, allmap, weight = mod.GaussianRasterizer(raster_settings=st)(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], scales=L["scales"],
                                                                                rotations=L["rotations"], cov3D_precomp=None, **kw)
    H, W = cam.image_height, cam.image_width
    gen = torch.Generator().manual_seed(seed)
    dcol = (torch.randn(C, H, W, generator=gen) / (H * W)).to(dev); dall = (torch.randn(7, H, W, generator=gen) / (H * W)).to(dev)
    ((color * dcol).sum() + (allmap * dall).sum()).backward()
    torch.cuda.synchronize()
    return color.detach(), allmap.detach(), {k: v.grad for k, v in L.items()}, f.grad


@pytest.mark.parametrize("C,sh", [(3, True), (5, False), (7, False)])
def test_raster_fp16_feature_storage(C, sh):
    from tests.test_raster_parity import _mod_for
    dev = torch.device("cuda:0")
    g, cam = small_scene(P=600, H=64, W=80, seed=3, C=C, sh=sh)
    mod = _mod_for(C)
    feats = g["shs"] if sh else g["colors_precomp"]
    h = feats.half()
    c16, a16, g16, f16 = _raster(mod, C, g, cam, 3, h, dev)
    c32, a32, g32, f32 = _raster(mod, C, g, cam, 3, h.float(), dev)                 # fp32 path on the fp16-rounded values
    assert f16.dtype == torch.float16 and f32.dtype == torch.float32               # the gradient comes back in the caller's dtype
    t = "fp16_storage_raster_C%d" % C
    assert torch.equal(c16, c32) and torch.equal(a16, a32)                         # identical arithmetic after an exact conversion
    for k in g32:
        check_close(t, "d" + k, g16[k].cpu().numpy(), g32[k].cpu().numpy(), tol=1e-5)
    sel = f32.abs() > 6.2e-5                                                       # (below that the half gradient is subnormal)
    check_close(t, "dfeat(half)", f16.float()[sel].cpu().numpy(), f32[sel].cpu().numpy(), tol=1e-3)
    # against the ORIGINAL fp32 features: the image moves by the quantisation of the colours only
    c_full, *_ = _raster(mod, C, g, cam, 3, feats, dev)
    record(t, "image_shift_vs_fp32_features", float((c16 - c_full).abs().max()))
    assert float((c16 - c_full).abs().max()) < 4e-3
    psnr = -10.0 * float(torch.log10(((c16 - c_full) ** 2).mean()))               # "PSNR delta vs fp32"
    record(t, "psnr_vs_fp32_features_dB", psnr)
    assert psnr > 60.0


@pytest.mark.parametrize("sparse", ["off", "on"])
@pytest.mark.parametrize("use_sh", [True, False])
def test_tracer_fp16_feature_storage(use_sh, sparse, request):
    """(sparse = "on": the same with the hits of sparse entries differentiated one lane per hit -- sparse_hits_bwd reads the half features too.)"""
    import diff_surfel_tracing as mod
    from envgs_amd import tracing
    from tests.test_trace_parity import _settings
    old_mode = tracing.SPARSE["mode"]
    tracing.SPARSE["mode"] = sparse
    request.addfinalizer(lambda: tracing.SPARSE.__setitem__("mode", old_mode))
    dev = torch.device("cuda:0")
    g, ro, rd = trace_scene(P=300, R=640, seed=8, camera=False)
    bg = torch.tensor([0.3, 0.1, 0.7])
    feats = g["shs"] if use_sh else g["colors_precomp"]
    out = {}
    for name, f in (("h", feats.half()), ("f", feats.half().float())):
        L = {k: g[k].to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities")}
        fd = f.to(dev).requires_grad_(True)
        o = ro.to(dev).requires_grad_(True); d = rd.to(dev).requires_grad_(True)
        v, fc = synth.get_disks(L["means3D"].detach(), L["scales"].detach(), L["rotations"].detach())
        tr = mod.SurfelTracer(); tr.build_acceleration_structure(v, fc, rebuild=True)
        outs = tr(o, d, v, means3D=L["means3D"], grads3D=None, shs=fd if use_sh else None, colors_precomp=None if use_sh else fd, others_precomp=None,
                  opacities=L["opacities"], scales=L["scales"], rotations=L["rotations"], cov3D_precomp=None, tracer_settings=_settings(mod, bg, 3, dev),
                  start_from_first=False)
        (outs[0] * torch.linspace(0.5, 1.5, 3, device=dev)).sum().backward()
        torch.cuda.synchronize()
        out[name] = (outs[0].detach(), {k: x.grad for k, x in L.items()}, fd.grad, o.grad, d.grad)
        assert (tracing.last_trace_counts()["sparse_hits"] > 0) == (sparse == "on")
    t = "fp16_storage_tracer_%s%s" % ("sh" if use_sh else "rgb", "_sparse" if sparse == "on" else "")
    assert out["h"][2].dtype == torch.float16
    out["full"] = None
    assert torch.equal(out["h"][0], out["f"][0])
    for k in out["f"][1]:
        check_close(t, "d" + k, out["h"][1][k].cpu().numpy(), out["f"][1][k].cpu().numpy(), tol=1e-5)
    check_close(t, "dray_d", out["h"][4].cpu().numpy(), out["f"][4].cpu().numpy(), tol=1e-5)
    sel = out["f"][2].abs() > 6.2e-5
    check_close(t, "dfeat(half)", out["h"][2].float()[sel].cpu().numpy(), out["f"][2][sel].cpu().numpy(), tol=1e-3)


class _F16Storage:
    """envgs_amd.set_feature_storage("f16") for the duration of a block: fp32 feature tensors in, half copies inside the nodes, fp32 gradients out."""
    def __enter__(self):
        import envgs_amd
        envgs_amd.set_feature_storage("f16")
    def __exit__(self, *a):
        import envgs_amd
        envgs_amd.set_feature_storage("f32")


@pytest.mark.parametrize("C,sh", [(3, True), (7, False)])
def test_raster_fp16_storage_keeps_fp32_gradients(C, sh):
    """The storage switch (configs[4]'s training form): the forward is the half-input path bit for bit, and the feature gradient -- which a half
    INPUT gets back rounded to fp16, i.e. flushed to zero below 6e-8 -- comes back in fp32 and equals the fp32 path's on EVERY element."""
    from tests.test_raster_parity import _mod_for
    dev = torch.device("cuda:0")
    g, cam = small_scene(P=600, H=64, W=80, seed=3, C=C, sh=sh)
    mod = _mod_for(C)
    feats = (g["shs"] if sh else g["colors_precomp"]).half().float()              # representable in half: storing them loses nothing
    c32, a32, g32, f32 = _raster(mod, C, g, cam, 3, feats, dev)
    ch, ah, gh, fh = _raster(mod, C, g, cam, 3, feats.half(), dev)
    with _F16Storage():
        cs, as_, gs, fs = _raster(mod, C, g, cam, 3, feats, dev)
    t = "fp16_storage_switch_raster_C%d" % C
    assert fs.dtype == torch.float32 and torch.equal(cs, ch) and torch.equal(as_, ah) and torch.equal(cs, c32)
    check_close(t, "dfeat", fs.cpu().numpy(), f32.cpu().numpy(), tol=1e-5)        # all elements, whatever their magnitude
    lost = float(((fh.float() == 0) & (f32 != 0)).float().mean())
    record(t, "elements_a_half_gradient_flushes_to_zero", lost)
    for k in g32:
        check_close(t, "d" + k, gs[k].cpu().numpy(), g32[k].cpu().numpy(), tol=1e-5)


def test_tracer_fp16_storage_against_the_oracle_on_the_rounded_features(request):
    """fp16 SH storage against the ORACLE run on the half-rounded features (not HIP against HIP): hit lists, values, every gradient at 1e-4."""
    from tests.test_trace_parity import _parity
    g, ro, rd = trace_scene(P=2000, R=1024, seed=7, camera=False)
    g["scales"] = g["scales"] * 0.35
    g["shs"] = g["shs"].half().float()
    with _F16Storage():
        res = _parity("fp16_storage_tracer_vs_oracle", g, ro, rd, torch.tensor([0.3, 0.1, 0.7]), 2, True, False)
    assert res["ref"]["nhits"].mean() > 1
