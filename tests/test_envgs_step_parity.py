"""GPU, end to end: one full EnvGS step -- ch05 base raster -> reflect -> env trace -> blend -> backward -- through the caller code
(envgs_amd/envgs_step.py, the re-derivation of EnvGSSampler.forward, pinned against the reference's own render() / render_gaussians() run in
tests/test_caller_contract.py) over the HIP extensions.  This is the test that the two extensions COMPOSE: the tracer's ray gradients must
flow back through the reflected-ray construction into the rasterizer's normal / depth gradients
(easyvolcap/models/samplers/envgs_sampler.py:420-455 with detach=False).

The 1e-4 contract is applied LINK BY LINK (tests/stagewise.py): the raster call and the traced call of the HIP step are each compared with
the oracle on bit-identical inputs -- the HIP step's own colours / reflected rays and the upstream gradients autograd delivered to them --
with the oracle's cond / unc noise floors; the glue between them (SH colours, reflected rays, the specular blend: torch expressions or the
fused HIP glue) is compared with the same expressions in float64 on the recorded tensors.  Round 2 compared the whole chain over the HIP
extensions with the whole chain over the oracle at 3e-3 (the two chains trace reflected rays that differ by fp32 rounding of the base
pass); that comparison stays as a recorded diagnostic."""
import numpy as np
import pytest
import torch

from envgs_amd import envgs_step, synth
from tests import stagewise
from tests.util import check_close, record, record_fragile, floor_rel_err, FRAGILE_RAYS_MAX

pytestmark = pytest.mark.gpu


def _scene(dev, H=48, W=64):
    base = synth.base_gaussians(1500, seed=3)
    base["scales"] = base["scales"] * 5.0
    base["opacities"] = torch.sigmoid(torch.randn(1500, 1, generator=torch.Generator().manual_seed(1)) + 1.5)
    env = synth.env_gaussians(800, seed=4, bound=12.0)
    mv = lambda d: {k: v.to(dev).clone().requires_grad_(True) for k, v in d.items()}
    camd = synth.orbit_camera(1, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
    return mv(base), mv(env), camd


def _upstream(H, W, dev, keep=None):
    gen = torch.Generator().manual_seed(7)
    dcol = (torch.randn(H, W, 3, generator=gen) / (H * W)).to(dev)
    dall = (torch.randn(7, H, W, generator=gen) / (H * W)).to(dev); dall[5:] = 0
    if keep is not None:                                 # upstream gradient only where the oracle's audits call the pixel / its reflected ray determined
        k = keep.to(dev)
        dcol = dcol * k[..., None]; dall = dall * k[None]
    return dcol, dall


def _run(pkg, tpkg, tracer, dev, keep=None, backward=True):
    base, env, cam = _scene(dev)
    rays = synth.get_rays(cam)
    bg = torch.zeros(3, device=dev); env_bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, base, env, bg, env_bg, torch.tensor([2], device=dev))
    g = {}
    if backward:
        dcol, dall = _upstream(cam.image_height, cam.image_width, dev, keep)
        out["base"]["colors"].retain_grad()
        loss = (out["rgb"] * dcol).sum() + (out["base"]["allmap"] * dall).sum()
        loss.backward()
        g = {("base." + k): v.grad for k, v in base.items() if v.grad is not None}
        g.update({("env." + k): v.grad for k, v in env.items() if v.grad is not None})
    return out, g, (base, env, cam, rays)


n = lambda t: t.detach().cpu().numpy()


def _glue_f64(cam, rays, img, allmap, rgb_env):
    """envgs_forward's glue between the two extension calls, float64, on CPU tensors (the expressions of envgs_step.py / envgs_sampler.py:420-474)."""
    V = cam.world_view_transform.detach().cpu().double()
    ray_o, ray_d = rays[0].detach().cpu().double(), rays[1].detach().cpu().double()
    alpha = allmap[1:2]
    normal = (allmap[2:5].permute(1, 2, 0) @ (V[:3, :3].T)).permute(2, 0, 1)
    depth = torch.nan_to_num(allmap[0:1] / alpha, 0, 0)
    nrm = normal.permute(1, 2, 0)
    nrm = nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-8)
    ref_d = ray_d - 2 * (ray_d * nrm).sum(-1, keepdim=True) * nrm
    ref_o = ray_o + ray_d * depth.permute(1, 2, 0)
    spec = img[3:4].permute(1, 2, 0)
    rgb = (1 - spec) * img[:3].permute(1, 2, 0) + spec * rgb_env
    return ref_o, ref_d, rgb


def _colors_f64(cam, base, deg):
    shs_view = base["shs"].transpose(1, 2)
    dir_pp = base["means3D"] - cam.camera_center.detach().cpu().double()[None]
    dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    dir_pp.retain_grad()
    raw = envgs_step.eval_sh(deg, shs_view, dir_pp) + 0.5
    return torch.cat([torch.clamp_min(raw, 0.0), base["specular"], base["roughness"]], dim=-1), raw, dir_pp


@pytest.mark.parametrize("fused_glue", [False, True])
def test_full_envgs_step_link_by_link(fused_glue):
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    from oracle import raster as orc, trace as otr
    test = "envgs_step[%s]" % ("fused" if fused_glue else "torch")
    dev = torch.device("cuda:0")
    deg = 2
    envgs_step.FUSED["on"] = fused_glue              # HIP run: torch glue or the fused HIP glue (envgs_amd.fused)
    try:
        # 1. forward only: what the HIP step hands to the two extensions; the oracle's audits of exactly that (fragile base pixels / reflected rays)
        with torch.no_grad():
            out1, _, (base, env, cam, rays) = _run(pkg, tpkg, tpkg.SurfelTracer(), dev, backward=False)
        H, W = cam.image_height, cam.image_width
        P = base["means3D"].shape[0]
        ref = orc.raster_forward(n(base["means3D"]), n(base["opacities"]), n(cam.world_view_transform), n(cam.full_proj_transform), n(cam.camera_center), W, H,
                                 scales=n(base["scales"]), rotations=n(base["rotations"]), colors_precomp=n(out1["base"]["colors"]).astype(np.float32),
                                 bg=np.zeros(3, np.float32))
        frag_px = orc.raster_audit(ref)["fragile"]
        env_np = {k: v.detach().cpu() for k, v in env.items()}
        ra = otr.trace_audit(n(out1["ref_o"]).reshape(-1, 3), n(out1["ref_d"]).reshape(-1, 3), n(env["means3D"]), n(env["scales"]), n(env["rotations"]),
                             n(env["opacities"]), start_from_first=False, shs=n(env["shs"]), sh_degree=deg)
        keep_np = ~(frag_px | ra["fragile"].reshape(H, W))
        keep = torch.from_numpy(keep_np)
        nex = int((~keep_np).sum())
        record_fragile(test, "excluded_pixels", ~keep_np, FRAGILE_RAYS_MAX, "(%d fragile pixels, %d fragile reflected rays)" % (int(frag_px.sum()), int(ra["fragile"].sum())))
        # 2. the step with gradients, every extension call tapped; upstream gradient only on the determined pixels
        with stagewise.RasterTap() as rtap, stagewise.TraceTap() as ttap:
            out, g_h, (base, env, cam, rays) = _run(pkg, tpkg, tpkg.SurfelTracer(), dev, keep)
        torch.cuda.synchronize()
    finally:
        envgs_step.FUSED["on"] = False
    assert len(rtap.calls) == 1 and len(ttap.calls) == 1
    assert torch.equal(out["ref_o"], out1["ref_o"]) and torch.equal(out["base"]["img"], out1["base"]["img"])          # the forward is deterministic
    okp = keep_np
    # ---- link 1: the raster call, on the colours the HIP step computed and the upstream gradients it received ---------------------------
    rc = rtap.calls[0]
    check_close(test, "raster.img", n(out["base"]["img"])[:, okp], ref["out_color"][:, okp], excluded=nex)
    for ch, nm in ((0, "depth"), (1, "alpha"), (2, "normal.x"), (3, "normal.y"), (4, "normal.z")):
        check_close(test, "raster.allmap." + nm, n(out["base"]["allmap"])[ch][okp], ref["allmap"][ch][okp], excluded=nex)
    rb = orc.raster_backward(ref, n(rc["dL_dcolor"]), n(rc["dL_dallmap"]), want_cond=True)
    assert float(np.abs(n(rc["dL_dallmap"])[0:5]).max()) > 0          # the tracer's ray gradients reached the rasterizer's depth / normal maps
    for k_hip, k_ref in (("means3D", "dmeans3D"), ("scales", "dscales"), ("rotations", "drots"), ("opacities", "dopacities"), ("means2D", "dmeans2D"),
                         ("colors_precomp", "dcolors")):
        check_close(test, "raster." + k_ref, n(rc["grads"][k_hip]).reshape(rb[k_ref].shape), rb[k_ref], excluded=nex, cond=rb["cond"][k_ref], unc=rb["unc"][k_ref])
    # ---- link 2: the traced call, on the reflected rays the HIP step built and the upstream gradients it received -------------------------
    tc = ttap.calls[0]
    assert tc["sff"] is False
    _, tb = stagewise.oracle_trace_call(test, "trace", tc, env_np, np.array([0.1, 0.2, 0.3], np.float32), deg, use_sh=True, others=False, nfr=nex,
                                        keep=keep_np.reshape(-1))
    stagewise.check_summed_param_grads(test, "trace", env, [tb], nfr=nex)
    assert float(out["rgb_env"].detach().abs().mean()) > 0.05 and float(out["base"]["spec"].detach().mean()) > 0.01        # the env pass matters
    assert float(np.abs(n(tc["o_in"].grad)).max()) > 0 and float(np.abs(tb["dray_d"]).max()) > 0                              # and its ray gradients are exercised
    # ---- link 3: the glue (torch expressions or the fused HIP kernels) vs the same expressions in float64 on the recorded tensors ---------
    dd = torch.float64
    img64 = out["base"]["img"].detach().cpu().to(dd).requires_grad_(True)
    all64 = out["base"]["allmap"].detach().cpu().to(dd).requires_grad_(True)
    env64 = out["rgb_env"].detach().cpu().to(dd).requires_grad_(True)
    ref_o, ref_d, rgb = _glue_f64(cam, rays, img64, all64, env64)
    stagewise.glue_check(test, "glue.ref_o", out["ref_o"], ref_o, floor=1.0)
    stagewise.glue_check(test, "glue.ref_d", out["ref_d"], ref_d, floor=1.0)
    stagewise.glue_check(test, "glue.rgb", out["rgb"], rgb, floor=1.0)
    dcol, dall = _upstream(H, W, torch.device("cpu"), keep)
    loss = (rgb * dcol.to(dd)).sum() + (all64 * dall.to(dd)).sum() + (ref_o * tc["o_in"].grad.cpu().to(dd)).sum() + (ref_d * tc["d_in"].grad.cpu().to(dd)).sum()
    loss.backward()
    fl = lambda t: float(t.abs().mean()) + 1e-12
    stagewise.glue_check(test, "glue.d_img", rc["dL_dcolor"], img64.grad, floor=fl(img64.grad))
    stagewise.glue_check(test, "glue.d_allmap", rc["dL_dallmap"], all64.grad, floor=fl(all64.grad))
    stagewise.glue_check(test, "glue.d_rgb_env", tc["up"][0], env64.grad, floor=fl(env64.grad))
    # the SH colours handed to the rasterizer, and their gradient back into the base parameters (upstream = the raster kernels' dcolors)
    B64 = {k: v.detach().cpu().to(dd).requires_grad_(True) for k, v in base.items()}
    col64, raw, dir64 = _colors_f64(cam, B64, deg)
    near_clamp = (raw.detach().abs() < 1e-5).any(dim=1).numpy()                      # a colour channel within rounding of the clamp at 0: either branch is right
    record(test, "glue.colors.near_clamp_surfels", float(near_clamp.mean()))
    stagewise.glue_check(test, "glue.colors", out["base"]["colors"].float()[torch.from_numpy(~near_clamp).to(dev)], col64[torch.from_numpy(~near_clamp)], floor=1.0)
    (col64 * rc["grads"]["colors_precomp"].detach().cpu().to(dd)).sum().backward()
    nc = torch.from_numpy(~near_clamp)
    for k in ("shs", "specular", "roughness"):
        stagewise.glue_check(test, "glue.d_" + k, g_h["base." + k][nc.to(dev)], B64[k].grad[nc], floor=fl(B64[k].grad))
    # d colour / d position goes through the NORMALISED view direction: dL/dp = (I - d d^T) dL/dd / |p - c| subtracts the radial part of
    # dL/dd, which is most of it -- the element is judged against the magnitude of what is subtracted, as the kernels' gradients are
    vlen = (B64["means3D"].detach() - cam.camera_center.detach().cpu().double()[None]).norm(dim=1, keepdim=True)
    # (and the glue part is read off the leaf as leaf - kernel part: that difference carries the rounding of the SUM, an ulp of the larger addend)
    cond_p = (dir64.grad.abs().sum(dim=1, keepdim=True) / vlen).expand(-1, 3) + rc["grads"]["means3D"].detach().cpu().double().abs()
    stagewise.glue_check(test, "glue.d_means3D", (g_h["base.means3D"] - rc["grads"]["means3D"])[nc.to(dev)], B64["means3D"].grad[nc], cond=cond_p[nc])
    # the leaves received exactly kernel gradient (+ glue): nothing else feeds them
    for k in ("scales", "rotations", "opacities"):
        assert torch.equal(g_h["base." + k].reshape(-1), rc["grads"][k].reshape(-1).to(g_h["base." + k].dtype))
    assert {"base.means3D", "base.rotations", "base.specular", "env.shs", "env.means3D"} <= set(g_h)
    assert float(g_h["base.rotations"].abs().max()) > 0


def test_full_envgs_step_chain_vs_oracle_chain_diagnostic():
    """The whole chain over the HIP extensions against the whole chain over the oracle packages (the round-2 form of this test).  The two base
    passes differ by fp32 rounding, so the two chains trace slightly different reflected rays: images are compared on the rays that did
    not flip, gradients are RECORDED in the error table and bounded loosely -- the contract is applied link by link above."""
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    from oracle import raster as orc, trace as otr
    from tests.oracle_packages import make_raster_pkg, make_trace_pkg
    test = "envgs_step_chain_diagnostic"
    opkg, otpkg = make_raster_pkg(5), make_trace_pkg()
    out_o, _, (base, env, cam, rays) = _run(opkg, otpkg, otpkg.SurfelTracer(), torch.device("cpu"), backward=False)
    H, W = cam.image_height, cam.image_width
    ref = orc.raster_forward(n(base["means3D"]), n(base["opacities"]), n(cam.world_view_transform), n(cam.full_proj_transform), n(cam.camera_center), W, H,
                             scales=n(base["scales"]), rotations=n(base["rotations"]),
                             colors_precomp=np.zeros((base["means3D"].shape[0], 5), np.float32), bg=np.zeros(3, np.float32))
    frag_px = orc.raster_audit(ref)["fragile"]
    ra = otr.trace_audit(n(out_o["ref_o"]).reshape(-1, 3), n(out_o["ref_d"]).reshape(-1, 3), n(env["means3D"]), n(env["scales"]), n(env["rotations"]),
                         n(env["opacities"]), start_from_first=False)
    keep = torch.from_numpy(~(frag_px | ra["fragile"].reshape(H, W)))
    out_h, g_h, _ = _run(pkg, tpkg, tpkg.SurfelTracer(), torch.device("cuda:0"), keep)
    out_o, g_o, _ = _run(opkg, otpkg, otpkg.SurfelTracer(), torch.device("cpu"), keep)
    k = keep.numpy()
    for nm in ("rgb", "rgb_env"):
        a, r = n(out_h[nm])[k], n(out_o[nm])[k]
        bad = (floor_rel_err(a, r)[0] > 1e-4).any(axis=-1)
        record(test, nm + ".flipped_rays", float(bad.mean()))
        assert bad.mean() <= 5e-3
        check_close(test, nm, a[~bad], r[~bad])
    assert set(g_h) == set(g_o)
    for kk in sorted(g_h):
        check_close(test, kk, n(g_h[kk]), n(g_o[kk]), tol=1e-2)


def test_step_with_deferred_env_surfel_gradients_and_activated_parameters():
    """The same with RAW env parameters behind activations (sigmoid / exp / normalize, as EasyVolcap's GaussianModel holds them): envgs_forward puts the
    activated tensors through tracing.defer_barrier before the base pass; the tracer defers, the barrier joins inside backward() -- nothing is pending
    afterwards -- and the raw parameters' gradients are those of the stream-ordered step."""
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    from envgs_amd import tracing
    dev = torch.device("cuda:0")
    was = (envgs_step.FUSED["on"], envgs_step.DEFER["on"])
    res, deferred_calls = {}, {}
    try:
        envgs_step.FUSED["on"] = True
        for defer in (False, True):
            envgs_step.DEFER["on"] = defer
            base, env, cam = _scene(dev)
            raw = dict(means3D=env["means3D"].detach().clone().requires_grad_(True), shs=env["shs"].detach().clone().requires_grad_(True),
                       opacities=torch.logit(env["opacities"].detach().clamp(1e-4, 1 - 1e-4)).requires_grad_(True),
                       scales=torch.log(env["scales"].detach()).requires_grad_(True), rotations=(env["rotations"].detach() * 1.3).requires_grad_(True))
            act = dict(means3D=raw["means3D"] * 1.0, shs=raw["shs"] * 1.0, opacities=torch.sigmoid(raw["opacities"]), scales=torch.exp(raw["scales"]),
                       rotations=torch.nn.functional.normalize(raw["rotations"], dim=-1))
            n = [0]
            orig = tracing.trace_backward
            def counting(saved, *a, **kw):
                r = orig(saved, *a, **kw)
                n[0] += int(saved["lists"].defer_reduce & 1)
                return r
            tracing.trace_backward = counting
            try:
                out = envgs_step.envgs_forward(pkg, tpkg, tpkg.SurfelTracer(), cam, synth.get_rays(cam), base, act, torch.zeros(3, device=dev),
                                               torch.tensor([0.1, 0.2, 0.3], device=dev), torch.tensor([2], device=dev))
                dcol, dall = _upstream(cam.image_height, cam.image_width, dev)
                ((out["rgb"] * dcol).sum() + (out["base"]["allmap"] * dall).sum()).backward()
            finally:
                tracing.trace_backward = orig
            assert not tracing._DEFERRED["pending"]
            deferred_calls[defer] = n[0]
            torch.cuda.synchronize()
            res[defer] = {("raw." + k): v.grad.clone() for k, v in raw.items()} | {("base." + k): v.grad.clone() for k, v in base.items() if v.grad is not None}
    finally:
        envgs_step.FUSED["on"], envgs_step.DEFER["on"] = was
    assert deferred_calls == {False: 0, True: 1}
    for k in res[False]:
        a, b = res[False][k], res[True][k]
        assert (float(a.abs().max()) > 0 or not k.startswith("raw.")) and float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12, k


def test_step_with_deferred_env_surfel_gradients():
    """envgs_step.DEFER (SurfelTracer.set_deferred_surfel_gradients; include/envgs_trace.h: defer_reduce): the fused step whose env-surfel gradients
    finish on the library's stream beside the base pass's backward, joined by FusedAdam.step -- the parameters after one optimizer step are those
    of the stream-ordered step, for the base set (which only needs the ray gradients) and the env set alike."""
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    from envgs_amd import tracing
    from envgs_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    was = (envgs_step.FUSED["on"], envgs_step.DEFER["on"])
    res = {}
    try:
        envgs_step.FUSED["on"] = True
        for defer in (False, True):
            envgs_step.DEFER["on"] = defer
            base, env, cam = _scene(dev)
            opt = FusedAdam([{"params": list(base.values()) + list(env.values()), "lr": 1e-3}], eps=1e-15)
            rays = synth.get_rays(cam)
            out = envgs_step.envgs_forward(pkg, tpkg, tpkg.SurfelTracer(), cam, rays, base, env, torch.zeros(3, device=dev),
                                           torch.tensor([0.1, 0.2, 0.3], device=dev), torch.tensor([2], device=dev))
            dcol, dall = _upstream(cam.image_height, cam.image_width, dev)
            ((out["rgb"] * dcol).sum() + (out["base"]["allmap"] * dall).sum()).backward()
            assert tracing._DEFERRED["pending"] == defer
            if defer:
                # autograd MOVED the tail's output buffers into .grad (it would copy a tensor somebody else still holds -- on the current stream,
                # at once, i.e. before the tail has written it): the hazard the promise of set_deferred_surfel_gradients is about, checked here
                owned = {st.data_ptr() for st in tracing._DEFERRED["keep"] if isinstance(st, torch.UntypedStorage)}
                for k in ("means3D", "scales", "rotations", "opacities", "shs"):
                    assert env[k].grad.untyped_storage().data_ptr() in owned, k
            opt.step()                                                   # joins
            assert not tracing._DEFERRED["pending"]
            torch.cuda.synchronize()
            res[defer] = ({("base." + k): v.grad.clone() for k, v in base.items() if v.grad is not None} |
                          {("env." + k): v.grad.clone() for k, v in env.items() if v.grad is not None},
                          {("base." + k): v.detach().clone() for k, v in base.items()} | {("env." + k): v.detach().clone() for k, v in env.items()})
    finally:
        envgs_step.FUSED["on"], envgs_step.DEFER["on"] = was
    assert set(res[False][0]) == set(res[True][0]) and any(k.startswith("env.") for k in res[True][0])
    for k in res[False][0]:
        a, b = res[False][0][k], res[True][0][k]
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12, k
    for k in res[False][1]:
        # (Adam's first step moves every touched element by ~lr whatever the gradient's size: compare where the gradient is not rounding noise)
        a, b = res[False][1][k], res[True][1][k]
        assert float((a - b).abs().max()) <= 2.1e-3, k
        gk = res[False][0].get(k)
        if gk is not None:
            big = gk.abs() > 1e-3 * gk.abs().max()
            assert float((a - b)[big].abs().max()) <= 1e-5 if big.any() else True, k
