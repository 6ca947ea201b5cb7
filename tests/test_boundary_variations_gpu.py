"""GPU: the call-to-call variability SURVEY.md section 8(b) lists for the two plugin surfaces, each checked against the plain call --
a non-default current stream (the runner copies on a second stream and syncs with wait_stream: volumetric_video_runner.py:382-394; kernels must
launch on torch.cuda.current_stream()), non-contiguous argument views (the raster caller does not call .contiguous(): `viewmatrix` is a transposed
view, gaussian2d_utils.py:86; `get_features` is a cat), float64 parameters, `debug=True`, `scale_modifier != 1`, and random non-PSD
`cov3D_precomp` (the reference's crash-repro input, tests/cuda_illegal_memory_access_tests.py:8-33)."""
import pytest
import torch

from envgs_amd import synth

pytestmark = pytest.mark.gpu

H, W = 80, 112


def _settings(mod, cam, dev, deg=3, debug=False, scale_modifier=1.0, viewmatrix=None):
    return mod.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.tensor([0.3, 0.1, 0.2], device=dev),
                                             scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform if viewmatrix is None else viewmatrix,
                                             projmatrix=cam.full_proj_transform, sh_degree=torch.tensor([deg], device=dev), campos=cam.camera_center, prefiltered=False, debug=debug)


def _raster(mod, st, leaves, cov=None):
    m2 = torch.zeros(leaves["means3D"].shape[0], 3, device=leaves["means3D"].device, dtype=leaves["means3D"].dtype, requires_grad=True)
    color, radii, allmap, weight = mod.GaussianRasterizer(raster_settings=st)(
        means3D=leaves["means3D"], means2D=m2, shs=leaves["shs"], colors_precomp=None, opacities=leaves["opacities"],
        scales=None if cov is not None else leaves["scales"], rotations=None if cov is not None else leaves["rotations"], cov3D_precomp=cov)
    return color, radii, allmap, weight


def _scene(dev, P=2500):
    g = synth.base_gaussians(P, seed=5); g["scales"] = g["scales"] * 4
    cam = synth.orbit_camera(2, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
    return {k: g[k].to(dev) for k in ("means3D", "shs", "opacities", "scales", "rotations")}, cam


def _fwd_bwd(mod, st, g):
    L = {k: v.clone().requires_grad_(True) for k, v in g.items()}
    color, radii, allmap, weight = _raster(mod, st, L)
    (color.sum() * 0.5 + allmap[:5].sum()).backward()
    return color.detach(), radii, allmap.detach(), {k: v.grad for k, v in L.items()}


def _close(a, b, tol=1e-5):
    return float((a.float() - b.float()).abs().max()) <= tol * (float(b.float().abs().max()) + 1e-30)


def test_rasterizer_on_a_side_stream_with_strided_views_double_parameters_and_debug():
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    g, cam = _scene(dev)
    ref_c, ref_r, ref_a, ref_g = _fwd_bwd(mod, _settings(mod, cam, dev), g)
    torch.cuda.synchronize()
    # 1. a non-default current stream, synchronised ONLY through that stream: if a kernel ran on the default stream the results would race
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        c, r, a, gr = _fwd_bwd(mod, _settings(mod, cam, dev), g)
    side.synchronize()
    assert torch.equal(c, ref_c) and torch.equal(r, ref_r) and torch.equal(a, ref_a)
    for k in ref_g:
        assert _close(gr[k], ref_g[k]), k
    # 2. strided / transposed views (no .contiguous() at the raster call site)
    P = g["means3D"].shape[0]
    big = torch.zeros(P, 7, device=dev); big[:, 2:5] = g["means3D"]
    shs_t = g["shs"].transpose(1, 2).contiguous().transpose(1, 2)                        # (P,16,3) view of a (P,3,16) buffer: eval_sh's layout
    vm_t = cam.world_view_transform.t().contiguous().t()                                 # the transposed view the caller builds (gaussian2d_utils.py:86)
    gv = dict(g, means3D=big[:, 2:5], shs=shs_t, scales=torch.cat([g["scales"], g["scales"]], 1)[:, 2:4], rotations=g["rotations"].flip(1).flip(1))
    assert not gv["means3D"].is_contiguous() and not gv["shs"].is_contiguous() and not vm_t.is_contiguous()
    c, r, a, gr = _fwd_bwd(mod, _settings(mod, cam, dev, viewmatrix=vm_t), gv)
    assert torch.equal(c, ref_c) and torch.equal(r, ref_r)
    for k in ref_g:
        assert gr[k].shape == ref_g[k].shape and _close(gr[k], ref_g[k]), k
    # 3. float64 parameters: computed in fp32, gradients come back in the parameters' dtype (autograd would reject anything else)
    c, r, a, gr = _fwd_bwd(mod, _settings(mod, cam, dev), {k: v.double() for k, v in g.items()})
    assert _close(c, ref_c, 1e-6) and torch.equal(r, ref_r)
    for k in ref_g:
        assert gr[k].dtype == torch.float64 and _close(gr[k], ref_g[k]), k
    # 4. debug=True (upstream: sync + check after every kernel): same results
    c, r, a, gr = _fwd_bwd(mod, _settings(mod, cam, dev, debug=True), g)
    assert torch.equal(c, ref_c) and torch.equal(r, ref_r)
    # 5. scale_modifier: s * scales with modifier 1 == scales with modifier s
    s = 0.7
    c1, r1, a1, _ = _fwd_bwd(mod, _settings(mod, cam, dev, scale_modifier=s), g)
    c2, r2, a2, _ = _fwd_bwd(mod, _settings(mod, cam, dev), dict(g, scales=g["scales"] * s))
    assert _close(c1, c2, 2e-5) and int((r1 != r2).sum()) <= 2


def test_rasterizer_random_covariances_do_not_crash():
    """The reference's own robustness repro: random (non-PSD, unbounded) `cov3D_precomp` rows -- here the (P,9) transMat the 2DGS caller passes
    (gaussian2d_utils.py:1050-1061).  Nothing to compare with; it must return, stay in bounds (compute-sanitizer is not available: the canary is a
    second, sane render right after on the same buffers' neighbours) and produce finite gradients for finite upstream."""
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    g, cam = _scene(dev)
    st = _settings(mod, cam, dev)
    gen = torch.Generator().manual_seed(3)
    for scale in (1.0, 1e3, 1e-6, 1e12):
        cov = (torch.randn(g["means3D"].shape[0], 9, generator=gen) * scale).to(dev).requires_grad_(True)
        L = {k: v.clone().requires_grad_(True) for k, v in g.items()}
        color, radii, allmap, weight = _raster(mod, st, L, cov=cov)
        color.sum().backward()
        torch.cuda.synchronize()
        assert color.shape == (3, H, W) and bool(torch.isfinite(color).all()), scale
        assert bool(torch.isfinite(cov.grad).all()) and bool(torch.isfinite(L["means3D"].grad).all()), scale
    ref_c, *_ = _fwd_bwd(mod, st, g)
    again, *_ = _fwd_bwd(mod, st, g)
    assert torch.equal(ref_c, again)


def test_tracer_on_a_side_stream_with_views_and_double_rays():
    import diff_surfel_tracing as tpkg
    dev = torch.device("cuda:0")
    e = {k: v.to(dev) for k, v in synth.env_gaussians(3000, seed=6, bound=12.0).items()}
    gen = torch.Generator().manual_seed(2)
    ro = ((torch.rand(48, 64, 3, generator=gen) * 2 - 1)).to(dev); rd = torch.randn(48, 64, 3, generator=gen).to(dev)
    ts = tpkg.SurfelTracingSettings(image_height=48, image_width=64, tanfovx=1.0, tanfovy=1.0, bg=torch.tensor([0.1, 0.2, 0.3], device=dev), scale_modifier=1.0,
                                    viewmatrix=torch.eye(4, device=dev), projmatrix=torch.eye(4, device=dev), sh_degree=torch.tensor([3], device=dev),
                                    campos=torch.zeros(3, device=dev), prefiltered=False, debug=False, max_trace_depth=0, specular_threshold=0.0)

    def run(e_, o_, d_, st_=ts):
        L = {k: e_[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        o = o_.clone().requires_grad_(True); d = d_.clone().requires_grad_(True)
        v, f = synth.get_disks(L["means3D"].detach().float(), L["scales"].detach().float(), L["rotations"].detach().float())
        t = tpkg.SurfelTracer(); t.build_acceleration_structure(v, f, rebuild=True)
        outs = t(o, d, v, means3D=L["means3D"], grads3D=None, shs=L["shs"], colors_precomp=None, others_precomp=None, opacities=L["opacities"], scales=L["scales"],
                 rotations=L["rotations"], cov3D_precomp=None, tracer_settings=st_, start_from_first=False)
        (outs[0].sum() + 0.1 * outs[1].sum()).backward()
        return [x.detach() for x in outs], {k: x.grad for k, x in L.items()}, o.grad, d.grad
    ref, gref, go, gd = run(e, ro, rd)
    torch.cuda.synchronize()
    assert ref[0].shape == (48, 64, 3) and ref[7].shape == (3000, 1) and float(ref[2].mean()) > 0.3
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        outs, grads, go2, gd2 = run(e, ro, rd)
    side.synchronize()
    for i in (0, 1, 2, 3, 5, 7):
        assert torch.equal(outs[i], ref[i]), i
    for k in gref:
        assert _close(grads[k], gref[k]), k
    assert _close(go2, go) and _close(gd2, gd)
    # float64 rays and parameters (ray_utils produces float32, but a caller is free not to): same result, gradients in the inputs' dtype
    outs, grads, go3, gd3 = run({k: v.double() for k, v in e.items()}, ro.double(), rd.double())
    assert _close(outs[0], ref[0], 1e-6) and go3.dtype == torch.float64 and _close(go3, go) and _close(grads["shs"], gref["shs"])
    # debug=True
    outs, _, _, _ = run(e, ro, rd, ts._replace(debug=True))
    assert torch.equal(outs[0], ref[0])


def test_two_outstanding_forwards_and_a_repeated_backward():
    """Multi-view accumulation (north_star's 8-view batch on fewer GPUs): several forwards of BOTH extensions are outstanding when backward() runs --
    each call's saved state must be its own (no scratch shared between calls that a later forward overwrites) -- and backward(retain_graph=True)
    may run twice over the same saved state."""
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    from envgs_amd import envgs_step
    dev = torch.device("cuda:0")
    Hh, Ww = 64, 80
    base = {k: v.to(dev) for k, v in synth.base_gaussians(1500, seed=3).items()}
    base["scales"] = base["scales"] * 5.0
    env = {k: v.to(dev) for k, v in synth.env_gaussians(800, seed=4, bound=12.0).items()}
    cams = [synth.orbit_camera(v, n_views=4, H=Hh, W=Ww, fx=1111.1 * Ww / 800.0, device=dev) for v in range(3)]
    rays = [synth.get_rays(c) for c in cams]
    bg = torch.zeros(3, device=dev); env_bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    deg = torch.tensor([2], device=dev)
    gen = torch.Generator().manual_seed(5)
    ups = [torch.randn(Hh, Ww, 3, generator=gen).to(dev) / (Hh * Ww) for _ in cams]
    for fused in (False, True):
        envgs_step.FUSED["on"] = fused
        try:
            def leaves():
                return ({k: v.clone().requires_grad_(True) for k, v in base.items()}, {k: v.clone().requires_grad_(True) for k, v in env.items()})

            def grads_of(b, e):
                return {**{"base." + k: v.grad.clone() for k, v in b.items() if v.grad is not None}, **{"env." + k: v.grad.clone() for k, v in e.items() if v.grad is not None}}
            # reference: one view at a time, gradients accumulated by autograd
            b, e = leaves()
            tracer = tpkg.SurfelTracer()
            for v in range(3):
                out = envgs_step.envgs_forward(pkg, tpkg, tracer, cams[v], rays[v], b, e, bg, env_bg, deg)
                (out["rgb"] * ups[v]).sum().backward()
            want = grads_of(b, e)
            # three forwards outstanding, ONE backward
            b, e = leaves()
            tracer = tpkg.SurfelTracer()
            loss = 0.0
            for v in range(3):
                out = envgs_step.envgs_forward(pkg, tpkg, tracer, cams[v], rays[v], b, e, bg, env_bg, deg)
                loss = loss + (out["rgb"] * ups[v]).sum()
            loss.backward(retain_graph=True)
            torch.cuda.synchronize()
            got = grads_of(b, e)
            assert set(got) == set(want)
            for k in want:
                assert _close(got[k], want[k], 2e-5), (fused, k, float((got[k] - want[k]).abs().max()), float(want[k].abs().max()))
            # ... and once more over the same saved state: the accumulated gradients double
            loss.backward()
            torch.cuda.synchronize()
            twice = grads_of(b, e)
            for k in want:
                assert _close(twice[k], 2.0 * want[k], 2e-5), (fused, "second backward", k)
        finally:
            envgs_step.FUSED["on"] = False
