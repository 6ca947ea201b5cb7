"""GPU: BASELINE config-5-like shapes run through every entry point without error (1600x1200, -ch07 raster, 2-bounce trace over the
700 k env cap) -- sanity of shapes / finiteness / memory sizing, not parity (parity is covered at sizes the oracle can follow)."""
import pytest
import torch

from envgs_amd import synth

pytestmark = pytest.mark.gpu


def test_config5_shapes_run():
    import diff_surfel_rasterization_wet_ch07 as pkg
    import diff_surfel_tracing as tpkg
    dev = torch.device("cuda:0")
    H, W, P = 1200, 1600, 300000
    g = synth.base_gaussians(P, seed=0, device=dev)
    cam = synth.orbit_camera(2, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
    colors = torch.cat([torch.rand(P, 3, device=dev), g["specular"].expand(-1, 3), g["roughness"]], dim=-1).requires_grad_(True)
    st = pkg.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev),
                                           scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                           sh_degree=torch.tensor([3], device=dev), campos=cam.camera_center, prefiltered=False, debug=False)
    m3 = g["means3D"].clone().requires_grad_(True)
    img, radii, allmap, weight = pkg.GaussianRasterizer(raster_settings=st)(
        means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), shs=None, colors_precomp=colors, opacities=g["opacities"],
        scales=g["scales"], rotations=g["rotations"], cov3D_precomp=None)
    assert img.shape == (7, H, W) and allmap.shape == (7, H, W) and torch.isfinite(img).all() and torch.isfinite(allmap).all()
    (img.mean() + allmap[:5].mean()).backward()
    assert torch.isfinite(m3.grad).all() and torch.isfinite(colors.grad).all() and float(colors.grad.abs().max()) > 0

    # 2-bounce trace of camera rays over the env cap (700 k surfels), a 300x400 crop of the rays
    e = synth.env_gaussians(700000, seed=2, device=dev)
    ro, rd = synth.get_rays(cam)
    ro, rd = ro[450:750, 600:1000].contiguous(), rd[450:750, 600:1000].contiguous()
    v, f = synth.get_disks(e["means3D"], e["scales"], e["rotations"])
    tr = tpkg.SurfelTracer()
    tr.build_acceleration_structure(v, f, rebuild=True)
    ts = tpkg.SurfelTracingSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev),
                                    scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                    sh_degree=torch.tensor([2], device=dev), campos=cam.camera_center, prefiltered=False, debug=False,
                                    max_trace_depth=2, specular_threshold=0.05)
    others = torch.cat([torch.rand(700000, 1, device=dev), torch.full((700000, 1), 0.5, device=dev)], dim=-1)
    with torch.no_grad():
        rgb, dpt, acc, norm, dist, aux, mid, wet = tr(ro, rd, v, means3D=e["means3D"], grads3D=None, shs=e["shs"], colors_precomp=None,
                                                       others_precomp=others, opacities=e["opacities"], scales=e["scales"],
                                                       rotations=e["rotations"], cov3D_precomp=None, tracer_settings=ts, start_from_first=True)
    assert rgb.shape == (300, 400, 3) and mid.shape == (300, 400, 48) and wet.shape == (700000, 1)
    for t in (rgb, dpt, acc, norm, aux, mid, wet):
        assert torch.isfinite(t).all()
    assert float(acc.max()) > 0.5 and float((mid[..., 16 + 3:16 + 6].abs().sum(-1) > 0).float().mean()) > 0.05     # some rays bounced

    # depth-0 list path with gradients on the same env set
    rays_o = ro.reshape(-1, 3).clone().requires_grad_(True)
    ts0 = ts._replace(max_trace_depth=0)
    sh = e["shs"].clone().requires_grad_(True)
    out = tr(rays_o, rd.reshape(-1, 3), None, means3D=e["means3D"], grads3D=None, shs=sh, colors_precomp=None, others_precomp=None,
             opacities=e["opacities"], scales=e["scales"], rotations=e["rotations"], cov3D_precomp=None, tracer_settings=ts0,
             start_from_first=True)
    out[0].sum().backward()
    assert torch.isfinite(sh.grad).all() and torch.isfinite(rays_o.grad).all() and float(sh.grad.abs().max()) > 0
