"""GPU parity: the HIP rasterizer (through the C-ABI) against the CPU oracle, stage by stage.
Bar (BASELINE.json north_star): bit-exact tile / sort indices; <=1e-4 rel on pixels and gradients."""
import numpy as np
import pytest
import torch

from tests.util import small_scene, cam_args, rel_err, assert_close_frac

pytestmark = pytest.mark.gpu

PIX_TOL = 1e-4
GRAD_TOL = 1e-4


def _settings(mod, cam, bg, deg, dev, scale_modifier=1.0):
    return mod.GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=bg.to(dev), scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(dev),
        projmatrix=cam.full_proj_transform.to(dev), sh_degree=torch.tensor([deg], device=dev), campos=cam.camera_center.to(dev),
        prefiltered=False, debug=False)


def _oracle(g, cam, bg, deg, C, sh, precomp_T=None, scale_modifier=1.0):
    from oracle import raster as orc
    ca = cam_args(cam)
    geo = dict(transmat_precomp=precomp_T.numpy()) if precomp_T is not None else dict(scales=g["scales"].numpy(), rotations=g["rotations"].numpy())
    col = dict(shs=g["shs"].numpy(), sh_degree=deg) if sh else dict(colors_precomp=g["colors_precomp"].numpy())
    return orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), ca["viewmatrix"].numpy(), ca["projmatrix"].numpy(),
                              ca["campos"].numpy(), ca["W"], ca["H"], bg=bg.numpy(), scale_modifier=scale_modifier, **geo, **col)


CASES = [
    dict(P=600, H=64, W=80, C=3, sh=True, deg=3, seed=0),
    dict(P=600, H=70, W=90, C=3, sh=True, deg=1, seed=1),          # ragged image (not a multiple of 16)
    dict(P=500, H=64, W=64, C=5, sh=False, deg=0, seed=2),         # ch05 with a 3-entry bg
    dict(P=500, H=48, W=100, C=7, sh=False, deg=0, seed=3),
    dict(P=3000, H=128, W=128, C=3, sh=True, deg=3, seed=4, scale_mul=6.0),   # long per-tile lists (> 256 per batch)
]


@pytest.mark.parametrize("case", CASES)
def test_forward_stages_vs_oracle(case):
    from envgs_amd import raster
    dev = torch.device("cuda:0")
    g, cam = small_scene(P=case["P"], H=case["H"], W=case["W"], seed=case["seed"], C=case["C"], sh=case["sh"],
                         scale_mul=case.get("scale_mul", 4.0))
    bg = torch.tensor([0.2, 0.5, 0.9])
    C = case["C"]
    import importlib
    mod = importlib.import_module({3: "diff_surfel_rasterization_wet", 5: "diff_surfel_rasterization_wet_ch05", 7: "diff_surfel_rasterization_wet_ch07"}[C])
    st = _settings(mod, cam, bg, case["deg"], dev)
    gd = {k: v.to(dev) for k, v in g.items()}
    outs, saved = raster.rasterize_forward(C, gd["means3D"], gd["shs"] if case["sh"] else None,
                                           None if case["sh"] else gd["colors_precomp"], gd["opacities"], gd["scales"],
                                           gd["rotations"], None, st, keep_binning=True)
    torch.cuda.synchronize()
    ref = _oracle(g, cam, bg, case["deg"], C, case["sh"])
    N = ref["N"]
    assert saved["N"] == N and N > 0

    # R1: integer outputs bit-exact, floats to 1e-5
    np.testing.assert_array_equal(saved["radii"].cpu().numpy(), ref["radii"])
    np.testing.assert_array_equal(saved["tiles_touched"].cpu().numpy().view(np.uint32), ref["tiles_touched"])
    np.testing.assert_array_equal(saved["offsets"].cpu().numpy().view(np.uint32), ref["offsets"])
    vis = ref["radii"] > 0
    geom = saved["geom"].cpu().numpy()
    np.testing.assert_array_equal(geom[vis, :9], ref["transmat"][vis])            # same op order, no FMA: exact
    np.testing.assert_array_equal(geom[vis, 9:11], ref["xy"][vis])
    np.testing.assert_array_equal(geom[vis, 11:15], ref["normal_opacity"][vis])
    np.testing.assert_array_equal(geom[vis, 15].view(np.uint32), ref["depth"][vis].view(np.uint32))
    if case["sh"]:
        np.testing.assert_allclose(saved["colors"].cpu().numpy()[vis], ref["rgb"][vis], rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(saved["clamped"].cpu().numpy()[vis], ref["clamped"][vis])

    # R3-R5: keys, sorted list, ranges bit-exact
    np.testing.assert_array_equal(saved["keys_unsorted"].cpu().numpy().view(np.uint64)[:N], ref["keys_unsorted"])
    np.testing.assert_array_equal(saved["vals_unsorted"].cpu().numpy().view(np.uint32)[:N], ref["vals_unsorted"])
    np.testing.assert_array_equal(saved["keys_sorted"].cpu().numpy().view(np.uint64)[:N], ref["keys_sorted"])
    np.testing.assert_array_equal(saved["point_list"].cpu().numpy().view(np.uint32)[:N], ref["point_list"])
    np.testing.assert_array_equal(saved["ranges"].cpu().numpy().view(np.uint32), ref["ranges"])

    # R6: pixels within 1e-4 relative; contributor counts equal except where a threshold sits inside fp noise
    color, radii, allmap, weight = [o.cpu().numpy() for o in outs]
    assert rel_err(color, ref["out_color"]) < PIX_TOL
    for ch in (0, 1, 2, 3, 4):
        assert rel_err(allmap[ch], ref["allmap"][ch]) < PIX_TOL, ch
    assert rel_err(allmap[6], ref["allmap"][6]) < 5e-3            # fp32 cancellation in both (see test_oracle_grad)
    nc = saved["n_contrib"].cpu().numpy()
    assert (nc[0] != ref["n_contrib"][0]).mean() < 2e-3
    assert (nc[1] != ref["n_contrib"][1]).mean() < 2e-3
    same = (nc[1] == ref["n_contrib"][1])
    assert rel_err(allmap[5][same], ref["allmap"][5][same]) < PIX_TOL
    assert rel_err(saved["final_T"].cpu().numpy(), ref["final_T"]) < PIX_TOL
    assert rel_err(weight[:, 0], ref["weight"]) < PIX_TOL


@pytest.mark.parametrize("case", CASES)
def test_backward_vs_oracle(case):
    from oracle import raster as orc
    import importlib
    dev = torch.device("cuda:0")
    C = case["C"]
    g, cam = small_scene(P=case["P"], H=case["H"], W=case["W"], seed=case["seed"], C=C, sh=case["sh"],
                         scale_mul=case.get("scale_mul", 4.0))
    bg = torch.tensor([0.2, 0.5, 0.9])
    mod = importlib.import_module({3: "diff_surfel_rasterization_wet", 5: "diff_surfel_rasterization_wet_ch05", 7: "diff_surfel_rasterization_wet_ch07"}[C])
    st = _settings(mod, cam, bg, case["deg"], dev)
    H, W = case["H"], case["W"]
    gen = torch.Generator().manual_seed(case["seed"] + 100)
    dcol = torch.randn(C, H, W, generator=gen) / (H * W)
    dall = torch.randn(7, H, W, generator=gen) / (H * W)

    leaves = {k: g[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations")}
    if case["sh"]: leaves["shs"] = g["shs"].to(dev).requires_grad_(True)
    else: leaves["colors_precomp"] = g["colors_precomp"].to(dev).requires_grad_(True)
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True) + 0
    means2D.retain_grad()
    color, radii, allmap, weight = mod.GaussianRasterizer(raster_settings=st)(
        means3D=leaves["means3D"], means2D=means2D, shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"),
        opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    assert not radii.requires_grad and not weight.requires_grad
    loss = (color * dcol.to(dev)).sum() + (allmap * dall.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()

    ref = _oracle(g, cam, bg, case["deg"], C, case["sh"])
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy())
    tol = GRAD_TOL * 5      # atomics: summation order differs; a handful of threshold flips in fp noise
    assert rel_err(leaves["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"]) < tol
    assert rel_err(leaves["means3D"].grad.cpu().numpy(), rb["dmeans3D"]) < tol
    assert rel_err(leaves["scales"].grad.cpu().numpy(), rb["dscales"]) < tol
    assert rel_err(leaves["rotations"].grad.cpu().numpy(), rb["drots"]) < tol
    assert rel_err(means2D.grad.cpu().numpy(), rb["dmeans2D"]) < tol
    if case["sh"]: assert rel_err(leaves["shs"].grad.cpu().numpy(), rb["dshs"]) < tol
    else: assert rel_err(leaves["colors_precomp"].grad.cpu().numpy(), rb["dcolors"]) < tol


def test_precomputed_transmat_path():
    """cov3D_precomp (the python transMat of gaussian2d_utils.py:1050-1061) instead of scales/rotations."""
    from envgs_amd import synth
    from oracle import raster as orc
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    g, cam = small_scene(P=500, H=64, W=64, seed=9, C=3, sh=True)
    bg = torch.zeros(3)
    tm = synth.transmat_python(cam, g["means3D"], g["scales"], g["rotations"])
    st = _settings(mod, cam, bg, 2, dev)
    tmd = tm.to(dev).requires_grad_(True)
    m3 = g["means3D"].to(dev).requires_grad_(True)
    shs = g["shs"].to(dev).requires_grad_(True)
    op = g["opacities"].to(dev).requires_grad_(True)
    means2D = torch.zeros_like(m3, requires_grad=True) + 0
    color, radii, allmap, weight = mod.GaussianRasterizer(raster_settings=st)(
        means3D=m3, means2D=means2D, shs=shs, colors_precomp=None, opacities=op, scales=None, rotations=None, cov3D_precomp=tmd)
    gen = torch.Generator().manual_seed(5)
    dcol = torch.randn(3, 64, 64, generator=gen) / 4096
    dall = torch.randn(7, 64, 64, generator=gen) / 4096
    ((color * dcol.to(dev)).sum() + (allmap * dall.to(dev)).sum()).backward()
    ref = _oracle(g, cam, bg, 2, 3, True, precomp_T=tm)
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy())
    assert rel_err(color.detach().cpu().numpy(), ref["out_color"]) < PIX_TOL
    assert rel_err(tmd.grad.cpu().numpy(), rb["dtransmat_precomp"]) < 5e-4
    assert rel_err(m3.grad.cpu().numpy(), rb["dmeans3D"]) < 5e-4          # SH view-direction term only
    assert rel_err(shs.grad.cpu().numpy(), rb["dshs"]) < 5e-4


def test_edge_cases():
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    g, cam = small_scene(P=64, H=32, W=48, seed=11)
    bg = torch.tensor([0.1, 0.2, 0.3])
    st = _settings(mod, cam, bg, 0, dev)
    R = mod.GaussianRasterizer(raster_settings=st)
    # nothing visible: every surfel behind the camera -> image == background, N == 0
    m3 = (g["means3D"] + torch.tensor([100.0, 100.0, 100.0])).to(dev).requires_grad_(True)
    args = dict(means2D=torch.zeros_like(m3), shs=g["shs"].to(dev), colors_precomp=None, opacities=g["opacities"].to(dev),
                scales=g["scales"].to(dev), rotations=g["rotations"].to(dev), cov3D_precomp=None)
    color, radii, allmap, weight = R(means3D=m3, **args)
    assert int((radii > 0).sum()) == 0 and float(weight.abs().max()) == 0.0
    assert torch.allclose(color, bg.to(dev)[:, None, None].expand_as(color))
    color.sum().backward()
    assert float(m3.grad.abs().max()) == 0.0
    # P == 0
    e = lambda *s: torch.zeros(*s, device=dev)
    color, radii, allmap, weight = R(means3D=e(0, 3), means2D=e(0, 3), shs=e(0, 16, 3), colors_precomp=None,
                                     opacities=e(0, 1), scales=e(0, 2), rotations=e(0, 4), cov3D_precomp=None)
    assert color.shape == (3, 32, 48) and radii.numel() == 0 and torch.allclose(color, bg.to(dev)[:, None, None].expand_as(color))
    # argument validation mirrors the reference wrapper
    with pytest.raises(Exception):
        R(means3D=m3, **{**args, "colors_precomp": torch.rand(64, 3, device=dev)})
    with pytest.raises(Exception):
        R(means3D=m3, **{**args, "scales": None})
    # inference_mode (reference test loop runs under it: volumetric_video_runner.py:540)
    with torch.inference_mode():
        color2, *_ = R(means3D=g["means3D"].to(dev), **args)
    assert torch.isfinite(color2).all()


def test_full_size_baseline_config_vs_oracle():
    """BASELINE configs[1] at full size: 300 k surfels, 800x800, SH degree 3 -- forward indices bit-exact,
    pixels and gradients within tolerance, plus size-independent properties (sortedness, range partition)."""
    from envgs_amd import raster, synth
    from oracle import raster as orc
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    P, H, W = 300000, 800, 800
    g = synth.base_gaussians(P, seed=0)
    cam = synth.orbit_camera(3, H=H, W=W)
    bg = torch.ones(3)
    st = _settings(mod, cam, bg, 3, dev)
    gd = {k: v.to(dev) for k, v in g.items()}
    outs, saved = raster.rasterize_forward(3, gd["means3D"], gd["shs"], None, gd["opacities"], gd["scales"], gd["rotations"],
                                           None, st, keep_binning=True)
    gen = torch.Generator().manual_seed(1)
    dcol = torch.randn(3, H, W, generator=gen) / (H * W)
    dall = torch.randn(7, H, W, generator=gen) / (H * W)
    grads = raster.rasterize_backward(saved, dcol.to(dev), dall.to(dev))
    torch.cuda.synchronize()
    N = saved["N"]
    # properties that need no oracle: keys sorted, ranges partition [0,N), every list entry is a visible surfel
    ks = saved["keys_sorted"].cpu().numpy().view(np.uint64)[:N]
    assert np.all(ks[1:] >= ks[:-1])
    r = saved["ranges"].cpu().numpy().view(np.uint32).astype(np.int64)
    assert int((r[:, 1] - r[:, 0]).sum()) == N
    pl = saved["point_list"].cpu().numpy().view(np.uint32)[:N]
    rad = saved["radii"].cpu().numpy()
    assert np.all(rad[pl] > 0)
    assert int(saved["tiles_touched"].cpu().numpy().view(np.uint32).astype(np.int64).sum()) == N

    ca = cam_args(cam)
    ref = orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), ca["viewmatrix"].numpy(), ca["projmatrix"].numpy(),
                             ca["campos"].numpy(), W, H, scales=g["scales"].numpy(), rotations=g["rotations"].numpy(),
                             shs=g["shs"].numpy(), sh_degree=3, bg=bg.numpy())
    assert ref["N"] == N
    np.testing.assert_array_equal(rad, ref["radii"])
    np.testing.assert_array_equal(pl, ref["point_list"])
    np.testing.assert_array_equal(r.astype(np.uint32), ref["ranges"])
    color, _, allmap, weight = [o.cpu().numpy() for o in outs]
    assert_close_frac(color, ref["out_color"], PIX_TOL, flip_bound=0.02, what="color")
    for ch in (0, 1, 2, 3, 4):
        assert_close_frac(allmap[ch], ref["allmap"][ch], PIX_TOL, flip_bound=0.02, what="allmap%d" % ch)
    assert (saved["n_contrib"].cpu().numpy()[0] != ref["n_contrib"][0]).mean() < 1e-4
    assert_close_frac(weight[:, 0], ref["weight"], PIX_TOL, flip_bound=0.02, what="weight")
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy())
    for k_hip, k_ref in (("means3D", "dmeans3D"), ("scales", "dscales"), ("rotations", "drots"), ("opacities", "dopacities"),
                         ("shs", "dshs"), ("means2D", "dmeans2D")):
        a = grads[k_hip].cpu().numpy().reshape(rb[k_ref].shape)
        assert_close_frac(a, rb[k_ref], 2e-4, max_bad_frac=1e-4, flip_bound=0.05, what=k_hip)
