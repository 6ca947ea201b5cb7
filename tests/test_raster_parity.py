"""GPU parity: the HIP rasterizer (through the C-ABI) against the CPU oracle, stage by stage.
Bar (BASELINE.json north_star): bit-exact tile / sort indices; <=1e-4 rel on pixels and gradients.

How the bar is applied (tests/util.py:check_close): EVERY element must be within 1e-4 elementwise (relative, with an absolute floor at the
tensor's mean magnitude).  Threshold flips are not absorbed by the tolerance: the oracle's audit (oracle/surfel_raster_oracle.c:
orc_render_audit) marks the pixels whose contributor set is not determined beyond rounding noise; on all OTHER pixels the per-pixel
contributor sets and n_contrib must be bit-exact (index work) and the values within 1e-4; gradients are compared with the upstream
gradient zeroed at the fragile pixels (in both implementations), so they are flip-free as well.  Fragile pixels are counted and bounded."""
import numpy as np
import pytest
import torch

from tests.util import small_scene, cam_args, rel_err, check_close, record, record_fragile, FRAGILE_PX_MAX

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



def _mod_for(C):
    import importlib
    return importlib.import_module({3: "diff_surfel_rasterization_wet", 5: "diff_surfel_rasterization_wet_ch05", 7: "diff_surfel_rasterization_wet_ch07"}[C])


def _compare_forward(test, outs, saved, ref, aud, sh, check_sets):
    """Index work bit-exact; contributor sets / n_contrib bit-exact and values within 1e-4 on every non-fragile pixel."""
    from envgs_amd import raster
    N = ref["N"]
    assert saved["N"] == N and N > 0
    frag = aud["fragile"]; ok = ~frag
    nfr = int(frag.sum())
    record_fragile(test, "fragile_px", frag, FRAGILE_PX_MAX, "(by the round-1..3 definition: %d)" % int(aud["legacy_fragile"].sum()))

    # R1: integer outputs bit-exact, geom bit-exact (same op order, no FMA)
    np.testing.assert_array_equal(saved["radii"].cpu().numpy(), ref["radii"])
    vis = ref["radii"] > 0
    geom = saved["geom"].cpu().numpy()
    np.testing.assert_array_equal(geom[vis, :9], ref["transmat"][vis])
    np.testing.assert_array_equal(geom[vis, 9:11], ref["xy"][vis])
    np.testing.assert_array_equal(geom[vis, 11:15], ref["normal_opacity"][vis])
    np.testing.assert_array_equal(geom[vis, 15].view(np.uint32), ref["depth"][vis].view(np.uint32))
    if sh:
        check_close(test, "sh_rgb", saved["colors"].cpu().numpy()[vis], ref["rgb"][vis], tol=1e-5)
        np.testing.assert_array_equal(saved["clamped"].cpu().numpy()[vis], ref["clamped"][vis])
    # R2-R5: offsets, keys, sorted list, ranges bit-exact
    if "offsets" in saved:
        np.testing.assert_array_equal(saved["tiles_touched"].cpu().numpy().view(np.uint32), ref["tiles_touched"])
        np.testing.assert_array_equal(saved["offsets"].cpu().numpy().view(np.uint32), ref["offsets"])
        # R3: the instances arrive partitioned by tile (slot order inside a tile's segment is whatever the LDS cursors handed out): as a
        # SET of (tile, depth bits, surfel id) triples they are exactly the reference's emission
        pairs = saved["tile_pairs"].cpu().numpy().view(np.uint64)[:N]
        rg = saved["ranges"].cpu().numpy().view(np.uint32).astype(np.int64)
        seg = np.cumsum(rg[:, 1] - rg[:, 0])
        assert seg[-1] == N and np.all((rg[:, 0] == seg - (rg[:, 1] - rg[:, 0])) | (rg[:, 1] == rg[:, 0]))     # segments tile the slots in tile order
        tile_of_slot = np.repeat(np.arange(rg.shape[0], dtype=np.uint64), rg[:, 1] - rg[:, 0])
        got_k, got_v = (tile_of_slot << np.uint64(32)) | (pairs >> np.uint64(32)), (pairs & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        o_got, o_ref = np.lexsort((got_v, got_k)), np.lexsort((ref["vals_unsorted"], ref["keys_unsorted"]))
        np.testing.assert_array_equal(got_k[o_got], ref["keys_unsorted"][o_ref])
        np.testing.assert_array_equal(got_v[o_got], ref["vals_unsorted"][o_ref])
        np.testing.assert_array_equal(saved["keys_sorted"].cpu().numpy().view(np.uint64)[:N], ref["keys_sorted"])
    np.testing.assert_array_equal(saved["point_list"].cpu().numpy().view(np.uint32)[:N], ref["point_list"])
    np.testing.assert_array_equal(saved["ranges"].cpu().numpy().view(np.uint32), ref["ranges"])

    # R6 index work: last / median contributor bit-exact on every non-fragile pixel; contributor SETS too (small cases)
    nc = saved["n_contrib"].cpu().numpy()
    np.testing.assert_array_equal(nc[0][ok], ref["n_contrib"][0][ok])
    np.testing.assert_array_equal(nc[1][ok], ref["n_contrib"][1][ok])
    if check_sets:
        # the audit instantiation also returns the per-surfel weight with the FRAGILE pixels left out (skip_px); the oracle leaves out the same
        # pixels (orc_render_weight), so `weight` -- which drives prune_visibility, gaussian2d_utils.py:849-865 -- is compared on EVERY surfel
        contrib, nc_a, col_a, w_masked = raster.render_audit(saved, aud["lmax"], skip_px=torch.from_numpy(frag), want_weight=True)
        okf = ok.reshape(-1)
        np.testing.assert_array_equal(contrib.cpu().numpy()[okf], aud["contrib"][okf])
        assert torch.equal(nc_a, saved["n_contrib"]) and torch.equal(col_a, outs[0])      # the audit instantiation IS the product kernel
        from oracle import raster as orc
        w_ref, w_unc = orc.raster_weight(ref, frag)
        check_close(test, "weight.all_surfels", w_masked.cpu().numpy().astype(np.float64), w_ref, excluded=0, cond=np.zeros_like(w_ref), unc=w_unc)
    # R6 values
    color, radii, allmap, weight = [o.cpu().numpy() for o in outs]
    check_close(test, "color", color[:, ok], ref["out_color"][:, ok], excluded=nfr)
    for ch, nm in ((0, "depth"), (1, "alpha"), (2, "normal.x"), (3, "normal.y"), (4, "normal.z"), (5, "median")):
        check_close(test, "allmap." + nm, allmap[ch][ok], ref["allmap"][ch][ok], excluded=nfr)
    # distortion: sum_j w_j (m_j^2 A + M2 - 2 m_j M1) cancels catastrophically in fp32 in ANY implementation of the moment form.  Round 6 (VERDICT r5
    # item 6a): compared with the oracle's evaluation in DOUBLE (of the float code's own alphas and depths) under a DERIVED per-pixel bound --
    # |hip - dist64| <= 1e-4 |dist64| + bound, bound = the a-priori fp32 rounding of the three moment terms, of a ~3 ulp difference in m and of the
    # weight (orc_render_dist64) -- instead of a floor of the mean alpha; the plain error against the float oracle stays in the table as a record
    from oracle import raster as orc_
    d64, dbound = orc_.raster_dist64(ref)
    check_close(test, "allmap.dist", allmap[6][ok], d64[ok], excluded=nfr, cond=np.zeros_like(dbound[ok]), unc=dbound[ok], k_unc=1.0)
    record(test, "allmap.dist.oracle_f32_vs_f64_over_bound", float((np.abs(ref["allmap"][6][ok] - d64[ok]) / (1e-4 * np.abs(d64[ok]) + dbound[ok] + 1e-300)).max()),
           note="(the float oracle's own distance from the double evaluation, in units of the asserted bound: must be <= 1 too)")
    check_close(test, "final_T", saved["final_T"].cpu().numpy()[:, ok], ref["final_T"][:, ok], excluded=nfr)
    clean = ~aud["tainted"]
    check_close(test, "weight", weight[clean, 0], ref["weight"][clean], excluded=int(aud["tainted"].sum()))
    if aud["tainted"].any():        # (the product kernel's own weight on surfels that touch a fragile pixel: one flipped splat there moves it by O(1)
        d = np.abs(weight[~clean, 0] - ref["weight"][~clean])                      #  pixel -- bounded here, COMPARED above with those pixels left out)
        record(test, "weight.tainted", d.max(), "(absolute, %d surfels)" % int((~clean).sum()))
        assert d.max() <= 1.0 * max(1, nfr)


@pytest.mark.parametrize("case", CASES)
def test_forward_stages_vs_oracle(case, request):
    from envgs_amd import raster
    from oracle import raster as orc
    dev = torch.device("cuda:0")
    g, cam = small_scene(P=case["P"], H=case["H"], W=case["W"], seed=case["seed"], C=case["C"], sh=case["sh"],
                         scale_mul=case.get("scale_mul", 4.0))
    bg = torch.tensor([0.2, 0.5, 0.9])
    C = case["C"]
    mod = _mod_for(C)
    st = _settings(mod, cam, bg, case["deg"], dev)
    gd = {k: v.to(dev) for k, v in g.items()}
    outs, saved = raster.rasterize_forward(C, gd["means3D"], gd["shs"] if case["sh"] else None,
                                           None if case["sh"] else gd["colors_precomp"], gd["opacities"], gd["scales"],
                                           gd["rotations"], None, st, keep_binning=True)
    torch.cuda.synchronize()
    ref = _oracle(g, cam, bg, case["deg"], C, case["sh"])
    aud = orc.raster_audit(ref, want_contrib=True)
    _compare_forward(request.node.name, outs, saved, ref, aud, case["sh"], check_sets=True)


def _masked_upstream(C, H, W, seed, frag):
    gen = torch.Generator().manual_seed(seed)
    dcol = torch.randn(C, H, W, generator=gen) / (H * W)
    dall = torch.randn(7, H, W, generator=gen) / (H * W)
    m = torch.from_numpy(~frag)
    return dcol * m, dall * m          # zero upstream gradient at fragile pixels, for BOTH implementations


def _sh_clamp_fragile(ref):
    """Per-surfel audit of the colour clamp clamp_min(SH + 0.5, 0) (R1 / R8), now part of the oracle (orc_sh_clamp_audit): a colour channel
    within its own fp32 rounding of zero flips the clamp, and with it all 48 dL/dSH elements of the surfel.  Such surfels are left out of
    the dshs comparison (and counted)."""
    from oracle import raster as orc
    return orc.sh_clamp_audit(ref)


GRAD_NAMES = (("means3D", "dmeans3D"), ("scales", "dscales"), ("rotations", "drots"), ("opacities", "dopacities"), ("means2D", "dmeans2D"))


@pytest.mark.parametrize("case", CASES)
def test_backward_vs_oracle(case, request):
    from oracle import raster as orc
    dev = torch.device("cuda:0")
    C = case["C"]
    g, cam = small_scene(P=case["P"], H=case["H"], W=case["W"], seed=case["seed"], C=C, sh=case["sh"],
                         scale_mul=case.get("scale_mul", 4.0))
    bg = torch.tensor([0.2, 0.5, 0.9])
    mod = _mod_for(C)
    st = _settings(mod, cam, bg, case["deg"], dev)
    H, W = case["H"], case["W"]
    ref = _oracle(g, cam, bg, case["deg"], C, case["sh"])
    aud = orc.raster_audit(ref)
    dcol, dall = _masked_upstream(C, H, W, case["seed"] + 100, aud["fragile"])

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

    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
    test = request.node.name
    nfr = int(aud["fragile"].sum())
    grads = dict(leaves, means2D=means2D)
    clampfrag = _sh_clamp_fragile(ref) if case["sh"] else None
    for k_hip, k_ref in GRAD_NAMES + ((("shs", "dshs"),) if case["sh"] else (("colors_precomp", "dcolors"),)):
        check_close(test, k_ref, grads[k_hip].grad.cpu().numpy().reshape(rb[k_ref].shape), rb[k_ref], excluded=nfr, cond=rb["cond"][k_ref], unc=rb["unc"][k_ref],
                    keep=(~clampfrag if k_ref == "dshs" else None))

def test_backward_sparse_distortion_gradient():
    """R7 picks its distortion-free instantiation per TILE from a ballot of `dL/d(dist) != 0`.  The ballot must see all 256
    pixels: here the distortion map's upstream gradient is non-zero ONLY on pixels that are not the origin of an 8x8 quadrant (and is the
    only upstream gradient, so a dropped distortion term shows as a zero gradient)."""
    from oracle import raster as orc
    import diff_surfel_rasterization_wet_ch05 as mod
    dev = torch.device("cuda:0")
    C, H, W = 5, 64, 80
    g, cam = small_scene(P=500, H=H, W=W, seed=12, C=C, sh=False)
    bg = torch.tensor([0.2, 0.5, 0.9])
    st = _settings(mod, cam, bg, 0, dev)
    ref = _oracle(g, cam, bg, 0, C, False)
    aud = orc.raster_audit(ref)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    sparse = ((xx % 8 == 3) & (yy % 8 == 5)) & torch.from_numpy(~aud["fragile"])
    assert int(sparse.sum()) > 30 and not bool(sparse[::8, ::8].any())
    dcol = torch.zeros(C, H, W)
    dall = torch.zeros(7, H, W)
    dall[6] = torch.randn(H, W, generator=torch.Generator().manual_seed(3)) / (H * W) * sparse
    leaves = {k: g[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors_precomp")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True) + 0
    means2D.retain_grad()
    color, radii, allmap, weight = mod.GaussianRasterizer(raster_settings=st)(
        means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors_precomp"],
        opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    ((color * dcol.to(dev)).sum() + (allmap * dall.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
    assert float(np.abs(rb["dmeans3D"]).max()) > 0 and float(leaves["means3D"].grad.abs().max()) > 0
    grads = dict(leaves, means2D=means2D)
    for k_hip, k_ref in GRAD_NAMES:
        # the ONLY upstream gradient is the distortion map's: every term is the three-term cancellation M2 + m^2 A - 2 m M1 of the forward's saved
        # moments, whose `unc` is an a-priori ONE-ulp bound (orc_render_bwd_unc), not a realised error.  Round 6: asserted at the
        # suite-wide K_UNC like everything else (rounds 4-5: 16) -- the table's sensitivity columns carry 4 / 0 next to it
        check_close("sparse_distortion_gradient", k_ref, grads[k_hip].grad.cpu().numpy().reshape(rb[k_ref].shape), rb[k_ref],
                    excluded=int(aud["fragile"].sum()), cond=rb["cond"][k_ref], unc=rb["unc"][k_ref])


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
    ref = _oracle(g, cam, bg, 2, 3, True, precomp_T=tm)
    aud = orc.raster_audit(ref)
    dcol, dall = _masked_upstream(3, 64, 64, 5, aud["fragile"])
    ((color * dcol.to(dev)).sum() + (allmap * dall.to(dev)).sum()).backward()
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
    ok = ~aud["fragile"]
    check_close("precomputed_transmat", "color", color.detach().cpu().numpy()[:, ok], ref["out_color"][:, ok], excluded=int((~ok).sum()))
    check_close("precomputed_transmat", "dtransmat", tmd.grad.cpu().numpy(), rb["dtransmat_precomp"], cond=rb["cond"]["dtransmat_precomp"], unc=rb["unc"]["dtransmat_precomp"])
    check_close("precomputed_transmat", "dmeans3D", m3.grad.cpu().numpy(), rb["dmeans3D"], cond=rb["cond"]["dmeans3D"], unc=rb["unc"]["dmeans3D"])          # SH view-direction term only
    check_close("precomputed_transmat", "dshs", shs.grad.cpu().numpy(), rb["dshs"], cond=rb["cond"]["dshs"], unc=rb["unc"]["dshs"])


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


def test_contribution_masks_are_exact_and_optional():
    """The forward's per-instance quadrant masks (envgs_raster.h: contrib_mask) are exactly the contributor sets of the audit kernel,
    and the backward gives the same gradients whether it walks them or falls back to the geometric quadrant test (mask = NULL)."""
    from envgs_amd import raster
    import diff_surfel_rasterization_wet_ch05 as mod
    dev = torch.device("cuda:0")
    H, W, C = 90, 104, 5
    g, cam = small_scene(P=3000, H=H, W=W, seed=21, C=C, sh=False)
    bg = torch.rand(C)
    st = _settings(mod, cam, bg, 0, dev)
    gd = {k: v.to(dev) for k, v in g.items()}
    args = (C, gd["means3D"], None, gd["colors_precomp"], gd["opacities"], gd["scales"], gd["rotations"], None, st)
    outs, saved = raster.rasterize_forward(*args, keep_binning=True)
    N = saved["N"]
    r = saved["ranges"].cpu().numpy().view(np.uint32).astype(np.int64)
    lmax = int((r[:, 1] - r[:, 0]).max())
    contrib, ncon, _ = raster.render_audit(saved, lmax)
    contrib = contrib.cpu().numpy().reshape(H, W, lmax).astype(bool)
    mask = saved["contrib_mask"].cpu().numpy()[:N]
    last = ncon.cpu().numpy()[0]
    gx = (W + 15) // 16
    for tile in range(r.shape[0]):
        tx, ty = tile % gx, tile // gx
        n = int(r[tile, 1] - r[tile, 0])
        # entries the forward staged are written; what lies behind the tile's deepest last-contributor is never read by the backward
        deep = int(last[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].max())
        for q in range(4):
            y0, x0 = ty * 16 + (q >> 1) * 8, tx * 16 + (q & 1) * 8
            blk = contrib[y0:y0 + 8, x0:x0 + 8, :n].reshape(-1, n).any(0) if (y0 < H and x0 < W) else np.zeros(n, bool)
            got = ((mask[r[tile, 0]:r[tile, 0] + n] >> q) & 1).astype(bool)
            assert np.array_equal(got[:deep], blk[:deep]), (tile, q)
    dcol = torch.randn(C, H, W, device=dev); dall = torch.randn(7, H, W, device=dev); dall[6] = 0
    ga = raster.rasterize_backward(saved, dcol, dall)
    saved2 = dict(saved); saved2["contrib_mask"] = None
    gb = raster.rasterize_backward(saved2, dcol, dall)
    for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"):
        a, b = ga[k].cpu().numpy(), gb[k].cpu().numpy()
        assert np.abs(a - b).max() <= 2e-5 * (np.abs(b).max() + 1e-30), k     # same terms, summed by LDS / L2 atomics in a different order
    raster.CONTRIB_MASK["on"] = False
    try:
        _, saved3 = raster.rasterize_forward(*args)
        assert saved3["contrib_mask"] is None
        gc = raster.rasterize_backward(saved3, dcol, dall)
        assert np.abs(gc["means3D"].cpu().numpy() - gb["means3D"].cpu().numpy()).max() <= 2e-5 * np.abs(gb["means3D"].cpu().numpy()).max()
    finally:
        raster.CONTRIB_MASK["on"] = True


def test_speculative_instance_count_is_exact_and_recovers_from_a_small_guess():
    """rasterize_forward sizes the binning buffers from the PREVIOUS call's instance count and reads this call's count only after R3-R6 are
    queued (no host sync in front of them).  A capacity above the count must give bit-identical results (padding keys sort last), a capacity
    below it must be noticed and repaired."""
    from envgs_amd import raster
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    H, W = 128, 144
    g, cam = small_scene(P=4000, H=H, W=W, seed=31, C=3, sh=True)
    st = _settings(mod, cam, torch.rand(3), 3, dev)
    gd = {k: v.to(dev) for k, v in g.items()}
    args = (3, gd["means3D"], gd["shs"], None, gd["opacities"], gd["scales"], gd["rotations"], None, st)
    key = (dev.index, 4000, H, W)

    def run(guess):
        if guess is None: raster._N_GUESS.pop(key, None)
        else: raster._N_GUESS[key] = guess
        outs, saved = raster.rasterize_forward(*args, keep_binning=True)
        torch.cuda.synchronize()
        N = saved["N"]
        return N, [saved["keys_sorted"][:N].cpu(), saved["point_list"][:N].cpu(), saved["ranges"].cpu(), saved["n_contrib"].cpu(),
                   outs[0].cpu(), outs[2].cpu(), saved["final_T"].cpu(), saved["contrib_mask"][:N].cpu()], saved
    N0, ref, _ = run(None)                                  # first call of a shape: waits for the count (exact buffers)
    assert N0 > 10000
    misses = raster.LAST_STATS.get("n_guess_misses", 0)
    N1, pad, saved = run(N0 + 70000)                         # capacity above the count: the tail of the N-sized buffers stays unused
    assert N1 == N0 and saved["point_list"].numel() == N0 + 70000 and raster.LAST_STATS.get("n_guess_misses", 0) == misses
    for a, b in zip(ref[:4], pad[:4]): assert torch.equal(a, b)
    for a, b in zip(ref[4:7], pad[4:7]): assert torch.equal(a, b)           # same kernels on the same lists: bit-identical images
    gb = raster.rasterize_backward(saved, torch.ones(3, H, W, device=dev), torch.zeros(7, H, W, device=dev))
    assert torch.isfinite(gb["means3D"]).all() and float(gb["means3D"].abs().max()) > 0
    N2, rep, saved2 = run(N0 // 2)                          # capacity below the count: detected, R3-R6 repeated with the exact size
    assert N2 == N0 and raster.LAST_STATS.get("n_guess_misses", 0) == misses + 1 and saved2["point_list"].numel() == N0
    for a, b in zip(ref[:7], rep[:7]): assert torch.equal(a, b)
    assert raster._N_GUESS[key] >= N0                        # and the next call starts from a sufficient capacity


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
    aud = orc.raster_audit(ref, want_contrib=True)             # contributor sets too: 640 k pixels x the longest tile list (~1.2 GB of flags per side)
    test = "full_size_300k_800x800"
    _compare_forward(test, outs, saved, ref, aud, True, check_sets=True)          # no tail allowance: EVERY element within tolerance
    # gradients: upstream zeroed at the fragile pixels for both implementations
    dcol, dall = _masked_upstream(3, H, W, 1, aud["fragile"])
    grads = raster.rasterize_backward(saved, dcol.to(dev), dall.to(dev))
    torch.cuda.synchronize()
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
    clampfrag = _sh_clamp_fragile(ref)
    record(test, "sh_clamp_fragile_surfels", float(clampfrag.mean()), "(%d of %d surfels)" % (int(clampfrag.sum()), clampfrag.size))
    for k_hip, k_ref in GRAD_NAMES + (("shs", "dshs"),):
        check_close(test, k_ref, grads[k_hip].cpu().numpy().reshape(rb[k_ref].shape), rb[k_ref], excluded=int(aud["fragile"].sum()), cond=rb["cond"][k_ref], unc=rb["unc"][k_ref],
                    keep=(~clampfrag if k_ref == "dshs" else None), k_unc=0.0)          # full size: NO measured-uncertainty term (tests/util.py: K_UNC)


def test_exact_math_attribution():
    """VERDICT r3 item 2, the attribution: what separates the HIP compositing kernels from the oracle's float code, measured WITHOUT any
    uncertainty floor (K_UNC = 0 column) on a mid-size scene, in the two arithmetic forms the diagnostic library carries:
      approx : the product kernels -- canonical operation order, 1/p.z = v_rcp_f32 + one Newton step, exp = v_exp_f32(x log2 e), other reciprocals v_rcp_f32
      exact  : the same kernels with IEEE divisions and the library expf (envgs_debug_set(ENVGS_DBG_RASTER_EXACT, 1))
    Rounds 1-3 (no canonical order: the compiler's FMA contraction against the oracle's uncontracted statements) are the r03 rows of
    profiles/r03_parity_errors.txt.  Both forms must meet the contract; the recorded maxima are the attribution."""
    from envgs_amd import raster, synth, _lib
    from oracle import raster as orc
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    P, H, W = 40000, 400, 400
    g = synth.base_gaussians(P, seed=5)
    g["scales"] = g["scales"] * 1.5
    cam = synth.orbit_camera(2, H=H, W=W, fx=1111.1 * W / 800.0)
    bg = torch.ones(3)
    st = _settings(mod, cam, bg, 3, dev)
    ca = cam_args(cam)
    ref = orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), ca["viewmatrix"].numpy(), ca["projmatrix"].numpy(),
                             ca["campos"].numpy(), W, H, scales=g["scales"].numpy(), rotations=g["rotations"].numpy(),
                             shs=g["shs"].numpy(), sh_degree=3, bg=bg.numpy())
    aud = orc.raster_audit(ref)
    ok = ~aud["fragile"]; nfr = int(aud["fragile"].sum())
    dcol, dall = _masked_upstream(3, H, W, 1, aud["fragile"])
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
    clampfrag = _sh_clamp_fragile(ref)
    gd = {k: v.to(dev) for k, v in g.items()}
    old = _lib.select("diag")
    try:
        lib = _lib.load()
        for form, sw in (("approx", 0), ("exact", 1)):
            lib.envgs_debug_set(3, sw)                      # ENVGS_DBG_RASTER_EXACT
            outs, saved = raster.rasterize_forward(3, gd["means3D"], gd["shs"], None, gd["opacities"], gd["scales"], gd["rotations"], None, st, keep_binning=True)
            grads = raster.rasterize_backward(saved, dcol.to(dev), dall.to(dev))
            torch.cuda.synchronize()
            test = "attribution." + form
            record_fragile(test, "fragile_px", aud["fragile"], FRAGILE_PX_MAX, "(by the round-1..3 definition: %d)" % int(aud["legacy_fragile"].sum()))
            nc = saved["n_contrib"].cpu().numpy()
            np.testing.assert_array_equal(nc[0][ok], ref["n_contrib"][0][ok])
            np.testing.assert_array_equal(nc[1][ok], ref["n_contrib"][1][ok])
            color, _, allmap, _ = [o.cpu().numpy() for o in outs]
            check_close(test, "color", color[:, ok], ref["out_color"][:, ok], excluded=nfr)
            for ch, nm in ((0, "depth"), (1, "alpha"), (2, "normal.x"), (3, "normal.y"), (4, "normal.z"), (5, "median")):
                check_close(test, "allmap." + nm, allmap[ch][ok], ref["allmap"][ch][ok], excluded=nfr)
            check_close(test, "final_T", saved["final_T"].cpu().numpy()[:, ok], ref["final_T"][:, ok], excluded=nfr)
            for k_hip, k_ref in GRAD_NAMES + (("shs", "dshs"),):
                check_close(test, k_ref, grads[k_hip].cpu().numpy().reshape(rb[k_ref].shape), rb[k_ref], excluded=nfr, cond=rb["cond"][k_ref], unc=rb["unc"][k_ref],
                            keep=(~clampfrag if k_ref == "dshs" else None))
    finally:
        _lib.load().envgs_debug_set(3, 0)
        _lib.select(old)
