"""GPU parity: the HIP LBVH tracer (through the C-ABI and the drop-in SurfelTracer) against the brute-force
CPU oracle.  Hit sets are validated implicitly: identical composited sums and per-surfel weights for every ray."""
import numpy as np
import pytest
import torch

from envgs_amd import synth
from tests.test_oracle_trace import trace_scene
from tests.util import rel_err, assert_close_frac

pytestmark = pytest.mark.gpu


def _settings(mod, bg, deg, dev, depth=0, thr=0.0, H=1, W=1):
    I = torch.eye(4, device=dev)
    return mod.SurfelTracingSettings(image_height=H, image_width=W, tanfovx=1.0, tanfovy=1.0, bg=bg.to(dev), scale_modifier=1.0,
                                     viewmatrix=I, projmatrix=I, sh_degree=torch.tensor([deg], device=dev), campos=torch.zeros(3, device=dev),
                                     prefiltered=False, debug=False, max_trace_depth=depth, specular_threshold=thr)


def _run_hip(g, ro, rd, bg, deg, use_sh, sff, grads=None, depth=0, thr=0.0, shape=None):
    import diff_surfel_tracing as mod
    dev = torch.device("cuda:0")
    L = {k: g[k].to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "others")}
    if use_sh: L["shs"] = g["shs"].to(dev).requires_grad_(True)
    else: L["colors_precomp"] = g["colors_precomp"].to(dev).requires_grad_(True)
    o = ro.to(dev).requires_grad_(True); d = rd.to(dev).requires_grad_(True)
    oo, dd = (o, d) if shape is None else (o.reshape(shape + (3,)), d.reshape(shape + (3,)))
    v, f = synth.get_disks(L["means3D"].detach(), L["scales"].detach(), L["rotations"].detach())
    tracer = mod.SurfelTracer()
    tracer.build_acceleration_structure(v.detach().clone(), f.detach().clone(), rebuild=True)
    g3 = torch.zeros_like(L["means3D"], requires_grad=True) + 0
    g3.retain_grad()
    outs = tracer(oo, dd, v, means3D=L["means3D"], grads3D=g3, shs=L.get("shs"), colors_precomp=L.get("colors_precomp"),
                  others_precomp=L["others"], opacities=L["opacities"], scales=L["scales"], rotations=L["rotations"],
                  cov3D_precomp=None, tracer_settings=_settings(mod, bg, deg, dev, depth, thr), start_from_first=sff)
    if grads is not None:
        rgb, dpt, acc, norm, dist, aux, mid, wet = outs
        R = ro.shape[0]
        loss = sum((x.reshape(R, -1) * y.to(dev).reshape(R, -1)).sum() for x, y in zip((rgb, dpt, acc, norm, aux), grads))
        loss.backward()
    torch.cuda.synchronize()
    return outs, L, o, d, g3


@pytest.mark.parametrize("use_sh,camera,deg,P,R", [(True, True, 3, 150, 400), (False, False, 0, 150, 400), (True, False, 2, 2000, 1024),
                                                   (True, False, 1, 1, 64), (False, True, 0, 40, 130)])
def test_trace_forward_backward_vs_oracle(use_sh, camera, deg, P, R):
    from oracle import trace as otr
    g, ro, rd = trace_scene(P=P, R=R, seed=7, camera=camera)
    if P > 500:
        g["scales"] = g["scales"] * 0.35                       # many small surfels: deep tree, > K hits per ray for some
    R = ro.shape[0]
    bg = torch.tensor([0.3, 0.1, 0.7])
    gen = torch.Generator().manual_seed(9)
    gr = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen),
          torch.randn(R, 3, generator=gen), torch.randn(R, 2, generator=gen)]
    outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, deg, use_sh, camera, grads=gr)
    rgb, dpt, acc, norm, dist, aux, mid, wet = [x.detach().cpu().numpy() for x in outs]
    ckw = dict(shs=g["shs"].numpy(), sh_degree=deg) if use_sh else dict(colors_precomp=g["colors_precomp"].numpy())
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                            g["opacities"].numpy(), others=g["others"].numpy(), bg=bg.numpy(), start_from_first=camera, **ckw)
    if P > 1: assert ref["nhits"].mean() > 1
    for a, b, nm in ((rgb, ref["rgb"], "rgb"), (dpt[:, 0], ref["dpt"], "dpt"), (acc[:, 0], ref["acc"], "acc"), (norm, ref["norm"], "norm"),
                     (aux, ref["aux"], "aux"), (wet[:, 0], ref["wet"], "wet")):
        assert_close_frac(a, b, 1e-4, max_bad_frac=2e-3, flip_bound=0.05, what=nm)
    assert_close_frac(dist[:, 0], ref["dist"], 5e-3, max_bad_frac=2e-3, what="dist")
    np.testing.assert_allclose(mid[:, 0:3], ro.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(mid[:, 13:16], rgb, rtol=1e-6, atol=1e-7)

    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    tol = 1e-3
    chk = lambda a, b, nm: assert_close_frac(a, b, tol, max_bad_frac=5e-3, flip_bound=0.2, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(g3.grad.cpu().numpy(), rb["dmeans3D"], "grads3D")
    chk(L["scales"].grad.cpu().numpy(), rb["dscales"], "dscales")
    chk(L["rotations"].grad.cpu().numpy(), rb["drots"], "drots")
    chk(L["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"], "dopac")
    chk(L["others"].grad.cpu().numpy(), rb["dothers"], "dothers")
    if use_sh: chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    else: chk(L["colors_precomp"].grad.cpu().numpy(), rb["dcolors"], "dcolors")
    chk(o.grad.cpu().numpy(), rb["dray_o"], "dray_o")
    chk(d.grad.cpu().numpy(), rb["dray_d"], "dray_d")


@pytest.mark.parametrize("stage_lists", [True, False])
def test_trace_bounces_and_image_shaped_rays(stage_lists):
    """Two specular bounces: as one list-path trace per stage (the default) and inside the K-buffer kernel -- same images, same per-stage
    `mid` records, same (stage-0) gradients as the oracle; the per-surfel weight comes from stage 0 only."""
    from oracle import trace as otr
    from envgs_amd import tracing
    g, ro, rd = trace_scene(P=150, R=400, seed=4, camera=True)       # 20x20 camera rays
    R = ro.shape[0]
    bg = torch.tensor([0.0, 0.0, 0.0])
    gen = torch.Generator().manual_seed(12)
    gr = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen), torch.randn(R, 3, generator=gen),
          torch.randn(R, 2, generator=gen)]
    old = tracing.BOUNCE_LISTS["on"]
    try:
        tracing.BOUNCE_LISTS["on"] = stage_lists
        outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, 1, True, True, grads=gr, depth=2, thr=0.1, shape=(20, 20))
    finally:
        tracing.BOUNCE_LISTS["on"] = old
    rgb, dpt, acc, norm, dist, aux, mid, wet = outs
    assert rgb.shape == (20, 20, 3) and dpt.shape == (20, 20, 1) and mid.shape == (20, 20, 48) and wet.shape == (150, 1)
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                            g["opacities"].numpy(), shs=g["shs"].numpy(), sh_degree=1, others=g["others"].numpy(), bg=bg.numpy(),
                            max_trace_depth=2, specular_threshold=0.1, start_from_first=True)
    assert (ref["mid"][:, 16 + 7] != 0).any() and (ref["mid"][:, 32 + 7] != 0).any()          # both bounce stages really ran for some rays
    assert_close_frac(rgb.detach().reshape(-1, 3).cpu().numpy(), ref["rgb"], 2e-4, max_bad_frac=5e-3, flip_bound=0.1, what="rgb")
    assert_close_frac(mid.detach().reshape(-1, 48).cpu().numpy(), ref["mid"], 2e-4, max_bad_frac=5e-3, flip_bound=0.2, what="mid")
    assert_close_frac(wet.detach().cpu().numpy()[:, 0], ref["wet"], 2e-4, max_bad_frac=5e-3, flip_bound=0.1, what="wet")
    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    chk = lambda a, b, nm: assert_close_frac(a, b, 1e-3, max_bad_frac=5e-3, flip_bound=0.3, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    chk(L["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"], "dopac")
    chk(o.grad.cpu().numpy(), rb["dray_o"], "dray_o")


def test_trace_edge_cases():
    import diff_surfel_tracing as mod
    dev = torch.device("cuda:0")
    g, ro, rd = trace_scene(P=50, R=64, seed=2, camera=False)
    bg = torch.tensor([0.2, 0.4, 0.6])
    # rays that miss everything: background, zero weight, zero gradients
    far = ro + torch.tensor([1000.0, 0, 0]); away = torch.tensor([1.0, 0, 0]).expand_as(rd).contiguous()
    outs, L, o, d, g3 = _run_hip(g, far, away, bg, 2, True, False, grads=[torch.ones(64, 3), torch.ones(64), torch.ones(64), torch.ones(64, 3), torch.ones(64, 2)])
    rgb, dpt, acc, norm, dist, aux, mid, wet = outs
    assert torch.allclose(rgb, bg.to(dev).expand_as(rgb)) and float(acc.abs().max()) == 0 and float(wet.abs().max()) == 0
    assert float(L["means3D"].grad.abs().max()) == 0 and float(o.grad.abs().max()) == 0
    # validation mirrors the reference wrapper; stale BVH after "densification" is an error, not silent garbage
    tracer = mod.SurfelTracer()
    v, f = synth.get_disks(g["means3D"], g["scales"], g["rotations"])
    tracer.build_acceleration_structure(v.to(dev), f.to(dev), rebuild=True)
    kw = dict(means3D=g["means3D"].to(dev), grads3D=None, shs=g["shs"].to(dev), colors_precomp=None, others_precomp=None,
              opacities=g["opacities"].to(dev), scales=g["scales"].to(dev), rotations=g["rotations"].to(dev), cov3D_precomp=None,
              tracer_settings=_settings(mod, bg, 0, dev), start_from_first=False)
    with pytest.raises(Exception):
        tracer(ro.to(dev), rd.to(dev), None, **{**kw, "colors_precomp": torch.rand(50, 3, device=dev)})
    with pytest.raises(Exception):
        tracer(ro.to(dev), rd.to(dev), None, **{**kw, "means3D": torch.rand(60, 3, device=dev)})
    # cached BVH (v=None at test time, optix_utils.py:83) under inference_mode, (1,S,3) rays
    with torch.inference_mode():
        rgb2, *_ = tracer(ro.to(dev)[None], rd.to(dev)[None], None, **kw)
    assert rgb2.shape == (1, 64, 3) and torch.isfinite(rgb2).all()


@pytest.mark.parametrize("force_cap,records", [(12, True), (20, False), (0, True), (512, False)])
def test_trace_list_path_overflow_handoff(force_cap, records):
    """Per-ray hit lists with a tiny capacity: rays that overflow must be handed to the K-buffer kernels and give the same
    result as the oracle (forward and backward); force_cap=0 disables the list path entirely."""
    from envgs_amd import tracing
    from oracle import trace as otr
    g, ro, rd = trace_scene(P=600, R=512, seed=11, camera=False)
    g["scales"] = g["scales"] * 0.6
    R = ro.shape[0]
    bg = torch.tensor([0.1, 0.2, 0.3])
    gen = torch.Generator().manual_seed(3)
    gr = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen),
          torch.randn(R, 3, generator=gen), torch.randn(R, 2, generator=gen)]
    old = dict(tracing.HIT_CAP)
    orig_fwd = tracing.trace_forward
    old_rec = tracing.USE_RECORDS["on"]
    try:
        tracing.USE_RECORDS["on"] = records
        if force_cap: tracing.HIT_CAP["force"] = force_cap
        else: tracing.trace_forward = lambda *a, **k: orig_fwd(*a, **{**k, "use_lists": False})
        outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, 3, True, False, grads=gr)
        cnt = tracing.last_trace_counts()
    finally:
        tracing.HIT_CAP.clear(); tracing.HIT_CAP.update(old); tracing.trace_forward = orig_fwd; tracing.USE_RECORDS["on"] = old_rec
    rgb, dpt, acc, norm, dist, aux, mid, wet = [x.detach().cpu().numpy() for x in outs]
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                            g["opacities"].numpy(), shs=g["shs"].numpy(), sh_degree=3, others=g["others"].numpy(), bg=bg.numpy(),
                            start_from_first=False)
    if force_cap and force_cap < 100:
        assert cnt["max_list"] > force_cap                 # the overflow path really ran
        assert (ref["nhits"] <= force_cap).any()           # and so did the list path
    assert_close_frac(rgb, ref["rgb"], 1e-4, max_bad_frac=2e-3, flip_bound=0.05, what="rgb")
    assert_close_frac(wet[:, 0], ref["wet"], 1e-4, max_bad_frac=2e-3, flip_bound=0.05, what="wet")
    assert_close_frac(mid[:, 13:16], ref["rgb"], 1e-4, max_bad_frac=2e-3, flip_bound=0.05, what="mid.rgb")
    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    chk = lambda a, b, nm: assert_close_frac(a, b, 1e-3, max_bad_frac=5e-3, flip_bound=0.2, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    chk(L["rotations"].grad.cpu().numpy(), rb["drots"], "drots")
    chk(o.grad.cpu().numpy(), rb["dray_o"], "dray_o")
    chk(d.grad.cpu().numpy(), rb["dray_d"], "dray_d")


def test_trace_baseline_size_env_set_sample_vs_oracle():
    """BASELINE env set at full size (163 840 surfels over +-50, the reference's initial fog) traced by a 3 072-ray sample of
    reflected-like rays: deep LBVH (stack spill path), long hit lists (termination bound, list sort), all gradients -- against the
    brute-force oracle.  The whole 640 k-ray view is covered by bench.py; the oracle needs ~10 s for this sample."""
    from oracle import trace as otr
    from envgs_amd import tracing
    dev = torch.device("cuda:0")
    P, R = 163840, 3072
    e = synth.env_gaussians(P, seed=1)
    gen = torch.Generator().manual_seed(5)
    ro = (torch.rand(R, 3, generator=gen) * 2 - 1) * 1.3
    rd = torch.randn(R, 3, generator=gen); rd = rd / rd.norm(dim=-1, keepdim=True) * (0.8 + 0.4 * torch.rand(R, 1, generator=gen))
    g = dict(means3D=e["means3D"], scales=e["scales"], rotations=e["rotations"], opacities=e["opacities"], shs=e["shs"],
             others=torch.rand(P, 2, generator=gen), colors_precomp=torch.rand(P, 3, generator=gen))
    bg = torch.tensor([0.0, 0.0, 0.0])
    gr = [torch.randn(R, 3, generator=gen) / R, torch.zeros(R), torch.zeros(R), torch.zeros(R, 3), torch.zeros(R, 2)]
    outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, 3, True, False, grads=gr)
    cnt = tracing.last_trace_counts()
    assert cnt["hits"] / R > 30 and cnt["max_list"] > 100            # a fog: long lists
    rgb, dpt, acc, norm, dist, aux, mid, wet = [x.detach().cpu().numpy() for x in outs]
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                            g["opacities"].numpy(), shs=g["shs"].numpy(), sh_degree=3, others=g["others"].numpy(), bg=bg.numpy(),
                            start_from_first=False)
    assert abs(int(ref["nhits"].sum()) - cnt["hits"]) <= 1e-3 * cnt["hits"]      # same composited hits (threshold flips aside)
    for a, b, nm in ((rgb, ref["rgb"], "rgb"), (dpt[:, 0], ref["dpt"], "dpt"), (acc[:, 0], ref["acc"], "acc"), (norm, ref["norm"], "norm"),
                     (wet[:, 0], ref["wet"], "wet")):
        assert_close_frac(a, b, 2e-4, max_bad_frac=2e-3, flip_bound=0.05, what=nm)
    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    chk = lambda a, b, nm: assert_close_frac(a, b, 1e-3, max_bad_frac=2e-3, flip_bound=0.3, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(L["scales"].grad.cpu().numpy(), rb["dscales"], "dscales")
    chk(L["rotations"].grad.cpu().numpy(), rb["drots"], "drots")
    chk(L["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"], "dopac")
    chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    chk(o.grad.cpu().numpy(), rb["dray_o"], "dray_o")
    chk(d.grad.cpu().numpy(), rb["dray_d"], "dray_d")


def test_trace_clustered_surfels_deep_tree():
    """Thousands of surfels packed into a tiny cluster (identical Morton prefixes -> a very deep LBVH: exercises the HBM stack-spill
    path of the collection pass and the index tie-break of the Karras build) plus a sparse far set; also exact ties in t (coplanar
    surfels) ordered by surfel id."""
    from oracle import trace as otr
    gen = torch.Generator().manual_seed(21)
    Pc, Pf = 3000, 200
    means = torch.cat([torch.tensor([0.0, 0.0, 5.0]) + 0.02 * torch.randn(Pc, 3, generator=gen), (torch.rand(Pf, 3, generator=gen) * 2 - 1) * 30])
    means[:50] = means[0]                                              # 50 exactly coincident centres
    P = Pc + Pf
    scales = torch.cat([0.3 + 0.3 * torch.rand(Pc, 2, generator=gen), 2 + 2 * torch.rand(Pf, 2, generator=gen)])
    q = torch.randn(P, 4, generator=gen); q[:50] = q[0]                # ... and coplanar: identical t for every ray
    rots = q / q.norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=gen) - 2.5)       # faint: long lists through the cluster
    g = dict(means3D=means, scales=scales, rotations=rots, opacities=opac, shs=torch.randn(P, 16, 3, generator=gen) * 0.3,
             others=torch.rand(P, 2, generator=gen), colors_precomp=torch.rand(P, 3, generator=gen))
    R = 512
    ro = torch.randn(R, 3, generator=gen) * 0.2
    tgt = torch.tensor([0.0, 0.0, 5.0]) + 0.3 * torch.randn(R, 3, generator=gen)
    rd = tgt - ro; rd = rd / rd.norm(dim=-1, keepdim=True)
    bg = torch.tensor([0.2, 0.2, 0.2])
    gr = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen), torch.randn(R, 3, generator=gen), torch.zeros(R, 2)]
    outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, 2, True, False, grads=gr)
    rgb, dpt, acc, norm, dist, aux, mid, wet = [x.detach().cpu().numpy() for x in outs]
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), means.numpy(), scales.numpy(), rots.numpy(), opac.numpy(), shs=g["shs"].numpy(), sh_degree=2,
                            others=g["others"].numpy(), bg=bg.numpy(), start_from_first=False)
    assert ref["nhits"].max() > 200
    for a, b, nm in ((rgb, ref["rgb"], "rgb"), (dpt[:, 0], ref["dpt"], "dpt"), (acc[:, 0], ref["acc"], "acc"), (wet[:, 0], ref["wet"], "wet")):
        assert_close_frac(a, b, 2e-4, max_bad_frac=5e-3, flip_bound=0.1, what=nm)
    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    chk = lambda a, b, nm: assert_close_frac(a, b, 2e-3, max_bad_frac=5e-3, flip_bound=0.3, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(L["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"], "dopac")
    chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    chk(d.grad.cpu().numpy(), rb["dray_d"], "dray_d")


@pytest.mark.parametrize("sort_rays", [True, False])
def test_trace_batch_table_overflow_and_unsorted_rays(sort_rays):
    """Incoherent rays through a dense set: a 64-ray batch blends far more distinct surfels than its 1024-slot merge table holds, so part
    of the hits become single entries (filed from the top of the batch's region) -- forward weights and every gradient must still match
    the oracle; with the coherence sort disabled the per-ray collection kernel feeds the same batch kernels."""
    from oracle import trace as otr
    from envgs_amd import tracing
    gen = torch.Generator().manual_seed(33)
    P, R = 6000, 1000                                                   # R is not a multiple of 64: a ragged last batch
    means = (torch.rand(P, 3, generator=gen) * 2 - 1) * 2.0
    scales = 0.12 + 0.1 * torch.rand(P, 2, generator=gen)
    q = torch.randn(P, 4, generator=gen)
    g = dict(means3D=means, scales=scales, rotations=q / q.norm(dim=-1, keepdim=True), opacities=torch.sigmoid(torch.randn(P, 1, generator=gen) - 2.0),
             shs=torch.randn(P, 16, 3, generator=gen) * 0.3, others=torch.rand(P, 2, generator=gen), colors_precomp=torch.rand(P, 3, generator=gen))
    ro = (torch.rand(R, 3, generator=gen) * 2 - 1) * 2.0
    rd = torch.randn(R, 3, generator=gen); rd = rd / rd.norm(dim=-1, keepdim=True)
    bg = torch.tensor([0.1, 0.0, 0.2])
    gr = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen), torch.randn(R, 3, generator=gen),
          torch.randn(R, 2, generator=gen)]
    old = tracing.SORT_RAYS["on"]
    try:
        tracing.SORT_RAYS["on"] = sort_rays
        outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, 3, True, False, grads=gr)
        table, singles = tracing.last_entry_counts()
        cnt = tracing.last_trace_counts()
    finally:
        tracing.SORT_RAYS["on"] = old
    assert table > 0 and singles > 0, (table, singles)                 # both kinds of entries were produced
    assert table + singles <= cnt["hits"]
    rgb, dpt, acc, norm, dist, aux, mid, wet = [x.detach().cpu().numpy() for x in outs]
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), g["opacities"].numpy(),
                            shs=g["shs"].numpy(), sh_degree=3, others=g["others"].numpy(), bg=bg.numpy(), start_from_first=False)
    for a, b, nm in ((rgb, ref["rgb"], "rgb"), (dpt[:, 0], ref["dpt"], "dpt"), (acc[:, 0], ref["acc"], "acc"), (aux, ref["aux"], "aux"),
                     (wet[:, 0], ref["wet"], "wet")):
        assert_close_frac(a, b, 2e-4, max_bad_frac=2e-3, flip_bound=0.05, what=nm)
    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    chk = lambda a, b, nm: assert_close_frac(a, b, 1e-3, max_bad_frac=5e-3, flip_bound=0.3, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(L["scales"].grad.cpu().numpy(), rb["dscales"], "dscales")
    chk(L["rotations"].grad.cpu().numpy(), rb["drots"], "drots")
    chk(L["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"], "dopac")
    chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    chk(L["others"].grad.cpu().numpy(), rb["dothers"], "dothers")
    chk(o.grad.cpu().numpy(), rb["dray_o"], "dray_o")
    chk(d.grad.cpu().numpy(), rb["dray_d"], "dray_d")


def test_trace_two_segment_forward_pipeline_vs_oracle():
    """Enough rays (>= 512 batches) for the forward to run as TWO batch segments on two streams (collect -> sort+composite -> register each):
    the joined result -- images, per-surfel weights, every gradient through the per-batch entries of both segments -- against the oracle."""
    from oracle import trace as otr
    from envgs_amd import tracing
    g, _, _ = trace_scene(P=1500, R=4, seed=17, camera=False)
    g["scales"] = g["scales"] * 0.5
    cam = synth.orbit_camera(3, H=192, W=192, fx=160.0, radius=1.0)                 # 36 864 coherent rays = 576 batches
    ro, rd = synth.get_rays(cam)
    ro, rd = ro.reshape(-1, 3).contiguous(), rd.reshape(-1, 3).contiguous()
    R = ro.shape[0]
    assert (R + 63) // 64 >= 512
    bg = torch.tensor([0.2, 0.3, 0.1])
    gen = torch.Generator().manual_seed(4)
    gr = [torch.randn(R, 3, generator=gen) / R, torch.randn(R, generator=gen) / R, torch.randn(R, generator=gen) / R, torch.randn(R, 3, generator=gen) / R,
          torch.randn(R, 2, generator=gen) / R]
    outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, 3, True, False, grads=gr)
    table, singles = tracing.last_entry_counts()
    assert table > 0
    rgb, dpt, acc, norm, dist, aux, mid, wet = [x.detach().cpu().numpy() for x in outs]
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), g["opacities"].numpy(),
                            shs=g["shs"].numpy(), sh_degree=3, others=g["others"].numpy(), bg=bg.numpy(), start_from_first=False)
    for a, b, nm in ((rgb, ref["rgb"], "rgb"), (dpt[:, 0], ref["dpt"], "dpt"), (acc[:, 0], ref["acc"], "acc"), (norm, ref["norm"], "norm"),
                     (aux, ref["aux"], "aux"), (wet[:, 0], ref["wet"], "wet")):
        assert_close_frac(a, b, 2e-4, max_bad_frac=2e-3, flip_bound=0.05, what=nm)
    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    chk = lambda a, b, nm: assert_close_frac(a, b, 1e-3, max_bad_frac=5e-3, flip_bound=0.3, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(L["scales"].grad.cpu().numpy(), rb["dscales"], "dscales")
    chk(L["rotations"].grad.cpu().numpy(), rb["drots"], "drots")
    chk(L["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"], "dopac")
    chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    chk(o.grad.cpu().numpy(), rb["dray_o"], "dray_o")
    chk(d.grad.cpu().numpy(), rb["dray_d"], "dray_d")


@pytest.mark.parametrize("force_cap", [1024, 512, 320])
def test_trace_very_long_lists(force_cap):
    """A stack of 900 faint sheets: most rays blend several hundred hits before they terminate, so the sort / composite pass for lists beyond
    256 entries runs (8 and 16 keys per lane), and with the smaller capacities the longest rays overflow into the K-buffer path."""
    from oracle import trace as otr
    from envgs_amd import tracing
    P, R = 900, 192
    gen = torch.Generator().manual_seed(21)
    means = torch.stack([torch.randn(P, generator=gen) * 0.15, torch.randn(P, generator=gen) * 0.15, 2.0 + torch.arange(P) * 0.01], dim=1)
    q = torch.tensor([1.0, 0, 0, 0]).expand(P, 4) + torch.randn(P, 4, generator=gen) * 0.05
    g = dict(means3D=means, scales=torch.full((P, 2), 0.3) + 0.3 * torch.rand(P, 2, generator=gen), rotations=q / q.norm(dim=-1, keepdim=True),
             opacities=0.02 + 0.04 * torch.rand(P, 1, generator=gen), shs=torch.randn(P, 16, 3, generator=gen) * 0.3,
             others=torch.rand(P, 2, generator=gen))
    # rays through the middle of the stack terminate after a few hundred hits, rays near its rim see falloffs below 1/255: 14 .. 712 hits
    ro = torch.cat([(torch.rand(R, 2, generator=gen) - 0.5) * 1.6, torch.zeros(R, 1)], dim=1)
    rd = torch.cat([torch.randn(R, 2, generator=gen) * 0.03, torch.ones(R, 1)], dim=1)
    bg = torch.tensor([0.1, 0.2, 0.3])
    gr = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen),
          torch.randn(R, 3, generator=gen), torch.randn(R, 2, generator=gen)]
    old = dict(tracing.HIT_CAP)
    try:
        tracing.HIT_CAP["force"] = force_cap
        outs, L, o, d, g3 = _run_hip(g, ro, rd, bg, 2, True, False, grads=gr)
    finally:
        tracing.HIT_CAP.clear(); tracing.HIT_CAP.update(old)
    cnt = tracing.last_trace_counts()
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), g["opacities"].numpy(),
                            shs=g["shs"].numpy(), sh_degree=2, others=g["others"].numpy(), bg=bg.numpy(), start_from_first=False)
    assert ref["nhits"].max() > 512 and ((ref["nhits"] > 256) & (ref["nhits"] <= 512)).sum() > 20 and (ref["nhits"] <= 256).sum() > 5 and cnt["max_list"] > 512
    rgb, dpt, acc, norm, dist, aux, mid, wet = [x.detach().cpu().numpy() for x in outs]
    for a, b, nm in ((rgb, ref["rgb"], "rgb"), (dpt[:, 0], ref["dpt"], "dpt"), (acc[:, 0], ref["acc"], "acc"), (norm, ref["norm"], "norm"),
                     (aux, ref["aux"], "aux"), (wet[:, 0], ref["wet"], "wet")):
        assert_close_frac(a, b, 3e-4, max_bad_frac=5e-3, flip_bound=0.05, what=nm)
    rb = otr.trace_backward(ref, *[x.numpy() for x in gr])
    chk = lambda a, b, nm: assert_close_frac(a, b, 2e-3, max_bad_frac=5e-3, flip_bound=0.3, what=nm)
    chk(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"], "dmeans3D")
    chk(L["opacities"].grad.cpu().numpy().reshape(-1), rb["dopacities"], "dopac")
    chk(L["shs"].grad.cpu().numpy(), rb["dshs"], "dshs")
    chk(L["others"].grad.cpu().numpy(), rb["dothers"], "dothers")
    chk(o.grad.cpu().numpy(), rb["dray_o"], "dray_o")
    chk(d.grad.cpu().numpy(), rb["dray_d"], "dray_d")


@pytest.mark.parametrize("P,R", [(50, 0), (0, 64), (0, 0), (1, 64)])
def test_trace_empty_inputs(P, R):
    """No rays, no surfels, neither, and a single surfel: shapes follow the inputs, an empty scene renders the background, backward runs."""
    import diff_surfel_tracing as mod
    dev = torch.device("cuda:0")
    g, ro, rd = trace_scene(P=50, R=64, seed=2, camera=False)
    bg = torch.tensor([0.2, 0.4, 0.6])
    L = {k: g[k][:P].to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    v, f = synth.get_disks(L["means3D"].detach(), L["scales"].detach(), L["rotations"].detach())
    tracer = mod.SurfelTracer()
    tracer.build_acceleration_structure(v, f, rebuild=True)
    o = ro[:R].to(dev).requires_grad_(True); d = rd[:R].to(dev).requires_grad_(True)
    outs = tracer(o, d, v, means3D=L["means3D"], grads3D=None, shs=L["shs"], colors_precomp=None, others_precomp=None, opacities=L["opacities"],
                  scales=L["scales"], rotations=L["rotations"], cov3D_precomp=None, tracer_settings=_settings(mod, bg, 1, dev), start_from_first=False)
    rgb, dpt, acc, norm, dist, aux, mid, wet = outs
    assert [tuple(x.shape) for x in outs] == [(R, 3), (R, 1), (R, 1), (R, 3), (R, 1), (R, 2), (R, 16), (P, 1)]
    (rgb.sum() + acc.sum()).backward()
    torch.cuda.synchronize()
    assert o.grad is not None and o.grad.shape == (R, 3) and torch.isfinite(o.grad).all()
    if P == 0 and R:
        assert torch.allclose(rgb.detach(), bg.to(dev).expand(R, 3)) and float(acc.detach().abs().max()) == 0 and float(o.grad.abs().max()) == 0
    if P == 1:
        from oracle import trace as otr
        ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"][:1].numpy(), g["scales"][:1].numpy(), g["rotations"][:1].numpy(),
                                g["opacities"][:1].numpy(), shs=g["shs"][:1].numpy(), sh_degree=1, bg=bg.numpy(), start_from_first=False)
        assert_close_frac(rgb.detach().cpu().numpy(), ref["rgb"], 2e-4, max_bad_frac=0.02, flip_bound=0.1, what="rgb")
