"""GPU: the fused caller-side glue (envgs_amd.fused) against the torch expressions it replaces (the reference's own lines,
re-derived in envgs_amd/envgs_step.py), forward and autograd backward.  Floating-point elementwise kernels: fp32 torch is the checker."""
import pytest
import torch

from envgs_amd import envgs_step, synth
from tests import reference_caller

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deg,S,M", [(0, 1, 16), (2, 1, 16), (3, 3, 16), (2, 1, 9), (1, 3, 4)])
def test_sh_colors_matches_torch(deg, S, M):
    """M = 16: the four-lanes-per-surfel kernels; other coefficient counts: the one-lane-per-surfel ones."""
    from envgs_amd import fused
    dev = torch.device("cuda:0")
    g = synth.base_gaussians(5000, seed=deg, device=dev)
    g["shs"] = g["shs"][:, :M].contiguous()
    cam = synth.orbit_camera(1, device=dev)
    spec = torch.rand(5000, S, device=dev)
    g["shs"][:200, 0] = -3.0                            # strongly negative DC: exercises the clamp and its zero gradient
    leaves = [g["means3D"].clone().requires_grad_(True), g["shs"].clone().requires_grad_(True), spec.clone().requires_grad_(True),
              g["roughness"].clone().requires_grad_(True)]
    out = fused.sh_colors(leaves[0], leaves[1], cam.camera_center, torch.tensor([deg], device=dev), leaves[2], leaves[3])
    ref_l = [t.detach().clone().requires_grad_(True) for t in leaves]
    d = ref_l[0] - cam.camera_center[None]; d = d / d.norm(dim=1, keepdim=True)
    ref = torch.cat([torch.clamp_min(envgs_step.eval_sh(deg, ref_l[1].transpose(1, 2), d) + 0.5, 0.0), ref_l[2], ref_l[3]], dim=-1)
    assert out.shape == ref.shape == (5000, 3 + S + 1)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-6)
    w = torch.randn_like(ref)
    (out * w).sum().backward(); (ref * w).sum().backward()
    for a, b in zip(leaves, ref_l):
        bg = b.grad if b.grad is not None else torch.zeros_like(b)          # degree 0 does not depend on the view direction
        torch.testing.assert_close(a.grad, bg, rtol=1e-4, atol=1e-6)
    assert (out[:, :3] == 0).any()                        # the clamp (and its zero gradient) is exercised


@pytest.mark.parametrize("ratio", [0.0, 0.3])
def test_reflect_matches_torch(ratio):
    from envgs_amd import fused
    dev = torch.device("cuda:0")
    H, W = 40, 56
    cam = synth.orbit_camera(2, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
    ro, rd = synth.get_rays(cam)
    gen = torch.Generator().manual_seed(3)
    allmap = torch.randn(7, H, W, generator=gen).to(dev)
    allmap[1] = torch.rand(H, W, generator=gen).to(dev) * 0.9 + 0.05
    allmap[0] = allmap[1] * (3 + torch.rand(H, W, generator=gen).to(dev))
    allmap[1, :2] = 0; allmap[0, :2] = 0                  # empty pixels: 0/0 -> nan_to_num -> 0, zero gradient
    allmap[2:5, 5, :4] = 0                                # zero normal: x/(|x|+eps) stays finite
    a1 = allmap.clone().requires_grad_(True); o1 = ro.clone().requires_grad_(True); d1 = rd.clone().requires_grad_(True)
    nw, dep, ref_o, ref_d = fused.reflect(a1, o1, d1, cam.world_view_transform, ratio)

    a2 = allmap.clone().requires_grad_(True); o2 = ro.clone().requires_grad_(True); d2 = rd.clone().requires_grad_(True)
    alpha = a2[1:2]
    nw2 = (a2[2:5].permute(1, 2, 0) @ cam.world_view_transform[:3, :3].T).permute(2, 0, 1)          # gaussian2d_utils.py:1123
    de = torch.nan_to_num(a2[0:1] / alpha, 0, 0); dm = torch.nan_to_num(a2[5:6], 0, 0)               # :1126-1131
    dep2 = de * (1 - ratio) + dm * ratio                                                               # :1136
    n = nw2.permute(1, 2, 0); n = n / (n.norm(dim=-1, keepdim=True) + 1e-8)                           # math_utils.normalize
    ref_d2 = d2 - 2 * (d2 * n).sum(-1, keepdim=True) * n                                               # envgs_sampler.py:424
    ref_o2 = o2 + d2 * dep2.permute(1, 2, 0)                                                           # :427
    for x, y in ((nw, nw2), (dep, dep2), (ref_o, ref_o2), (ref_d, ref_d2)):
        torch.testing.assert_close(x, y, rtol=1e-5, atol=1e-5)
    ws = [torch.randn_like(t) for t in (nw2, dep2, ref_o2, ref_d2)]
    sum((x * w).sum() for x, w in zip((nw, dep, ref_o, ref_d), ws)).backward()
    sum((x * w).sum() for x, w in zip((nw2, dep2, ref_o2, ref_d2), ws)).backward()
    # torch's own backward of nan_to_num(0/0) is 0 * inf = NaN at empty pixels (harmless upstream: such pixels have no contributors);
    # the fused kernel returns 0 there, so compare where alpha > 0 and require finiteness everywhere
    live = allmap[1] > 0
    torch.testing.assert_close(a1.grad[2:], a2.grad[2:], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a1.grad[:2][:, live], a2.grad[:2][:, live], rtol=1e-4, atol=1e-4)
    assert float(a1.grad[:2][:, ~live].abs().max()) == 0.0
    torch.testing.assert_close(o1.grad, o2.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d1.grad, d2.grad, rtol=1e-4, atol=1e-4)
    assert torch.isfinite(a1.grad).all()


@pytest.mark.parametrize("ratio", [0.0, 0.4])
def test_surface_normal_matches_torch(ratio):
    """fused.surface_normal (depth select + dpt2norm + alpha scaling, one kernel each way) vs the torch expressions of render()'s tail
    (tests/reference_caller.py: surface_maps; its dpt2norm is pinned against the reference's by tests/test_golden.py)."""
    from envgs_amd import fused
    dev = torch.device("cuda:0")
    H, W = 44, 60
    cam = synth.orbit_camera(3, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
    gen = torch.Generator().manual_seed(5)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    allmap = torch.randn(7, H, W, generator=gen)
    allmap[1] = torch.rand(H, W, generator=gen) * 0.9 + 0.05
    allmap[0] = allmap[1] * (3.0 + 0.01 * xx + 0.3 * torch.sin(yy / 5.0) + 0.05 * torch.rand(H, W, generator=gen))
    allmap[5] = 3.2 + 0.02 * yy + 0.05 * torch.rand(H, W, generator=gen)
    allmap[1, :3, :7] = 0; allmap[0, :3, :7] = 0          # empty pixels: depth 0/0 -> 0
    allmap = allmap.to(dev)
    a1 = allmap.clone().requires_grad_(True)
    sd, sn = fused.surface_normal(a1, cam, ratio)
    a2 = allmap.clone().requires_grad_(True)
    sd2, sn2 = reference_caller.surface_maps(cam, a2, ratio)
    assert sd.shape == (1, H, W) and sn.shape == (3, H, W)
    torch.testing.assert_close(sd, sd2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sn, sn2, rtol=2e-4, atol=2e-5)
    snd = sn.detach()
    assert float(snd[:, 0].abs().max()) == 0 and float(snd[:, :, -1].abs().max()) == 0 and float(snd[:, 5:-5, 10:-5].abs().min(dim=0).values.max()) > 0
    wd, wn = torch.randn_like(sd2), torch.randn_like(sn2)
    ((sd * wd).sum() + (sn * wn).sum()).backward()
    ((sd2 * wd).sum() + (sn2 * wn).sum()).backward()
    live = allmap[1] > 0
    g1, g2 = a1.grad, a2.grad
    assert torch.isfinite(g1).all()
    scale = float(g2[:, live].abs().max())
    for ch in (0, 1, 5):
        err = (g1[ch][live] - g2[ch][live]).abs().max()
        assert float(err) <= 2e-4 * scale + 1e-6, (ch, float(err), scale)
    assert float(g1[[2, 3, 4, 6]].abs().max()) == 0 and float(g1[:2][:, ~live].abs().max()) == 0


def test_surfel_quads_match_get_disks():
    """fused.surfel_quads == the python get_disks twin (itself pinned against the reference's own output by tests/test_golden.py)."""
    from envgs_amd import fused, synth
    dev = torch.device("cuda:0")
    e = synth.env_gaussians(5000, seed=3, device=dev)
    q = e["rotations"] * (0.5 + torch.rand(5000, 1, device=dev))           # un-normalised quaternions, as the optimizer leaves them
    v, f = fused.surfel_quads(e["means3D"], e["scales"], q)
    v2, f2 = synth.get_disks(e["means3D"], e["scales"], q)
    assert v.shape == v2.shape and f.shape == f2.shape and f.dtype == f2.dtype
    assert torch.equal(f, f2)
    assert float((v - v2).abs().max()) <= 2e-6 * float(v2.abs().max())
    v3, f3 = fused.surfel_quads(e["means3D"], e["scales"], q)               # the cached face table
    assert f3 is f and torch.equal(v3, v)


@pytest.mark.parametrize("C", [5, 7])
def test_blend_matches_torch(C):
    """fused.blend == (1 - spec) * rgb_base + spec * rgb_env on slices of the rasterizer's output, values and both gradients."""
    from envgs_amd import fused
    dev = torch.device("cuda:0")
    H, W, S = 37, 53, C - 4
    gen = torch.Generator().manual_seed(4)
    img = torch.rand(C, H, W, generator=gen).to(dev).requires_grad_(True)
    env = torch.rand(H, W, 3, generator=gen).to(dev).requires_grad_(True)
    up = torch.randn(H, W, 3, generator=gen).to(dev)
    out = fused.blend(img, env)
    (out * up).sum().backward()
    gi, ge = img.grad.clone(), env.grad.clone()
    img.grad = None; env.grad = None
    spec = img[3:3 + S].permute(1, 2, 0)
    ref = (1 - spec) * img[:3].permute(1, 2, 0) + spec * env
    (ref * up).sum().backward()
    assert float((out - ref).abs().max()) <= 1e-6
    assert float((gi - img.grad).abs().max()) <= 1e-5 * float(img.grad.abs().max()) and float((ge - env.grad).abs().max()) <= 1e-6
    assert float(gi[C - 1].abs().max()) == 0.0



def test_bounce_stage_glue_matches_torch_f64():
    """fused.bounce_rays / bounce_blend / bounce_pack_mid == the torch expressions of a bounce stage (gaussian2d_sampler.py:413-426 as restated in
    tracing.py:_forward_bounces), values and every gradient against float64 autograd; rows that do not bounce receive exactly zero (rays) or pass
    their gradient through (colour)."""
    from envgs_amd import fused
    dev = torch.device("cuda:0")
    R = 5000
    gen = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=gen)
    o, d = mk(R, 3), mk(R, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    dpt = torch.rand(R, 1, generator=gen) * 3 + 0.5; acc = torch.rand(R, 1, generator=gen) * 0.5 + 0.5
    norm = mk(R, 3) * 0.3; aux = torch.rand(R, 2, generator=gen); rgb = torch.rand(R, 3, generator=gen)
    sel = torch.nonzero(torch.rand(R, generator=gen) < 0.6)[:, 0]
    n = sel.numel()
    col_next = torch.rand(n, 3, generator=gen)
    up_o, up_d, up_c = mk(n, 3), mk(n, 3), mk(R, 3)

    def run(dtype, device, fusedp):
        L = [t.to(device=device, dtype=dtype).requires_grad_(True) for t in (o, d, dpt, acc, norm, aux, rgb, col_next)]
        o_, d_, dpt_, acc_, norm_, aux_, rgb_, cn_ = L
        s_ = sel.to(device)
        if fusedp:
            o2, d2 = fused.bounce_rays(o_, d_, dpt_, acc_, norm_, s_)
            col = fused.bounce_blend(rgb_, aux_, cn_, s_)
        else:
            nh = norm_[s_] / norm_[s_].norm(dim=-1, keepdim=True)
            o2 = o_[s_] + d_[s_] * (dpt_[s_] / acc_[s_])
            d2 = d_[s_] - 2.0 * (d_[s_] * nh).sum(-1, keepdim=True) * nh
            sp = aux_[s_, 0:1]
            col = rgb_.index_put((s_,), (1.0 - sp) * rgb_[s_] + sp * cn_)
        loss = (o2 * up_o.to(device=device, dtype=dtype)).sum() + (d2 * up_d.to(device=device, dtype=dtype)).sum() + (col * up_c.to(device=device, dtype=dtype)).sum()
        loss.backward()
        return [x.detach().double().cpu() for x in (o2, d2, col)], [t.grad.detach().double().cpu() for t in L]

    (ro2, rd2, rcol), rg = run(torch.float64, "cpu", False)
    (go2, gd2, gcol), gg = run(torch.float32, dev, True)
    for a, b in ((ro2, go2), (rd2, gd2), (rcol, gcol)):
        assert float((a - b).abs().max()) <= 2e-6 * (1.0 + float(a.abs().max()))
    names = ("ray_o", "ray_d", "dpt", "acc", "norm", "aux", "rgb", "col_next")
    for nm, a, b in zip(names, rg, gg):
        assert float((a - b).abs().max()) <= 2e-5 * (float(a.abs().max()) + 1e-12), nm
    keep = torch.ones(R, dtype=torch.bool); keep[sel] = False
    for i in (0, 1, 2, 3, 4):                                                # ray-side inputs: rows that do not bounce get exactly nothing
        assert float(gg[i][keep].abs().max()) == 0.0
    assert torch.equal(gg[6][keep].float(), up_c[keep])                      # colour: passed through
    # mid: 16 channels of stage 1 at the rows `sel`, stage 0 at every row
    mid = torch.zeros(R, 32, device=dev)
    st0 = [t.to(dev) for t in (o, d, dpt, acc, norm, aux, rgb)]
    fused.bounce_pack_mid(mid, 0, 2, None, *st0)
    st1 = [t.to(dev) for t in (go2.float(), gd2.float(), dpt[sel], acc[sel], norm[sel], aux[sel], col_next)]
    fused.bounce_pack_mid(mid, 1, 2, sel.to(dev), *st1)
    want = torch.zeros(R, 32)
    want[:, :16] = torch.cat([o, d, dpt, acc, norm, aux, rgb], 1)
    want[sel, 16:] = torch.cat([t.cpu() for t in st1], 1)
    assert torch.equal(mid.cpu(), want)
