"""CPU: the brute-force C tracer oracle (forward + analytic backward) against float64 autograd of the dense
torch-eager restatement (oracle/eager_trace.py)."""
import numpy as np
import pytest
import torch

from envgs_amd import synth
from oracle import eager_trace, trace as otr
from tests.util import rel_err


def trace_scene(P=150, R=400, seed=0, camera=True):
    """Surfels scattered in a shell around the origin, rays from inside (reflection-like) or from a camera."""
    g = torch.Generator().manual_seed(seed)
    dirs = torch.randn(P, 3, generator=g); dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    means = dirs * (3.0 + 4.0 * torch.rand(P, 1, generator=g))
    scales = 0.5 + 1.2 * torch.rand(P, 2, generator=g)
    q = torch.randn(P, 4, generator=g); rots = q / q.norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g))
    shs = torch.cat([torch.rand(P, 1, 3, generator=g) * 3 - 1.5, torch.randn(P, 15, 3, generator=g) * 0.2], dim=1)
    others = torch.rand(P, 2, generator=g)
    if camera:
        cam = synth.orbit_camera(2, H=20, W=20, fx=25.0, radius=1.0)
        ro, rd = synth.get_rays(cam)
        ro, rd = ro.reshape(-1, 3)[:R], rd.reshape(-1, 3)[:R]
    else:
        ro = torch.randn(R, 3, generator=g) * 0.3
        rd = torch.randn(R, 3, generator=g); rd = rd / rd.norm(dim=-1, keepdim=True) * (0.5 + torch.rand(R, 1, generator=g))
    return dict(means3D=means, scales=scales, rotations=rots, opacities=opac, shs=shs, others=others,
                colors_precomp=torch.rand(P, 3, generator=g)), ro.contiguous(), rd.contiguous()


@pytest.mark.parametrize("use_sh,camera,deg", [(True, True, 3), (False, False, 0), (True, False, 2)])
def test_trace_oracle_vs_autograd(use_sh, camera, deg):
    g, ro, rd = trace_scene(seed=3, camera=camera)
    R = ro.shape[0]
    bg = torch.tensor([0.3, 0.1, 0.7])
    gen = torch.Generator().manual_seed(9)
    gr = [torch.randn(R, 3, generator=gen), torch.randn(R, generator=gen), torch.randn(R, generator=gen),
          torch.randn(R, 3, generator=gen), torch.randn(R, 2, generator=gen)]
    ckw = dict(shs=g["shs"].numpy(), sh_degree=deg) if use_sh else dict(colors_precomp=g["colors_precomp"].numpy())
    fwd = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                            g["opacities"].numpy(), others=g["others"].numpy(), bg=bg.numpy(), start_from_first=camera, **ckw)
    bwd = otr.trace_backward(fwd, *[x.numpy() for x in gr])
    assert fwd["nhits"].mean() > 2

    d = torch.float64
    L = {k: g[k].to(d).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "others")}
    if use_sh: L["shs"] = g["shs"].to(d).requires_grad_(True)
    else: L["colors_precomp"] = g["colors_precomp"].to(d).requires_grad_(True)
    o64 = ro.to(d).requires_grad_(True); d64 = rd.to(d).requires_grad_(True)
    rgb, dpt, acc, norm, aux, wet = eager_trace.trace(o64, d64, L["means3D"], L["scales"], L["rotations"], L["opacities"],
                                                      shs=L.get("shs"), colors_precomp=L.get("colors_precomp"), others=L["others"],
                                                      sh_degree=deg, bg=bg, start_from_first=camera)
    for a, b in ((fwd["rgb"], rgb), (fwd["dpt"], dpt), (fwd["acc"], acc), (fwd["norm"], norm), (fwd["aux"], aux), (fwd["wet"], wet)):
        assert rel_err(a, b.detach().numpy()) < 2e-4
    loss = sum((x * y.to(d)).sum() for x, y in zip((rgb, dpt, acc, norm, aux), gr))
    loss.backward()
    tol = 2e-3
    assert rel_err(bwd["dmeans3D"], L["means3D"].grad.numpy()) < tol
    assert rel_err(bwd["dscales"], L["scales"].grad.numpy()) < tol
    assert rel_err(bwd["dopacities"], L["opacities"].grad.reshape(-1).numpy()) < tol
    assert rel_err(bwd["dothers"], L["others"].grad.numpy()) < tol
    q = g["rotations"].double()
    proj = lambda v: v - (v * q).sum(-1, keepdim=True) * q
    assert rel_err(proj(torch.from_numpy(bwd["drots"])).numpy(), proj(L["rotations"].grad).numpy()) < tol
    if use_sh: assert rel_err(bwd["dshs"], L["shs"].grad.numpy()) < tol
    else: assert rel_err(bwd["dcolors"], L["colors_precomp"].grad.numpy()) < tol
    assert rel_err(bwd["dray_o"], o64.grad.numpy()) < tol
    assert rel_err(bwd["dray_d"], d64.grad.numpy()) < tol


def test_trace_oracle_bounces_fill_mid():
    g, ro, rd = trace_scene(seed=4, camera=True)
    fwd = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                            g["opacities"].numpy(), shs=g["shs"].numpy(), sh_degree=1, others=g["others"].numpy(),
                            max_trace_depth=2, specular_threshold=0.1, start_from_first=True)
    mid = fwd["mid"].reshape(-1, 3, 16)
    np.testing.assert_allclose(mid[:, 0, 0:3], ro.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(mid[:, 0, 6], fwd["dpt"], rtol=1e-6)
    np.testing.assert_allclose(mid[:, 0, 11:13], fwd["aux"], rtol=1e-6)
    bounced = np.abs(mid[:, 1, 3:6]).sum(-1) > 0
    assert bounced.any() and (~bounced).any()
    # a bounced ray starts on the stage-0 surface point
    o1 = ro.numpy() + rd.numpy() * (fwd["dpt"] / np.maximum(fwd["acc"], 1e-9))[:, None]
    np.testing.assert_allclose(mid[bounced, 1, 0:3], o1[bounced], rtol=1e-4, atol=1e-5)
