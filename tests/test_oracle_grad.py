"""CPU: the C oracle's forward and hand-written backward (R6-R8) against float64 autograd of the
torch-eager restatement (oracle/eager.py).  This is what makes the C oracle trustworthy as the
parity target for the HIP kernels: every gradient is checked against a true derivative."""
import numpy as np
import pytest
import torch

from oracle import eager, raster as orc
from tests.util import small_scene, cam_args, rel_err


def _run(sh, C, precomp_T, seed):
    g, cam = small_scene(P=300, H=48, W=64, seed=seed, C=C, sh=sh)
    ca = cam_args(cam)
    W, H = ca["W"], ca["H"]
    bg = torch.tensor([0.3, 0.6, 0.1])
    gen = torch.Generator().manual_seed(seed + 1)
    dcol = torch.randn(C, H, W, generator=gen) / (H * W)
    dall = torch.randn(7, H, W, generator=gen) / (H * W)

    kw_np = dict(scales=g["scales"].numpy(), rotations=g["rotations"].numpy())
    tm = None
    if precomp_T:
        from envgs_amd import synth
        tm = synth.transmat_python(cam, g["means3D"], g["scales"], g["rotations"])
        kw_np = dict(transmat_precomp=tm.numpy())
    col_kw = dict(shs=g["shs"].numpy(), sh_degree=3) if sh else dict(colors_precomp=g["colors_precomp"].numpy())
    fwd = orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), ca["viewmatrix"].numpy(), ca["projmatrix"].numpy(),
                             ca["campos"].numpy(), W, H, bg=bg.numpy(), **kw_np, **col_kw)
    bwd = orc.raster_backward(fwd, dcol.numpy(), dall.numpy())

    d = torch.float64
    leaves = {k: g[k].to(d).requires_grad_(True) for k in ("means3D", "opacities")}
    if precomp_T: leaves["transmat_precomp"] = tm.to(d).requires_grad_(True)
    else: leaves.update({k: g[k].to(d).requires_grad_(True) for k in ("scales", "rotations")})
    if sh: leaves["shs"] = g["shs"].to(d).requires_grad_(True)
    else: leaves["colors_precomp"] = g["colors_precomp"].to(d).requires_grad_(True)
    out_color, radii, allmap, weight = eager.rasterize(
        leaves["means3D"], leaves["opacities"], ca["viewmatrix"].to(d), ca["projmatrix"].to(d), ca["campos"].to(d), W, H,
        scales=leaves.get("scales"), rotations=leaves.get("rotations"), transmat_precomp=leaves.get("transmat_precomp"),
        shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"), sh_degree=3, bg=bg)
    loss = (out_color * dcol.to(d)).sum() + (allmap * dall.to(d)).sum()
    loss.backward()
    return g, fwd, bwd, (out_color, radii, allmap, weight), leaves


@pytest.mark.parametrize("sh,C,precomp_T", [(True, 3, False), (False, 5, False), (False, 7, True)])
def test_oracle_forward_and_backward_vs_autograd(sh, C, precomp_T):
    g, fwd, bwd, (out_color, radii, allmap, weight), leaves = _run(sh, C, precomp_T, seed=3)
    assert (fwd["radii"] > 0).sum() > 100 and fwd["N"] > 500
    np.testing.assert_array_equal(fwd["radii"], radii.numpy())
    assert rel_err(fwd["out_color"], out_color.detach().numpy()) < 2e-4
    for ch in (0, 1, 2, 3, 4):
        assert rel_err(fwd["allmap"][ch], allmap[ch].detach().numpy()) < 2e-4, ch
    # distortion = sum w (m^2 A + M2 - 2 m M1) cancels catastrophically in fp32 (the reference computes it in fp32 too)
    assert rel_err(fwd["allmap"][6], allmap[6].detach().numpy()) < 5e-3
    # median depth is a selection: allow a handful of pixels to pick a neighbouring splat
    med_bad = np.abs(fwd["allmap"][5] - allmap[5].detach().numpy()) > 1e-3
    assert med_bad.mean() < 2e-3
    assert rel_err(fwd["weight"], weight.detach().numpy()) < 2e-4

    tol = 2e-3
    assert rel_err(bwd["dopacities"], leaves["opacities"].grad.reshape(-1).numpy()) < tol
    if sh:
        assert rel_err(bwd["dshs"], leaves["shs"].grad.numpy()) < tol
    else:
        assert rel_err(bwd["dcolors"], leaves["colors_precomp"].grad.numpy()) < tol
    if precomp_T:
        assert rel_err(bwd["dtransmat_precomp"], leaves["transmat_precomp"].grad.numpy()) < tol
        assert leaves["means3D"].grad is None or float(leaves["means3D"].grad.abs().max()) == 0.0
    else:
        assert rel_err(bwd["dmeans3D"], leaves["means3D"].grad.numpy()) < tol
        assert rel_err(bwd["dscales"], leaves["scales"].grad.numpy()) < tol
        # the kernel returns dL/d(q/|q|); torch's own normalize backward projects it -- compare projected
        q = g["rotations"].double()
        proj = lambda v: v - (v * q).sum(-1, keepdim=True) * q
        assert rel_err(proj(torch.from_numpy(bwd["drots"]).double()).numpy(), proj(leaves["rotations"].grad).numpy()) < tol


def test_means2d_grad_is_the_densification_proxy():
    g, fwd, bwd, _, _ = _run(True, 3, False, seed=5)
    W, H = fwd["W"], fwd["H"]
    vis = fwd["radii"] > 0
    exp_x = (bwd["rec_dT"][:, 2] * fwd["transmat"][:, 8] * 0.5 * W)[vis]
    exp_y = (bwd["rec_dT"][:, 5] * fwd["transmat"][:, 8] * 0.5 * H)[vis]
    np.testing.assert_allclose(bwd["dmeans2D"][vis, 0], exp_x, rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(bwd["dmeans2D"][vis, 1], exp_y, rtol=1e-5, atol=1e-12)
    assert np.all(bwd["dmeans2D"][:, 2] == 0) and np.all(bwd["dmeans2D"][~vis] == 0)
    assert np.abs(bwd["dmeans2D"][vis]).max() > 0
