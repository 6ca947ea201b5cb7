"""Image loss 0.8 * l1 + 0.2 * (1 - ssim): the numpy oracle against the reference's own outputs
(tests/golden/loss_golden.npz), and the fused HIP loss against the oracle."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_golden.npz")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_loss_oracle_matches_reference_golden(tag):
    from oracle import loss_oracle as lo
    z = np.load(GOLD)
    assert z["win"].dtype == np.float32 and np.array_equal(np.asarray(lo.WINDOW_F32, np.float32), z["win"])      # the reference's float32 window
    loss, l1, ssim, grad = lo.l1_ssim(z["x_" + tag], z["y_" + tag])
    assert abs(l1 - float(z["l1_" + tag])) < 1e-12 and abs(ssim - float(z["ssim_" + tag])) < 1e-12 and abs(loss - float(z["loss_" + tag])) < 1e-12
    g = z["grad_" + tag]
    nz = z["x_" + tag] != z["y_" + tag]                                   # sign(0): autograd's abs gives 0 there, as numpy's sign does
    assert np.allclose(grad, g, rtol=1e-9, atol=1e-14), float(np.abs(grad - g).max())
    assert nz.mean() > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_fused_loss_matches_golden(tag):
    from envgs_amd import loss as eloss
    z = np.load(GOLD)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(z["x_" + tag]).float().to(dev).requires_grad_(True)
    y = torch.from_numpy(z["y_" + tag]).float().to(dev)
    out = eloss.l1_ssim_loss(x, y)
    out.backward()
    assert abs(float(out) - float(z["loss_" + tag])) < 1e-4 * abs(float(z["loss_" + tag]))          # north_star tolerance: 1e-4 rel
    g = z["grad_" + tag]
    err = np.abs(x.grad.cpu().numpy() - g).max() / np.abs(g).max()
    assert err < 1e-4, err


@pytest.mark.gpu
def test_fused_loss_full_size_vs_oracle_and_layouts():
    from oracle import loss_oracle as lo
    from envgs_amd import loss as eloss
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(5)
    H, W = 800, 800
    x = torch.rand(3, H, W, generator=gen)
    y = (x + 0.1 * torch.randn(3, H, W, generator=gen)).clamp(0, 1)
    xd = x.to(dev).requires_grad_(True)
    out = eloss.l1_ssim_loss(xd, y.to(dev), w_l1=0.8, w_ssim=0.2)
    (out * 3.0).backward()                                                # a non-unit upstream gradient
    loss, l1, ssim, grad = lo.l1_ssim(x.numpy(), y.numpy())
    assert abs(float(out) - loss) < 1e-4 * loss
    err = np.abs(xd.grad.cpu().numpy() / 3.0 - grad).max() / np.abs(grad).max()
    assert err < 1e-4, err
    # (H, W, 3) channels-last views, as the sampler hands them over, give the same loss
    xl = x.permute(1, 2, 0).contiguous().to(dev).requires_grad_(True)
    out2 = eloss.l1_ssim_loss(xl.permute(2, 0, 1), y.to(dev))
    out2.backward()
    assert abs(float(out2) - float(out)) < 1e-6 and torch.allclose(xl.grad.permute(2, 0, 1), xd.grad / 3.0, rtol=1e-4, atol=1e-9)
