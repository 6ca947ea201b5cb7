"""CPU restatement (numpy, float64) of the reference's image loss  w_l1 * l1 + w_ssim * (1 - ssim)  and its analytic gradient.
TEST INFRASTRUCTURE ONLY.  Pinned: tests/golden/loss_golden.npz holds the reference's own outputs (tests/golden/make_loss_golden.py imports
easyvolcap/utils/ssim_utils.py and differentiates with autograd); tests/test_loss.py checks this file against them.

Follows easyvolcap/utils/ssim_utils.py:11-27 (window), :30-55 (separable 'same' = zero-padded filter), :58-104 (_ssim), :107-167 (mean over
the map, data_range 1, K = (0.01, 0.03)), easyvolcap/utils/loss_utils.py:319-333 (l1 = mean |x - y|) and the weights of
configs/models/envgs.yaml:70-72."""
import numpy as np


# The reference builds its 11-tap sigma-1.5 window in float32 (ssim_utils.py:19-25: torch.arange(dtype=float) ... g /= g.sum()) and only then
# casts it to the image dtype; these are those float32 values (also stored as "win" in the golden fixture, and the constants of the HIP kernel).
WINDOW_F32 = [float.fromhex(h) for h in ("0x1.0d9570p-10", "0x1.f1fe02p-8", "0x1.26eb18p-5", "0x1.bff0fep-4", "0x1.b43c3ep-3", "0x1.106560p-2",
                                         "0x1.b43c3ep-3", "0x1.bff0fep-4", "0x1.26eb18p-5", "0x1.f1fe02p-8", "0x1.0d9570p-10")]


def window():
    return np.asarray(WINDOW_F32, np.float64)


def blur(img, w):
    """(C,H,W) zero-padded separable correlation (the window is symmetric), rows then columns as gaussian_filter does."""
    r = len(w) // 2
    C, H, W = img.shape
    out = img
    if H >= len(w):
        p = np.pad(out, ((0, 0), (r, r), (0, 0)))
        out = sum(w[k] * p[:, k:k + H, :] for k in range(len(w)))
    if W >= len(w):
        p = np.pad(out, ((0, 0), (0, 0), (r, r)))
        out = sum(w[k] * p[:, :, k:k + W] for k in range(len(w)))
    return out


def l1_ssim(x, y, w_l1=0.8, w_ssim=0.2, want_grad=True):
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    w = window()
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = blur(x, w), blur(y, w)
    s1 = blur(x * x, w) - mu1 * mu1
    s2 = blur(y * y, w) - mu2 * mu2
    s12 = blur(x * y, w) - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    B1, B2 = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    smap = (A1 / B1) * (A2 / B2)
    ssim = smap.mean()
    l1 = np.abs(x - y).mean()
    loss = w_l1 * l1 + w_ssim * (1.0 - ssim)
    if not want_grad:
        return loss, l1, ssim, None
    # d smap / d(E[x^2]), d(E[xy]), d(mu1) at every map pixel (s1 and s12 depend on mu1 too)
    d_ex2 = -(A1 * A2) / (B1 * B2 * B2)
    d_exy = 2.0 * A1 / (B1 * B2)
    d_mu1 = 2.0 * mu2 * A2 / (B1 * B2) - 2.0 * mu1 * A1 * A2 / (B1 * B1 * B2) - 2.0 * mu1 * d_ex2 - mu2 * d_exy
    n = x.size
    g_ssim = blur(d_mu1, w) + 2.0 * x * blur(d_ex2, w) + y * blur(d_exy, w)        # symmetric window: the adjoint of blur is blur
    grad = w_l1 * np.sign(x - y) / n - w_ssim * g_ssim / n
    return loss, l1, ssim, grad
