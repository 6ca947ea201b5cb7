"""Generate tests/golden/loss_golden.npz by IMPORTING the reference's image loss (authoring container only):
0.8 * l1 + 0.2 * (1 - ssim) of configs/models/envgs.yaml:70-72, with
  l1   = easyvolcap/utils/loss_utils.py:319-333   (mean absolute difference)
  ssim = easyvolcap/utils/loss_utils.py:547-549 -> easyvolcap/utils/ssim_utils.py:11-167 (11-tap sigma-1.5 separable Gaussian, padding='same',
         data_range 1, K = (0.01, 0.03), mean over the map)
and its gradient w.r.t. the rendered image (torch autograd, float64).  Fixture = seeded inputs + the reference's outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def main():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ssim_utils", "/root/reference/easyvolcap/utils/ssim_utils.py")
    ssim_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ssim_utils)
    out = {"win": ssim_utils._fspecial_gauss_1d(11, 1.5).numpy().reshape(-1)}
    for tag, (H, W, seed) in {"a": (40, 36, 0), "b": (11, 23, 1), "c": (64, 50, 2)}.items():
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(1, 3, H, W, generator=g, dtype=torch.float64)
        y = (x + 0.2 * torch.randn(1, 3, H, W, generator=g, dtype=torch.float64)).clamp(0, 1)
        if tag == "c":
            y[:, :, :20] = x[:, :, :20]                                   # a region with zero residual
        x.requires_grad_(True)
        l1 = (x - y).abs().mean()                                          # loss_utils.l1 -> l1_reg
        s = ssim_utils.ssim(x, y, data_range=1.0, win_size=11, win_sigma=1.5, K=(0.01, 0.03))
        loss = 0.8 * l1 + 0.2 * (1.0 - s)
        loss.backward()
        out["x_" + tag] = x.detach().numpy()[0]; out["y_" + tag] = y.numpy()[0]
        out["l1_" + tag] = l1.item(); out["ssim_" + tag] = s.item(); out["loss_" + tag] = loss.item()
        out["grad_" + tag] = x.grad.numpy()[0]
    np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), **out)
    print({k: (v if np.isscalar(v) else v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
