"""Round 6: the elements of test_backward_sparse_distortion_gradient that exceed the contract at K_UNC = 1 -- value, oracle, cond, unc; two runs (determinism)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import small_scene
from tests import test_raster_parity as T
from oracle import raster as orc
import diff_surfel_rasterization_wet_ch05 as mod
dev = torch.device("cuda:0")
C, H, W = 5, 64, 80
g, cam = small_scene(P=500, H=H, W=W, seed=12, C=C, sh=False)
bg = torch.tensor([0.2, 0.5, 0.9])
st = T._settings(mod, cam, bg, 0, dev)
ref = T._oracle(g, cam, bg, 0, C, False)
aud = orc.raster_audit(ref)
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
sparse = ((xx % 8 == 3) & (yy % 8 == 5)) & torch.from_numpy(~aud["fragile"])
dcol = torch.zeros(C, H, W); dall = torch.zeros(7, H, W)
dall[6] = torch.randn(H, W, generator=torch.Generator().manual_seed(3)) / (H * W) * sparse
rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
vals = []
for run in range(3):
    leaves = {k: g[k].to(dev).requires_grad_(True) for k in ("means3D", "opacities", "scales", "rotations", "colors_precomp")}
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True) + 0
    means2D.retain_grad()
    color, radii, allmap, weight = mod.GaussianRasterizer(raster_settings=st)(
        means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors_precomp"],
        opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    ((color * dcol.to(dev)).sum() + (allmap * dall.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    a = means2D.grad.cpu().numpy().astype(np.float64); b = rb["dmeans2D"].astype(np.float64)
    cond = rb["cond"]["dmeans2D"].reshape(b.shape); unc = rb["unc"]["dmeans2D"].reshape(b.shape)
    fl = 0.01 * np.abs(b).mean() + 0.02 * cond + 1e4 * unc
    err = np.abs(a - b) / (np.abs(b) + fl)
    idx = np.argsort(err.reshape(-1))[::-1][:4]
    print("run", run, "max err %.3g" % err.max())
    for i in idx:
        p_, c_ = divmod(int(i), b.shape[1])
        print("   surfel %d comp %d: hip %.6e oracle %.6e diff %.3e  cond %.3e unc %.3e  err %.3g  radius %d" % (p_, c_, a[p_, c_], b[p_, c_], a[p_, c_] - b[p_, c_], cond[p_, c_], unc[p_, c_], err[p_, c_], ref["radii"][p_]))
    vals.append(a.copy())
print("run-to-run max |diff|:", np.abs(vals[0] - vals[1]).max(), np.abs(vals[1] - vals[2]).max())
