"""Diagnosis: full-size raster gradients, HIP vs oracle -- dumps |a-b|, |b|, cond per element (dmeans3D, dshs dc-block, dopacities) plus per-pixel
audit flags and contributor-set mismatches, for offline analysis of what the errors scale with."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import raster, synth
from oracle import raster as orc
from tests.util import cam_args
from tests.test_raster_parity import _settings, _masked_upstream
import diff_surfel_rasterization_wet as mod

dev = torch.device("cuda:0")
P, H, W = 300000, 800, 800
g = synth.base_gaussians(P, seed=0); cam = synth.orbit_camera(3, H=H, W=W); bg = torch.ones(3)
st = _settings(mod, cam, bg, 3, dev)
gd = {k: v.to(dev) for k, v in g.items()}
outs, saved = raster.rasterize_forward(3, gd["means3D"], gd["shs"], None, gd["opacities"], gd["scales"], gd["rotations"], None, st, keep_binning=True)
ca = cam_args(cam)
ref = orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), ca["viewmatrix"].numpy(), ca["projmatrix"].numpy(), ca["campos"].numpy(), W, H,
                         scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), shs=g["shs"].numpy(), sh_degree=3, bg=bg.numpy())
aud = orc.raster_audit(ref, want_contrib=True)
print("lmax", aud["lmax"], "fragile", aud["fragile"].mean(), "flips", aud["flips"].mean(), "illcond", aud["illcond"].mean())
contrib, nc_a, col_a = raster.render_audit(saved, aud["lmax"])
c = contrib.cpu().numpy()
mism = (c != aud["contrib"]).any(axis=1).reshape(H, W)
print("pixels with a different contributor set:", int(mism.sum()), "of which NOT flagged fragile:", int((mism & ~aud["fragile"]).sum()), " not flagged as flips:", int((mism & ~aud["flips"]).sum()))
dcol, dall = _masked_upstream(3, H, W, 1, aud["fragile"] | mism)
grads = raster.rasterize_backward(saved, dcol.to(dev), dall.to(dev))
rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
out = {}
for kh, kr in (("means3D", "dmeans3D"), ("opacities", "dopacities"), ("rotations", "drots")):
    a = grads[kh].cpu().numpy().reshape(rb[kr].shape).astype(np.float64); b = rb[kr].astype(np.float64); cd = rb["cond"][kr]
    out[kr + "_err"] = np.abs(a - b).astype(np.float32); out[kr + "_b"] = np.abs(b).astype(np.float32); out[kr + "_cond"] = cd.astype(np.float32)
    un = rb["unc"][kr]
    out[kr + "_unc"] = un.astype(np.float32)
    for kunc in (0.0, 1.0, 4.0, 16.0):
        e = np.abs(a - b) / (np.abs(b) + 0.02 * cd + kunc * 1e4 * un + 1e-300)
        print(kr, "K_UNC", kunc, "max", e.max(), "n>1e-4", int((e > 1e-4).sum()), "of", e.size)
np.savez_compressed("gpurun_out/grad_err_dump.npz", **out)
