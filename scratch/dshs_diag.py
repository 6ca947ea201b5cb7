import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import raster
from oracle import raster as orc
from tests.util import small_scene
from tests.test_raster_parity import _settings, _masked_upstream, _oracle, CASES
import diff_surfel_rasterization_wet as mod
dev = torch.device("cuda:0")
for case in (CASES[0], CASES[4]):
    g, cam = small_scene(P=case["P"], H=case["H"], W=case["W"], seed=case["seed"], C=3, sh=True, scale_mul=case.get("scale_mul", 4.0))
    bg = torch.tensor([0.2, 0.5, 0.9])
    st = _settings(mod, cam, bg, case["deg"], dev)
    ref = _oracle(g, cam, bg, case["deg"], 3, True)
    aud = orc.raster_audit(ref)
    dcol, dall = _masked_upstream(3, case["H"], case["W"], case["seed"] + 100, aud["fragile"])
    gd = {k: v.to(dev) for k, v in g.items()}
    outs, saved = raster.rasterize_forward(3, gd["means3D"], gd["shs"], None, gd["opacities"], gd["scales"], gd["rotations"], None, st)
    grads = raster.rasterize_backward(saved, dcol.to(dev), dall.to(dev))
    rb = orc.raster_backward(ref, dcol.numpy(), dall.numpy(), want_cond=True)
    a = grads["shs"].cpu().numpy().astype(np.float64); b = rb["dshs"].astype(np.float64); c = rb["cond"]["dshs"]; u = rb["unc"]["dshs"]
    e = np.abs(a - b) / (np.abs(b) + 0.02 * c + 4e4 * u + 1e-300)
    idx = np.argsort(-e.reshape(-1))[:10]
    print("case", case)
    rec = grads["grad_rec"].cpu().numpy()
    for i in idx:
        gi, k, ch = np.unravel_index(i, e.shape)
        print("  g %d k %d ch %d  hip %.4e  orc %.4e  cond %.3e unc %.3e err %.2e | rec dcolor hip %s orc %s clamped %s radii %d" % (
            gi, k, ch, a[gi, k, ch], b[gi, k, ch], c[gi, k, ch], u[gi, k, ch], e[gi, k, ch], rec[gi, 15:18], rb["rec_dcolor"][gi], ref["clamped"][gi], ref["radii"][gi]))
