"""How full are R7's wavefronts?  From the oracle's per-pixel contributor flags of the 300 k / 800x800 bench view: for every (tile, list entry,
8x8 quadrant) the number of pixels that blended the entry (= lanes with `act` in composite_bwd) and whether any pixel is still a candidate
(ci < last).  CPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from envgs_amd import synth
from tests.util import cam_args
from oracle import raster as orc
P, H, W = 300000, 800, 800
g = synth.base_gaussians(P, seed=0); cam = synth.orbit_camera(0, H=H, W=W); bg = np.zeros(3, np.float32)
ca = cam_args(cam)
ref = orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), ca["viewmatrix"].numpy(), ca["projmatrix"].numpy(), ca["campos"].numpy(), W, H,
                         scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), shs=g["shs"].numpy(), sh_degree=3, bg=bg)
aud = orc.raster_audit(ref, want_contrib=True)
lmax = aud["lmax"]; c = aud["contrib"].reshape(H, W, lmax)
r = ref["ranges"].astype(np.int64); ln = r[:, 1] - r[:, 0]
gx = W // 16
N = int(ln.sum())
tot_ws = 0; cand_ws = 0; act_ws = 0; lanes = 0; hist = np.zeros(65, np.int64); inst_any = 0
last = ref["n_contrib"][0]
for t in range(gx * (H // 16)):
    tx, ty = t % gx, t // gx
    L = int(ln[t])
    if L == 0: continue
    blk = c[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16, :L]                       # (16,16,L)
    lst = last[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    ml = int(lst.max())
    any_inst = np.zeros(L, bool)
    for q in range(4):
        qb = blk[(q >> 1) * 8:(q >> 1) * 8 + 8, (q & 1) * 8:(q & 1) * 8 + 8, :].reshape(64, L)
        ql = lst[(q >> 1) * 8:(q >> 1) * 8 + 8, (q & 1) * 8:(q & 1) * 8 + 8].reshape(64)
        n_act = qb.sum(0)                                                            # (L,)
        cand = (ql[:, None] > np.arange(L)[None]).any(0)
        tot_ws += min(L, ml); cand_ws += int(cand.sum()); act_ws += int((n_act > 0).sum()); lanes += int(n_act.sum())
        hist += np.bincount(n_act, minlength=65)[:65]
        any_inst |= n_act > 0
    inst_any += int(any_inst.sum())
print("tile instances N = %d; instances blended by >= 1 pixel: %d (%.1f %%)" % (N, inst_any, 100.0 * inst_any / N))
print("(wave, entry) pairs within max_last: %d; with a candidate pixel: %d; with >= 1 active lane: %d" % (tot_ws, cand_ws, act_ws))
print("active lanes per active (wave, entry): %.1f of 64 (%.1f %%)" % (lanes / act_ws, 100.0 * lanes / act_ws / 64))
h = hist[1:]; cs = np.cumsum(h) / h.sum()
print("active-lane histogram quantiles: <=8 lanes %.2f, <=16 %.2f, <=32 %.2f, <=48 %.2f" % (cs[7], cs[15], cs[31], cs[47]))
