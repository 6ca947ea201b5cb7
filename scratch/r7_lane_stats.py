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
# round 5: how many passes would PAIRING save?  Two passes of a wavefront can share one if their active lanes lie in different
# halves of the wavefront (rows 0-3 / rows 4-7 of the 8x8 quadrant: the transpose-reduce then stops one stage early and yields both sums) and no
# pass between them touches a pixel of the one that moves.  Greedy, in list order, within a staged batch of 192 entries.
pair_saved = 0; half_only = 0; pair_adjacent = 0
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
        up = qb[:32].sum(0); lo = qb[32:].sum(0)
        idx = np.nonzero(n_act > 0)[0]                     # the passes this wavefront runs, in list order (the kernel walks them back to front: same pairs)
        kind = np.where(lo[idx] == 0, 1, np.where(up[idx] == 0, 2, 0))          # 1 = upper half only, 2 = lower half only, 0 = both
        half_only += int((kind > 0).sum())
        # adjacent pairs of opposite halves inside the same staged batch (no pass in between, so no ordering question at all)
        j = 0
        while j + 1 < len(idx):
            if kind[j] and kind[j + 1] and kind[j] != kind[j + 1] and idx[j] // 192 == idx[j + 1] // 192:
                pair_adjacent += 1; j += 2
            else:
                j += 1
        # greedy with look-ahead: an open half-pass waits for a partner of the other half; a pass that touches the open one's half closes it
        open_kind = 0
        for kk, ii in zip(kind, idx):
            if open_kind and kk and kk != open_kind:
                pair_saved += 1; open_kind = 0
            elif kk:
                open_kind = kk                              # (a same-half pass replaces the open one: it touches its pixels)
            else:
                open_kind = 0
    inst_any += int(any_inst.sum())
print("tile instances N = %d; instances blended by >= 1 pixel: %d (%.1f %%)" % (N, inst_any, 100.0 * inst_any / N))
print("(wave, entry) pairs within max_last: %d; with a candidate pixel: %d; with >= 1 active lane: %d" % (tot_ws, cand_ws, act_ws))
print("active lanes per active (wave, entry): %.1f of 64 (%.1f %%)" % (lanes / act_ws, 100.0 * lanes / act_ws / 64))
h = hist[1:]; cs = np.cumsum(h) / h.sum()
print("active-lane histogram quantiles: <=8 lanes %.2f, <=16 %.2f, <=32 %.2f, <=48 %.2f" % (cs[7], cs[15], cs[31], cs[47]))
print("passes active in ONE half of the wavefront only: %d (%.1f %% of the active passes)" % (half_only, 100.0 * half_only / act_ws))
print("pairing potential: adjacent opposite-half pairs %d (%.1f %% of the passes saved); greedy with look-ahead %d (%.1f %%)" % (
    pair_adjacent, 100.0 * pair_adjacent / act_ws, pair_saved, 100.0 * pair_saved / act_ws))
