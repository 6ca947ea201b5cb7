"""Which rays of the determinism test differ between two runs of the same traced call, by forced list capacity (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from envgs_amd import synth, tracing
from tests.test_trace_parity import _run_hip

P, R = 20000, 8192
e = synth.env_gaussians(P, seed=5)
gen = torch.Generator().manual_seed(12)
ro = (torch.rand(R, 3, generator=gen) * 2 - 1) * 1.3
rd = torch.randn(R, 3, generator=gen); rd = rd / rd.norm(dim=-1, keepdim=True)
g = dict(means3D=e["means3D"] * 0.2, scales=e["scales"] * 0.5, rotations=e["rotations"], opacities=e["opacities"], shs=e["shs"],
         others=torch.rand(P, 2, generator=gen))
tracing.KEEP_LISTS["on"] = True
for force in (None, 128, 256, 320, 512, 1024):
    tracing.HIT_CAP.clear()
    if force: tracing.HIT_CAP["force"] = force
    runs = []
    for _ in range(3):
        outs = _run_hip(g, ro, rd, torch.tensor([0.2, 0.3, 0.4]), 3, True, False)[0]
        ids, tb, n_used, hit_cnt = [x.clone() for x in tracing.last_hit_lists()]
        runs.append(([x.detach().clone() for x in outs], n_used, hit_cnt, tracing.last_trace_counts()))
    a = runs[0]
    for k, b in enumerate(runs[1:]):
        diff = (a[0][0] != b[0][0]).any(-1)
        nd = int(diff.sum())
        msg = "cap %s run0 vs run%d: %d rays differ in rgb; max_list %d/%d cap %d" % (force, k + 1, nd, a[3]["max_list"], b[3]["max_list"], a[3]["cap"])
        if nd:
            i = diff.nonzero()[:8, 0]
            msg += "\n   rays %s\n   hit_cnt a %s\n   hit_cnt b %s\n   n_used a %s\n   n_used b %s\n   |drgb| %s" % (
                i.tolist(), a[2][i].tolist(), b[2][i].tolist(), a[1][i].tolist(), b[1][i].tolist(), (a[0][0][i] - b[0][0][i]).abs().max(-1).values.tolist())
            for j, nm in enumerate(("rgb", "dpt", "acc", "norm", "dist", "aux", "mid", "wet")):
                msg += "\n   %s differs in %d elements" % (nm, int((a[0][j] != b[0][j]).sum()))
        print(msg)
