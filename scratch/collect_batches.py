"""Per-batch cost of the packet traversal on the bench scene (debug switch 2048): steps, leaf tests and shader cycles of every 64-ray batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from envgs_amd import synth, tracing, _lib, envgs_step
import diff_surfel_rasterization_wet_ch05 as pkg
import diff_surfel_tracing as tpkg
dev = torch.device("cuda:0")
lib = _lib.load()
H = W = 800
g = synth.base_gaussians(300000, seed=0, device=dev); ge = synth.env_gaussians(163840, seed=1, device=dev)
cam = synth.orbit_camera(0, H=H, W=W, device=dev); rays = synth.get_rays(cam)
base = dict(g); base["specular"] = g["specular"]; base["roughness"] = g["roughness"]
tracer = tpkg.SurfelTracer()
envgs_step.FUSED["on"] = True
tracing.KEEP_LISTS["on"] = True
keep = {}
orig = tracing.trace_forward
def spy(*a, **k):
    outs, saved = orig(*a, **k); keep["saved"] = saved; return outs, saved
tracing.trace_forward = spy
with torch.no_grad():
    for it in range(3):
        lib.envgs_debug_set(0, 2048 if it == 2 else 0)
        out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, base, ge, torch.zeros(3, device=dev), torch.zeros(3, device=dev), torch.tensor([3], device=dev))
torch.cuda.synchronize()
lib.envgs_debug_set(0, 0)
sp = keep["saved"]["keep"]["spill"].cpu().numpy()
nb = (H * W + 63) // 64
d = sp[:4 * nb].reshape(nb, 4)
steps, leaves, cyc = d[:, 0], d[:, 1], d[:, 2].astype(np.int64) * 16
print("batches", nb, "steps: mean %.0f median %.0f p90 %.0f p99 %.0f max %d" % (steps.mean(), np.median(steps), np.quantile(steps, .9), np.quantile(steps, .99), steps.max()))
print("leaf tests: mean %.0f max %d" % (leaves.mean(), leaves.max()))
print("cycles per batch: mean %.3g median %.3g p90 %.3g p99 %.3g max %.3g  (x 1/2.4e9 s: mean %.3f ms, max %.3f ms)" % (cyc.mean(), np.median(cyc), np.quantile(cyc, .9), np.quantile(cyc, .99), cyc.max(), cyc.mean() / 2.4e6, cyc.max() / 2.4e6))
print("cycles per step: mean %.0f; correlation(steps+leaves, cycles) = %.3f" % ((cyc / np.maximum(steps, 1)).mean(), np.corrcoef(steps + leaves, cyc)[0, 1]))
cnt = tracing.last_trace_counts(); print(cnt)
