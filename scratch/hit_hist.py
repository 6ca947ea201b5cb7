"""Found / composited hit counts per ray on the bench scene: what the sort network sizes (64 / 128 / 256 keys) are spent on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from envgs_amd import synth, tracing, _lib, envgs_step
import diff_surfel_rasterization_wet_ch05 as pkg
import diff_surfel_tracing as tpkg
dev = torch.device("cuda:0")
H = W = 800
g = synth.base_gaussians(300000, seed=0, device=dev); ge = synth.env_gaussians(163840, seed=1, device=dev)
cam = synth.orbit_camera(0, H=H, W=W, device=dev); rays = synth.get_rays(cam)
base = dict(g); base["specular"] = g["specular"]; base["roughness"] = g["roughness"]
tracer = tpkg.SurfelTracer()
envgs_step.FUSED["on"] = True
tracing.KEEP_LISTS["on"] = True
with torch.no_grad():
    for it in range(3):
        out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, base, ge, torch.zeros(3, device=dev), torch.zeros(3, device=dev), torch.tensor([3], device=dev))
torch.cuda.synchronize()
ids, tb, n_used, hit_cnt = tracing.last_hit_lists()
n = hit_cnt.cpu().numpy().astype(np.int64); u = n_used.cpu().numpy().astype(np.int64)
print("rays", n.size, "found mean %.1f  composited mean %.1f" % (n.mean(), u.mean()))
for name, v in (("found", n), ("composited", u)):
    edges = [0, 1, 33, 65, 97, 129, 161, 193, 257, 10**9]
    h = np.histogram(v, bins=edges)[0]
    print(name, {("%d-%d" % (edges[i], edges[i + 1] - 1)): "%.1f%%" % (100.0 * h[i] / v.size) for i in range(len(h))})
# what a sort sized by ceil(n/64) chunks would cost vs the power-of-two networks: layers x registers
def cost(v, sizes):
    c = 0
    for s, w in sizes:
        pass
    return c
E_pow2 = np.where(n <= 64, 1, np.where(n <= 128, 2, 4)); E_pow2 = np.where(n == 0, 0, E_pow2)
layers = {0: 0, 1: 21, 2: 28, 4: 36}
print("bitonic element-layers per ray (found): mean %.1f;  if sized by composited hits: %.1f" % (
    np.mean([layers[e] * e for e in E_pow2]), np.mean([layers[e] * e for e in np.where(u <= 64, 1, np.where(u <= 128, 2, 4))])))
print(tracing.last_trace_counts())
