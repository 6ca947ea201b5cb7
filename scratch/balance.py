"""Experiment: per-batch work distribution of collect_hits and what scheduling could give."""
import sys, os, torch, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import synth, tracing, envgs_step
import diff_surfel_rasterization_wet_ch05 as pkg
import diff_surfel_tracing as tpkg
dev = torch.device("cuda", 0)
P, PE, H, W = 300000, 163840, 800, 800
g = synth.base_gaussians(P, seed=0, device=dev); ge = synth.env_gaussians(PE, seed=1, device=dev)
cam = synth.orbit_camera(0, n_views=8, H=H, W=W, fx=1111.1, device=dev)
names = ["means3D", "shs", "opacities", "scales", "rotations"]
params = {k: g[k].clone() for k in names + ["specular", "roughness"]}
envp = {k: ge[k].clone() for k in names}
envgs_step.FUSED["on"] = True
tracer = tpkg.SurfelTracer()
rays = synth.get_rays(cam)
sh_degree = torch.tensor([3], device=dev)
with torch.no_grad():
    out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, params, envp, torch.zeros(3, device=dev), torch.zeros(3, device=dev), sh_degree)
ro, rd = out["ref_o"].reshape(-1, 3).contiguous(), out["ref_d"].reshape(-1, 3).contiguous()
ts = tpkg.SurfelTracingSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
    viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False,
    max_trace_depth=0, specular_threshold=0.0)
for _ in range(2):
    outs, saved = tracing.trace_forward(tracer.nodes, ro, rd, envp["means3D"], envp["shs"], None, None, envp["opacities"], envp["scales"], envp["rotations"], ts, False)
torch.cuda.synchronize()
keep = saved["keep"]; R = ro.shape[0]
order = keep["ray_order"][R:2 * R].long()
found = keep["hit_cnt"].float()[order]
nb = R // 64
cost = found[:nb * 64].reshape(nb, 64).amax(1) * 0 + found[:nb * 64].reshape(nb, 64).sum(1)      # proxy: hits found per batch
c = cost.cpu().numpy()
import numpy as np
print("batches", nb, "cost mean %.0f  std %.0f  min %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (c.mean(), c.std(), c.min(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max()))
def greedy(costs, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for x in costs:
        t = heapq.heappop(h); heapq.heappush(h, t + x)
    return max(h)
for slots in (6144, 8192):
    ideal = max(c.sum() / slots, c.max())
    print("slots", slots, "ideal %.0f  in-order greedy %.0f  LPT %.0f" % (ideal, greedy(c, slots), greedy(np.sort(c)[::-1], slots)))
# chord-length proxy
nodes = tracer.nodes
n0 = nodes[0]
lo = torch.minimum(n0[0:3], n0[6:9]); hi = torch.maximum(n0[3:6], n0[9:12])
inv = 1.0 / rd
t0 = (lo - ro) * inv; t1 = (hi - ro) * inv
tn = torch.minimum(t0, t1).amax(1).clamp(min=0); tf = torch.maximum(t0, t1).amin(1)
chord = ((tf - tn).clamp(min=0) * rd.norm(dim=1))[order]
cb = chord[:nb * 64].reshape(nb, 64).sum(1).cpu().numpy()
print("corr(chord, cost) = %.3f" % np.corrcoef(cb, c)[0, 1])
idx = np.argsort(-cb)
print("LPT by chord proxy: slots 6144 -> %.0f" % greedy(c[idx], 6144))
