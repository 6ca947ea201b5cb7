"""Experiment: distribution of hits per (batch, surfel) entry in the bench scene."""
import sys, os, torch
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
keep = saved["keep"]
ne = keep["n_entries"].long()
ent = keep["entries"]
nb = ne.shape[0]
D = ne[:, 0]
idx = torch.arange(ent.shape[1], device=dev)[None]
m = idx < D[:, None]
cnt = ((ent[m] >> 24) & 63) + 1
print("batches", nb, "entries", int(cnt.numel()), "singles", int(ne[:, 1].sum()), "hits", int(cnt.sum()), "mean", float(cnt.float().mean()))
h = torch.bincount(cnt, minlength=65).cpu().numpy()
cum = 0
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 64)):
    c = int(h[lo:hi + 1].sum()); hh = int((h[lo:hi + 1] * torch.arange(lo, hi + 1).numpy()).sum())
    print("cnt %2d-%2d: %5.1f%% of entries, %5.1f%% of hits" % (lo, hi, 100.0 * c / cnt.numel(), 100.0 * hh / int(cnt.sum())))
print("entries per batch: mean %.0f max %d" % (float(D.float().mean()), int(D.max())))

# dump a sample of batches for offline packing simulations
import numpy as np
sel = torch.arange(0, nb, 50, device=dev)
cap = saved["cap"]
out = dict(D=D[sel].cpu().numpy(), entries=[], pairs=[])
for b in sel.tolist():
    d = int(D[b])
    out["entries"].append(ent[b, :d].cpu().numpy())
    tot = int((((ent[b, :d] >> 24) & 63) + 1).sum())
    out["pairs"].append(keep["pairs"][b, :tot].cpu().numpy())
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/batches_sample.npz", D=out["D"], entries=np.concatenate(out["entries"]), pairs=np.concatenate(out["pairs"]))
print("dumped", len(sel), "batches")
