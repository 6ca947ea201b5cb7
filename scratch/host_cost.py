"""Host-side cost per call of the package entry points (tiny inputs: the GPU work is negligible, the wall time is launch + Python overhead)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from envgs_amd import synth, tracing, raster, envgs_step
import diff_surfel_rasterization_wet_ch05 as pkg
import diff_surfel_tracing as tpkg
dev = torch.device("cuda:0")
H = W = 64
g = synth.base_gaussians(1000, seed=0, device=dev); ge = synth.env_gaussians(1000, seed=1, device=dev)
cam = synth.orbit_camera(0, H=H, W=W, device=dev); rays = synth.get_rays(cam)
base = dict(g); base["specular"] = g["specular"]; base["roughness"] = g["roughness"]
tracer = tpkg.SurfelTracer()
envgs_step.FUSED["on"] = True
for k in list(base): base[k] = base[k].clone().requires_grad_(True)
for k in list(ge): ge[k] = ge[k].clone().requires_grad_(True)
bg = torch.zeros(3, device=dev); deg = torch.tensor([3], device=dev)

def timeit(name, fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print("%-44s %.1f us per call" % (name, (time.perf_counter() - t) / n * 1e6))

def fwd():
    return envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, base, ge, bg, bg, deg)
def fwd_bwd():
    out = fwd(); (out["rgb"].sum()).backward()
with torch.no_grad():
    timeit("envgs_forward (no grad)", fwd)
timeit("envgs_forward (grad)", fwd)
timeit("envgs_forward + backward", fwd_bwd)
# pieces
st = envgs_step  # noqa
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(200): fwd_bwd()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:7000])
