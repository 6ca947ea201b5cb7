"""How long does a 2-bounce trace (K-buffer path) take on the bench scene?"""
import sys, os, time, torch
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
others = torch.rand(PE, 2, device=dev)
for depth in (0, 1, 2):
    ts = tpkg.SurfelTracingSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False,
        max_trace_depth=depth, specular_threshold=0.3)
    for _ in range(2):
        outs, saved = tracing.trace_forward(tracer.nodes, ro, rd, envp["means3D"], envp["shs"], None, others, envp["opacities"], envp["scales"], envp["rotations"], ts, False, need_grad=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        outs, saved = tracing.trace_forward(tracer.nodes, ro, rd, envp["means3D"], envp["shs"], None, others, envp["opacities"], envp["scales"], envp["rotations"], ts, False, need_grad=False)
    torch.cuda.synchronize()
    print("max_trace_depth", depth, "forward ms", (time.perf_counter() - t0) / 3 * 1e3)
