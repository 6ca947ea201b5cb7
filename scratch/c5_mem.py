"""What holds the memory of a configs[4] step right before its backward?  (live CUDA storages, largest first)"""
import sys, os, gc, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import synth, tracing, envgs_step
import diff_surfel_rasterization_wet_ch07 as pkg
import diff_surfel_tracing as tpkg
dev = torch.device("cuda", 0)
P, PE, H, W = 300000, 163840, 1200, 1600
g = synth.base_gaussians(P, seed=0, device=dev); ge = synth.env_gaussians(PE, seed=1, device=dev)
cam = synth.orbit_camera(0, n_views=8, H=H, W=W, fx=1111.1 * 2, device=dev)
names = ["means3D", "shs", "opacities", "scales", "rotations"]
params = {k: g[k].clone().requires_grad_(True) for k in names}
params["specular"] = g["specular"].repeat(1, 3).contiguous().clone().requires_grad_(True); params["roughness"] = g["roughness"].clone().requires_grad_(True)
envp = {k: ge[k].clone().requires_grad_(True) for k in names}
envp["others"] = torch.rand(PE, 2, generator=torch.Generator().manual_seed(3)).to(dev)
envgs_step.FUSED["on"] = True
envgs_step.TRACE.update(depth=2, specular_threshold=0.5)
envgs_step.FEATURE_F16["on"] = True
tracer = tpkg.SurfelTracer()
rays = synth.get_rays(cam)
sh_degree = torch.tensor([3], device=dev)
for it in range(3):
    torch.cuda.reset_peak_memory_stats()
    out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, params, envp, torch.zeros(3, device=dev), torch.zeros(3, device=dev), sh_degree)
    torch.cuda.synchronize()
    if it == 2:
        seen = {}
        for ob in gc.get_objects():
            try:
                if torch.is_tensor(ob) and ob.is_cuda:
                    st = ob.untyped_storage()
                    seen[st.data_ptr()] = max(seen.get(st.data_ptr(), (0, None))[0], st.nbytes()), (tuple(ob.shape), ob.dtype)
            except Exception:
                pass
        tot = sum(v[0] for v in seen.values())
        print("live storages %d, %.2f GB; allocated %.2f GB, peak %.2f GB" % (len(seen), tot / 1e9, torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9))
        for ptr, (nb, meta) in sorted(seen.items(), key=lambda kv: -kv[1][0])[:28]:
            print("  %8.3f GB  %s %s" % (nb / 1e9, meta[0], meta[1]))
    out["rgb"].sum().backward()
    torch.cuda.synchronize()
    print("iter", it, "peak after backward %.2f GB" % (torch.cuda.max_memory_allocated() / 1e9))
    for p_ in list(params.values()) + list(envp.values()):
        p_.grad = None
