"""Key distribution of the BOUNCE stages' rays at the configs[4] size: how many rays share a sort bucket (top 13 key bits)?"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import synth, tracing, envgs_step
from tests.test_tile_binning import _numpy_ray_keys
import diff_surfel_rasterization_wet_ch07 as pkg
import diff_surfel_tracing as tpkg
dev = torch.device("cuda", 0)
P, PE, H, W = 300000, 163840, 1200, 1600
g = synth.base_gaussians(P, seed=0, device=dev); ge = synth.env_gaussians(PE, seed=1, device=dev)
cam = synth.orbit_camera(0, n_views=8, H=H, W=W, fx=1111.1 * 2, device=dev)
names = ["means3D", "shs", "opacities", "scales", "rotations"]
params = {k: g[k].clone() for k in names}
params["specular"] = torch.rand(P, 3, device=dev); params["roughness"] = torch.rand(P, 1, device=dev)
envp = {k: ge[k].clone() for k in names}
envp["others"] = torch.rand(PE, 2, device=dev)
envgs_step.FUSED["on"] = True
envgs_step.TRACE["depth"] = 2; envgs_step.TRACE["specular_threshold"] = 0.3
tracer = tpkg.SurfelTracer()
rays = synth.get_rays(cam)
sh_degree = torch.tensor([3], device=dev)
cap = []
orig = tracing._TraceSurfels.apply
def spy(o, d, *a):
    cap.append((o.detach().cpu().numpy().astype(np.float32), d.detach().cpu().numpy().astype(np.float32)))
    return orig(o, d, *a)
tracing._TraceSurfels.apply = staticmethod(spy)
with torch.no_grad():
    out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, params, envp, torch.zeros(3, device=dev), torch.zeros(3, device=dev), sh_degree)
torch.cuda.synchronize()
for k, (o, d) in enumerate(cap):
    o = o.reshape(-1, 3); d = d.reshape(-1, 3)
    key = _numpy_ray_keys(o, d)
    b = key >> np.uint32(31 - 13)
    cnt = np.bincount(b, minlength=8192)
    fin = np.isfinite(o).all(1) & np.isfinite(d).all(1)
    print("stage", k, "rays", len(o), "non-finite", int((~fin).sum()), "origin box", o[fin].min(0), o[fin].max(0), "pctl |o|", np.percentile(np.abs(o[fin]).max(1), [50, 99, 99.99, 100]))
    print("   buckets > 16384:", int((cnt > 16384).sum()), "largest", np.sort(cnt)[-5:], "rays in long buckets", int(cnt[cnt > 16384].sum()))
    print("   buckets > 2048:", int((cnt > 2048).sum()), "rays in them", int(cnt[cnt > 2048].sum()), "; > 8192:", int((cnt > 8192).sum()), "rays", int(cnt[cnt > 8192].sum()))
    # time the library's ray sort on exactly these rays
    from envgs_amd import _lib
    lib = _lib.load()
    R = len(o)
    od, dd = torch.from_numpy(o).to(dev).contiguous(), torch.from_numpy(d).to(dev).contiguous()
    pairs = torch.zeros(R, dtype=torch.int64, device=dev); order = torch.zeros(R, dtype=torch.int32, device=dev)
    tb = lib.envgs_trace_ray_sort_temp_bytes(R); temp = torch.zeros((tb + 3) // 4, dtype=torch.int32, device=dev)
    st = _lib.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(2):
        lib.envgs_trace_ray_order(R, _lib.ptr(od), _lib.ptr(dd), None, 0, _lib.ptr(pairs), _lib.ptr(order), _lib.ptr(temp), tb, st)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        lib.envgs_trace_ray_order(R, _lib.ptr(od), _lib.ptr(dd), None, 0, _lib.ptr(pairs), _lib.ptr(order), _lib.ptr(temp), tb, st)
    torch.cuda.synchronize()
    print("   ray sort: %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
    if k == 1:
        for dup in (1, 3):
            top = np.argsort(cnt)[-dup * 2:]                       # the densest buckets, each doubled: 2-6 lists beyond the 16384-entry LDS sort
            m = np.isin(b, top)
            o3 = np.concatenate([o, o[m]]); d3 = np.concatenate([d, d[m]])
            R3 = len(o3)
            od, dd = torch.from_numpy(o3).to(dev).contiguous(), torch.from_numpy(d3).to(dev).contiguous()
            pairs = torch.zeros(R3, dtype=torch.int64, device=dev); order = torch.zeros(R3, dtype=torch.int32, device=dev)
            tb = lib.envgs_trace_ray_sort_temp_bytes(R3); temp = torch.zeros((tb + 3) // 4, dtype=torch.int32, device=dev)
            for _ in range(2):
                lib.envgs_trace_ray_order(R3, _lib.ptr(od), _lib.ptr(dd), None, 0, _lib.ptr(pairs), _lib.ptr(order), _lib.ptr(temp), tb, st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                lib.envgs_trace_ray_order(R3, _lib.ptr(od), _lib.ptr(dd), None, 0, _lib.ptr(pairs), _lib.ptr(order), _lib.ptr(temp), tb, st)
            torch.cuda.synchronize()
            print("   with %d doubled buckets (%s rays each): ray sort %.3f ms" % (dup * 2, 2 * cnt[top], (time.perf_counter() - t0) / 5 * 1e3))
    dn = np.linalg.norm(d, axis=1)
    print("   |d| pctl", np.percentile(dn[fin], [0, 1, 50, 99, 100]), "zero dirs", int((dn == 0).sum()))
