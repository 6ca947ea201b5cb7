"""Experiment: how coherent are 64-ray batches of the bench's reflected rays under different orderings?"""
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
print("dir norm", rd.norm(dim=1).mean().item(), "nan", torch.isnan(rd).any().item())
ts = tpkg.SurfelTracingSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
    viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False,
    max_trace_depth=0, specular_threshold=0.0)
for _ in range(2):
    outs, saved = tracing.trace_forward(tracer.nodes, ro, rd, envp["means3D"], envp["shs"], None, None, envp["opacities"], envp["scales"], envp["rotations"], ts, False)
torch.cuda.synchronize()
keep = saved["keep"]; R = ro.shape[0]; cap = saved["cap"]
sid = keep["hit_lists"][:, :, 1].long()
nu = keep["n_used"].long(); nf = keep["hit_cnt"].long().clamp(max=cap)
order = keep["ray_order"][R:2 * R].long()
print("R", R, "cap", cap, "used/ray", nu.float().mean().item(), "found/ray", nf.float().mean().item())
kk = torch.arange(cap, device=dev)[None]

def ratio(rank, n, label):
    # rank[r] = position of ray r in the processing order
    batch = (rank // 64)[:, None].expand(-1, cap)
    m = kk < n[:, None]
    keys = (batch[m] * PE + sid[m])
    u = torch.unique(keys).numel()
    tot = int(m.sum())
    print("%-34s hits %.1fM  unique(batch,surfel) %.2fM  dedupe x%.2f   union/ray-avg = %.2f" % (label, tot / 1e6, u / 1e6, tot / u, (u / (R / 64)) / (tot / R)))

rank_cur = torch.empty(R, dtype=torch.long, device=dev); rank_cur[order] = torch.arange(R, device=dev)
for n, nm in ((nu, "composited"), (nf, "found")):
    ratio(rank_cur, n, nm + " current sort")
    ratio(torch.arange(R, device=dev), n, nm + " pixel rows")
    py, px = torch.arange(R, device=dev) // W, torch.arange(R, device=dev) % W
    ratio(((py // 8) * (W // 8) + (px // 8)) * 64 + (py % 8) * 8 + px % 8, n, nm + " 8x8 pixel tiles")
    # coarse direction (octahedral 2^b cells) then perpendicular-plane position
    d = rd / rd.norm(dim=1, keepdim=True)
    inv = 1.0 / d.abs().sum(1)
    u_, v_ = d[:, 0] * inv, d[:, 1] * inv
    neg = d[:, 2] < 0
    uu = torch.where(neg, (1 - v_.abs()) * torch.sign(u_ + 1e-30), u_); vv = torch.where(neg, (1 - u_.abs()) * torch.sign(v_ + 1e-30), v_)
    for db, pb in ((3, 6), (4, 5), (5, 4), (2, 7)):
        qu = ((uu * 0.5 + 0.5) * (1 << db)).clamp(0, (1 << db) - 1).long(); qv = ((vv * 0.5 + 0.5) * (1 << db)).clamp(0, (1 << db) - 1).long()
        # perpendicular plane basis from the cell-centre direction: use two fixed axes least aligned with d
        ax = d.abs().argmin(1)
        e = torch.zeros_like(d); e[torch.arange(R), ax] = 1
        t1 = torch.cross(d, e, dim=1); t1 = t1 / t1.norm(dim=1, keepdim=True); t2 = torch.cross(d, t1, dim=1)
        a, b = (ro * t1).sum(1), (ro * t2).sum(1)
        qa = ((a - a.min()) / (a.max() - a.min()) * (1 << pb)).clamp(0, (1 << pb) - 1).long(); qb = ((b - b.min()) / (b.max() - b.min()) * (1 << pb)).clamp(0, (1 << pb) - 1).long()
        def mort(x, y, bits):
            k = torch.zeros_like(x)
            for i in range(bits):
                k |= ((x >> i) & 1) << (2 * i) | ((y >> i) & 1) << (2 * i + 1)
            return k
        key = (mort(qu, qv, db) * 8 + ax) << (2 * pb) | mort(qa, qb, pb)
        rk = torch.empty(R, dtype=torch.long, device=dev); rk[torch.argsort(key, stable=True)] = torch.arange(R, device=dev)
        ratio(rk, n, nm + " dir 2^%d + perp 2^%d" % (db, pb))
