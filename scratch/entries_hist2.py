"""Round 6: distribution of hits per (batch, surfel) entry (compact layout) -- configs[2] env rays, and camera rays / bounce rays over the base set.
How much of batch_surfel_bwd's per-ENTRY cost goes to entries with one or two hits?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import synth, tracing, envgs_step, fused
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


def hist(label, S, o, d, sff):
    v, f = fused.surfel_quads(S["means3D"], S["scales"], S["rotations"])
    nodes, _ = tracing.build_bvh(v, S["opacities"])
    caps = tracing.CapState()
    for _ in range(3):
        outs, saved = tracing.trace_forward(nodes, o, d, S["means3D"], S["shs"], None, None, S["opacities"], S["scales"], S["rotations"], ts, sff, caps=caps)
        torch.cuda.synchronize()
    keep = saved["keep"]
    ne = keep["n_entries"].long()                       # (nbatch, 2): table entries, single (unmerged) entries
    br = keep["batch_rows"].view(-1, 2).long()          # (nbatch, 2): first row, rows
    D = ne[:, 0]
    start = br[:, 0]
    idx = torch.repeat_interleave(start, D) + (torch.arange(int(D.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(D, 0) - D, D))
    ent = keep["entries"][idx]
    cnt = ((ent >> 24) & 63) + 1
    nsing = int(ne[:, 1].sum())
    print("%s: batches %d table entries %d (+ %d unmerged singles) hits %d mean %.1f hits/entry" % (label, ne.shape[0], int(cnt.numel()), nsing, int(cnt.sum()) + nsing, float(cnt.float().mean())))
    h = torch.bincount(cnt, minlength=65).cpu().numpy()
    tot_e = cnt.numel() + nsing; tot_h = int(cnt.sum()) + nsing
    for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 64)):
        c = int(h[lo:hi + 1].sum()) + (nsing if lo == 1 else 0); hh = int((h[lo:hi + 1] * torch.arange(lo, hi + 1).numpy()).sum()) + (nsing if lo == 1 else 0)
        print("   hits %2d-%2d: %5.1f%% of entries, %5.1f%% of hits" % (lo, hi, 100.0 * c / tot_e, 100.0 * hh / tot_h))
    return outs


hist("configs[2] env rays", envp, ro, rd, False)
o0, d0 = rays[0].reshape(-1, 3).contiguous(), rays[1].reshape(-1, 3).contiguous()
base = {k: g[k] for k in names}
outs = hist("camera rays over the base set", base, o0, d0, True)
# one bounce off the base set (random surfel orientations: incoherent reflected rays)
rgb, dpt, acc, norm = outs[0], outs[1], outs[2], outs[3]
sel = ((acc[:, 0] > 0.5) & (norm.norm(dim=-1) > 0)).nonzero()[:, 0]
o1, d1 = fused.bounce_rays(o0, d0, dpt, acc, norm, sel)
hist("bounce rays over the base set", base, o1.contiguous(), d1.contiguous(), 2)
