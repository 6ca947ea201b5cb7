"""Forward-only timing of the tracer list kernels (register_hits knock-outs); derived from bsb_ab.py.
Round 6: per-hit cost of the tracer's list backward, colour-only form (batch_surfel_bwd<true>) against the generic form (<false>), on the SAME
forward: the configs[2] view's reflected rays over the 163 840-surfel env set, traced with `others` and the full per-hit state; then the backward is
run with (a) the colour's upstream gradient only, (b) all five.  Also camera rays over the 300 k base set (the reference's use_base_tracing call).
    python scratch/bsb_ab.py [tag]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from envgs_amd import synth, tracing, _lib, fused
import diff_surfel_tracing as tpkg
import diff_surfel_rasterization_wet_ch05 as pkg
from envgs_amd import envgs_step

tag = sys.argv[1] if len(sys.argv) > 1 else "run"
dev = torch.device("cuda:0")
lib = _lib.load()
H = W = 800
g = synth.base_gaussians(300000, seed=0, device=dev)
ge = synth.env_gaussians(163840, seed=1, device=dev)
cam = synth.orbit_camera(0, n_views=8, H=H, W=W, fx=1111.1, device=dev)
rays = synth.get_rays(cam)
envgs_step.FUSED["on"] = True
base = {k: g[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")}
base["specular"] = g["specular"]; base["roughness"] = g["roughness"]
tracer = tpkg.SurfelTracer()
with torch.no_grad():
    out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, base, {k: ge[k] for k in base if k in ge}, torch.zeros(3, device=dev), torch.zeros(3, device=dev), torch.tensor([3], device=dev))
ref_o, ref_d = out["ref_o"].reshape(-1, 3).contiguous(), out["ref_d"].reshape(-1, 3).contiguous()


def names():
    n = 0
    out = []
    while lib.envgs_prof_kernel_name(n):
        out.append(lib.envgs_prof_kernel_name(n).decode()); n += 1
    return out
NAMES = names()


def drain():
    r = {}
    for k, nm in enumerate(NAMES):
        t_, c_ = ctypes.c_double(0), ctypes.c_int(0)
        lib.envgs_prof_read(k, ctypes.byref(t_), ctypes.byref(c_))
        if c_.value:
            r[nm] = (t_.value / c_.value, c_.value)
    return r



def case(label, P_set, ro, rd, sff, others):
    ts = tpkg.SurfelTracingSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                    viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=3, campos=cam.camera_center,
                                    prefiltered=False, debug=False, max_trace_depth=0, specular_threshold=0.0)
    v, f = fused.surfel_quads(P_set["means3D"], P_set["scales"], P_set["rotations"])
    nodes, _ = tracing.build_bvh(v, P_set["opacities"])
    caps = tracing.CapState()
    caps.colour_only = True
    for it in range(4):
        tracing.trace_forward(nodes, ro, rd, P_set["means3D"], P_set["shs"], None, others, P_set["opacities"], P_set["scales"], P_set["rotations"], ts, sff, caps=caps)
        torch.cuda.synchronize()
    lib.envgs_prof_enable(1); drain()
    for _ in range(10):
        tracing.trace_forward(nodes, ro, rd, P_set["means3D"], P_set["shs"], None, others, P_set["opacities"], P_set["scales"], P_set["rotations"], ts, sff, caps=caps)
    torch.cuda.synchronize()
    lib.envgs_prof_enable(0)
    d = drain()
    print(tag, label, "  ".join("%s %.4f" % (k.replace("trace.", ""), d[k][0]) for k in ("trace.collect_hits", "trace.sort_composite_fwd", "trace.register_hits", "trace_fwd") if k in d), flush=True)

case("env rays, colour-only forward", ge, ref_o, ref_d, False, None)
