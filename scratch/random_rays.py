"""SURVEY 8(d) stress case: uniformly random unit directions from random origins (no view coherence before the ray sort)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from envgs_amd import synth, tracing
import diff_surfel_tracing as tpkg
dev = torch.device("cuda", 0)
PE, R = 163840, 640000
ge = synth.env_gaussians(PE, seed=1, device=dev)
gen = torch.Generator().manual_seed(9)
ro = ((torch.rand(R, 3, generator=gen) * 2 - 1) * 1.3).to(dev)
rd = torch.randn(R, 3, generator=gen); rd = (rd / rd.norm(dim=-1, keepdim=True)).to(dev)
tracer = tpkg.SurfelTracer()
v, f = synth.get_disks(ge["means3D"], ge["scales"], ge["rotations"])
tracer.build_acceleration_structure(v, f, rebuild=True)
ts = tpkg.SurfelTracingSettings(image_height=800, image_width=800, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
    viewmatrix=torch.eye(4, device=dev), projmatrix=torch.eye(4, device=dev), sh_degree=torch.tensor([3], device=dev), campos=torch.zeros(3, device=dev),
    prefiltered=False, debug=False, max_trace_depth=0, specular_threshold=0.0)
P = {k: ge[k].clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
def step():
    o = ro.clone().requires_grad_(True); d = rd.clone().requires_grad_(True)
    outs = tracer(o[None], d[None], v, means3D=P["means3D"], grads3D=torch.zeros_like(P["means3D"], requires_grad=True) + 0, shs=P["shs"], colors_precomp=None,
                  others_precomp=None, opacities=P["opacities"], scales=P["scales"], rotations=P["rotations"], cov3D_precomp=None, tracer_settings=ts,
                  start_from_first=False)
    (outs[0].sum() / R).backward()
for srt in (True, False):
    tracing.SORT_RAYS["on"] = srt
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    print("random rays, coherence sort", srt, ": trace fwd+bwd %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), tracing.last_trace_counts())
