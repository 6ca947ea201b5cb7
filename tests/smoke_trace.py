"""Tiny GPU check of the tracer against the brute-force oracle, called from __graft_entry__.smoke()."""
import numpy as np
import torch


def smoke_trace():
    import diff_surfel_tracing as mod
    from envgs_amd import synth
    from oracle import trace as otr
    from tests.test_oracle_trace import trace_scene
    dev = torch.device("cuda:0")
    g, ro, rd = trace_scene(P=200, R=512, seed=1, camera=False)
    L = {k: g[k].to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    o = ro.to(dev).requires_grad_(True); d = rd.to(dev).requires_grad_(True)
    v, f = synth.get_disks(L["means3D"].detach(), L["scales"].detach(), L["rotations"].detach())
    tr = mod.SurfelTracer()
    tr.build_acceleration_structure(v, f, rebuild=True)
    I = torch.eye(4, device=dev)
    st = mod.SurfelTracingSettings(image_height=1, image_width=1, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
                                   viewmatrix=I, projmatrix=I, sh_degree=torch.tensor([3], device=dev), campos=torch.zeros(3, device=dev),
                                   prefiltered=False, debug=False, max_trace_depth=0, specular_threshold=0.0)
    rgb, dpt, acc, norm, dist, aux, mid, wet = tr(o, d, v, means3D=L["means3D"], grads3D=None, shs=L["shs"], colors_precomp=None,
                                                   others_precomp=None, opacities=L["opacities"], scales=L["scales"], rotations=L["rotations"],
                                                   cov3D_precomp=None, tracer_settings=st, start_from_first=False)
    rgb.sum().backward()
    torch.cuda.synchronize()
    ref = otr.trace_forward(ro.numpy(), rd.numpy(), g["means3D"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                            g["opacities"].numpy(), shs=g["shs"].numpy(), sh_degree=3, start_from_first=False)
    R = ro.shape[0]
    rb = otr.trace_backward(ref, np.ones((R, 3), np.float32), np.zeros(R, np.float32), np.zeros(R, np.float32),
                            np.zeros((R, 3), np.float32), np.zeros((R, 2), np.float32))
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-12))
    e1 = rel(rgb.detach().cpu().numpy(), ref["rgb"]); e2 = rel(L["means3D"].grad.cpu().numpy(), rb["dmeans3D"]); e3 = rel(d.grad.cpu().numpy(), rb["dray_d"])
    assert e1 < 1e-3 and e2 < 5e-3 and e3 < 5e-3, (e1, e2, e3)
    print("smoke ok: tracer rgb rel err %.2e, dmeans3D %.2e, dray_d %.2e (vs brute-force CPU oracle)" % (e1, e2, e3))
