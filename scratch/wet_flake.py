"""Hunt for a rare wrong `wet` (clustered scene, list capacity below the longest lists: a mix of list-path and K-buffer rays whose split depends on
the collection's timing).  Repeats the forward and compares wet / rgb against the first run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from envgs_amd import synth, tracing
import diff_surfel_tracing as mod
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(21)
Pc, Pf = 3000, 200
means = torch.cat([torch.tensor([0.0, 0.0, 5.0]) + 0.02 * torch.randn(Pc, 3, generator=gen), (torch.rand(Pf, 3, generator=gen) * 2 - 1) * 30])
means[:50] = means[0]
P = Pc + Pf
scales = torch.cat([0.3 + 0.3 * torch.rand(Pc, 2, generator=gen), 2 + 2 * torch.rand(Pf, 2, generator=gen)])
q = torch.randn(P, 4, generator=gen); q[:50] = q[0]
rots = q / q.norm(dim=-1, keepdim=True)
opac = torch.sigmoid(torch.randn(P, 1, generator=gen) - 2.5)
shs = torch.randn(P, 16, 3, generator=gen) * 0.3
others = torch.rand(P, 2, generator=gen)
R = 512
ro = torch.randn(R, 3, generator=gen) * 0.2
tgt = torch.tensor([0.0, 0.0, 5.0]) + 0.3 * torch.randn(R, 3, generator=gen)
rd = tgt - ro; rd = rd / rd.norm(dim=-1, keepdim=True)
T = lambda x: x.to(dev)
means, scales, rots, opac, shs, others, ro, rd = map(T, (means, scales, rots, opac, shs, others, ro, rd))
I = torch.eye(4, device=dev)
st = mod.SurfelTracingSettings(image_height=1, image_width=1, tanfovx=1.0, tanfovy=1.0, bg=torch.tensor([0.2, 0.2, 0.2], device=dev), scale_modifier=1.0,
                               viewmatrix=I, projmatrix=I, sh_degree=torch.tensor([2], device=dev), campos=torch.zeros(3, device=dev),
                               prefiltered=False, debug=False, max_trace_depth=0, specular_threshold=0.0)
v, f = synth.get_disks(means, scales, rots)
tracer = mod.SurfelTracer()
tracing.KEEP_LISTS["on"] = True
ref = None
bad = 0
for it in range(120):
    cap = (128, 192, 256, 320, 384, 448, 512, 1024)[it % 8]
    tracing.HIT_CAP["force"] = cap
    tracer.build_acceleration_structure(v, f, rebuild=True)
    means.requires_grad_(True)
    if True:
        outs = tracer(ro, rd, v, means3D=means, grads3D=None, shs=shs, colors_precomp=None, others_precomp=others, opacities=opac, scales=scales,
                      rotations=rots, cov3D_precomp=None, tracer_settings=st, start_from_first=False)
    torch.cuda.synchronize()
    (outs[0].sum() + outs[1].sum()).backward()
    torch.cuda.synchronize()
    rgb, wet = outs[0].detach().cpu().numpy(), outs[7].detach().cpu().numpy()[:, 0]
    ids, tb, n_used, hit_cnt = tracing.last_hit_lists()
    hc = hit_cnt.cpu().numpy()
    if ref is None:
        ref = (rgb, wet)
        print("reference: cap", cap, "rays over cap", int((hc > cap).sum()), "max found", hc.max())
        continue
    er = np.abs(rgb - ref[0]).max(); ew = np.abs(wet - ref[1]); 
    rel = ew / (np.abs(ref[1]) + np.abs(ref[1]).mean())
    if rel.max() > 1e-4 or er > 1e-5:
        bad += 1
        i = int(rel.argmax())
        print("it %d cap %d: rays over cap %d; rgb err %.2e; wet rel err %.3g at surfel %d (%.6f vs %.6f); %d surfels off" % (it, cap, int((hc > cap).sum()), er, rel.max(), i, wet[i], ref[1][i], int((rel > 1e-4).sum())))
print("bad runs:", bad, "of 119")
