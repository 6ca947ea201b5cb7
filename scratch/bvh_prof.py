"""rocprofv3 target: a few full builds and refits of the environment structure (per-kernel split of T1).  python scratch/bvh_prof.py [P]"""
import sys, torch
sys.path.insert(0, ".")
from envgs_amd import synth, tracing, fused
P = int(sys.argv[1]) if len(sys.argv) > 1 else 700000
dev = torch.device("cuda:0")
e = synth.env_gaussians(P, seed=1, device=dev)
v, _ = fused.surfel_quads(e["means3D"], e["scales"], e["rotations"])
nodes, _ = tracing.build_bvh(v, e["opacities"])
for _ in range(6):
    tracing.build_bvh(v, e["opacities"])
for _ in range(6):
    tracing.build_bvh(v, e["opacities"], refit=nodes)
torch.cuda.synchronize()
