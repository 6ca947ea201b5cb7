"""Round 6: what the '4x4 pixel block x 4 splat slots per pass' mapping of R7 would buy, counted on the bench view from the audit kernel's contributor
flags: passes and live lanes of (a) today's (8x8 quadrant, splat) passes, (b) (4x4 block, 4 consecutive live splats) passes, (c) (8x4 half, 2 splats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from envgs_amd import synth, raster
import diff_surfel_rasterization_wet_ch05 as pkg
dev = torch.device("cuda:0")
P, H, W, C = 300000, 800, 800, 5
g = synth.base_gaussians(P, seed=0, device=dev)
cam = synth.orbit_camera(0, n_views=8, H=H, W=W, fx=1111.1, device=dev)
st = pkg.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(C, device=dev), scale_modifier=1.0,
                                       viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=torch.tensor([3], device=dev),
                                       campos=cam.camera_center, prefiltered=False, debug=False)
with torch.no_grad():
    _, sv = raster.rasterize_forward(C, g["means3D"], None, torch.zeros(P, C, device=dev), g["opacities"], g["scales"], g["rotations"], None, st)
    rg = sv["ranges"].view(-1, 2).long()
    lmax = int((rg[:, 1] - rg[:, 0]).max())
    contrib, _, _ = raster.render_audit(sv, lmax)                   # (H*W, lmax) uint8: pixel blended entry j of its tile's list
    c = contrib.view(H, W, lmax)
    pairs = int(c.sum())
    def blocks(bh, bw):
        q = c.view(H // bh, bh, W // bw, bw, lmax).permute(0, 2, 4, 1, 3).reshape(H // bh, W // bw, lmax, bh * bw)
        live = q.sum(-1, dtype=torch.int32)                         # live pixels per (block, list entry)
        return live
    l8 = blocks(8, 8); p8 = int((l8 > 0).sum())
    print("live (pixel, splat) pairs %d" % pairs)
    print("(8x8 quadrant, 1 splat): passes %d, lanes/pass %.2f of 64" % (p8, pairs / p8))
    for bh, bw, slots in ((4, 4, 4), (8, 4, 2), (4, 8, 2), (2, 2, 16), (8, 8, 1)):
        lb = blocks(bh, bw)
        sb = (lb > 0).sum(-1)                                       # live splats per block
        passes = int(((sb + slots - 1) // slots).sum())
        print("(%dx%d block, %d splat slots): passes %d = %.3f of today's, lanes/pass %.2f of 64" % (bh, bw, slots, passes, passes / p8, pairs / passes))
