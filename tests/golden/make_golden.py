"""Generate tests/golden/*.npz by IMPORTING the reference's pure-torch helpers (runs in the authoring
container only; /root/reference does not exist on the GPU box).  The fixtures are data: seeded inputs
plus the reference's outputs.  Re-run:  python tests/golden/make_golden.py

Pinned pieces:
  1. transMat / camera matrices : easyvolcap/utils/gaussian2d_utils.py:1050-1061, :67-100, :145-200
  2. SH colour                  : easyvolcap/utils/sh_utils.py:642-727 + gaussian2d_utils.py:1072-1076
  3. surfel quads               : easyvolcap/utils/optix_utils.py:39-69
  4. camera rays                : easyvolcap/utils/ray_utils.py:24-80
  5. reflection rays / dpt2norm : easyvolcap/models/samplers/envgs_sampler.py:420-431, gaussian2d_utils.py:1158-1206
The kernel bodies themselves are NOT in the reference tree, so no golden exists for them ("parity unpinned").
"""
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    for m in ("pdbr", "pdbr.utils", "ruamel", "ruamel.yaml", "plyfile", "diff_surfel_tracing"):
        sys.modules[m] = MagicMock()
    sys.modules["ujson"] = json
    sys.path.insert(0, REF)
    from easyvolcap.utils import gaussian2d_utils as g2d
    from easyvolcap.utils.sh_utils import eval_sh
    from easyvolcap.utils.ray_utils import get_rays
    from easyvolcap.utils.base_utils import dotdict
    from easyvolcap.utils import optix_utils
    return g2d, eval_sh, get_rays, dotdict, optix_utils


def main():
    g2d, eval_sh, get_rays, dotdict, optix_utils = _import_reference()
    torch.manual_seed(0)
    P, H, W = 64, 96, 128
    xyz = (torch.rand(P, 3) * 2 - 1) * 1.3
    scales = torch.exp(torch.rand(P, 2) * 2.3 - 5.5)
    quat = torch.randn(P, 4)                      # deliberately NOT unit length
    K = torch.tensor([[150.0, 0, W / 2], [0, 160.0, H / 2], [0, 0, 1]])
    az = 0.7
    c = torch.tensor([4 * np.cos(az) * 0.9, 4 * np.sin(az) * 0.9, 1.6], dtype=torch.float32)
    fwd = -c / c.norm(); up = torch.tensor([0., 0., 1.])
    right = torch.linalg.cross(fwd, up); right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    R = torch.stack([right, down, fwd]); T = -(R @ c).reshape(3, 1)
    n, f = torch.tensor(2.0), torch.tensor(6.0)

    batch = dotdict(H=[H], W=[W], K=K[None], R=R[None], T=T[None], n=n[None], f=f[None],
                    meta=dotdict(H=[H], W=[W], K=K[None], R=R[None], T=T[None], n=n[None], f=f[None]))
    cam = g2d.prepare_gaussian_camera(batch)

    # 1. transMat exactly as render() builds it (gaussian2d_utils.py:1050-1061) -- on CPU
    for mod in (1.0, 0.7):
        s2w = g2d.build_cov(xyz, scales, mod, quat)
        ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2],
                                [0, 0, float(f - n), float(n)], [0, 0, 0, 1]]).float().T
        world2pix = cam.full_proj_transform @ ndc2pix
        tm = (s2w[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
        if mod == 1.0: transmat, splat2world = tm, s2w
        else: transmat_mod = tm
    rot = g2d.build_rotation(quat)

    # 2. SH colours, degrees 0..3
    shs = torch.cat([torch.rand(P, 1, 3) * 3 - 1.5, torch.randn(P, 15, 3) * 0.3], dim=1)   # (P,16,3) as get_features
    shs_view = shs.transpose(1, 2).reshape(-1, 3, 16)
    dir_pp = xyz - cam.camera_center.repeat(P, 1)
    dir_n = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    colors = [torch.clamp_min(eval_sh(d, shs_view, dir_n) + 0.5, 0.0) for d in range(4)]

    # 3. get_disks through the reference method with a minimal stand-in for GaussianModel.get_covariance
    class _Pcd:
        def get_covariance(self, scaling_modifier=1): return g2d.build_cov(xyz, scales, scaling_modifier, quat)
    v, fidx = optix_utils.HardwareRendering.get_disks(None, _Pcd())

    # 4. rays
    ray_o, ray_d = get_rays(H, W, K, R, T, z_depth=True, correct_pix=True)

    # 5. dpt2norm on a smooth synthetic depth map (device='cpu')
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    dpt = 3.0 + 0.004 * xx + 0.002 * yy + 0.2 * torch.sin(xx / 9.0)
    camf = dotdict({k: (v.float() if torch.is_tensor(v) else v) for k, v in cam.items()})
    snorm = g2d.dpt2norm(camf, dpt[None], device="cpu")

    np.savez_compressed(
        os.path.join(HERE, "boundary_golden.npz"),
        xyz=xyz.numpy(), scales=scales.numpy(), quat=quat.numpy(), K=K.numpy(), R=R.numpy(), T=T.numpy(),
        n=float(n), f=float(f), H=H, W=W,
        world_view_transform=cam.world_view_transform.numpy(), projection_matrix=cam.projection_matrix.numpy(),
        full_proj_transform=cam.full_proj_transform.numpy(), camera_center=cam.camera_center.numpy(),
        FoVx=float(cam.FoVx), FoVy=float(cam.FoVy),
        rotmat=rot.numpy(), splat2world=splat2world.numpy(), transmat=transmat.numpy(), transmat_mod07=transmat_mod.numpy(),
        shs=shs.numpy(), colors_deg0=colors[0].numpy(), colors_deg1=colors[1].numpy(), colors_deg2=colors[2].numpy(),
        colors_deg3=colors[3].numpy(), disks_v=v.numpy(), disks_f=fidx.numpy(),
        ray_o=ray_o.numpy(), ray_d=ray_d.numpy(), dpt=dpt.numpy(), dpt2norm=snorm.numpy())
    print("wrote boundary_golden.npz")


if __name__ == "__main__":
    main()
