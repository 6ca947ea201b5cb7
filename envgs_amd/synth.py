"""Seeded synthetic scenes, cameras and rays for tests and bench.py.

Self-contained re-derivations of the small reference helpers that sit either side of the hot
path; each is pinned against the reference by tests/golden (tests/test_golden.py):
  camera matrices   : easyvolcap/utils/gaussian2d_utils.py:24-100 (getWorld2View, getProjectionMatrix, prepare_gaussian_camera)
  splat2world       : easyvolcap/utils/gaussian2d_utils.py:145-200 (build_rotation, build_cov)
  transMat (python) : easyvolcap/utils/gaussian2d_utils.py:1050-1061
  surfel quads      : easyvolcap/utils/optix_utils.py:39-69 (get_disks)
  camera rays       : easyvolcap/utils/ray_utils.py:24-80 (get_rays, z_depth=True, correct_pix=True)
"""
import math
from types import SimpleNamespace

import torch


def make_camera(K, R, T, H, W, n, f, device="cpu"):
    """K (3,3), R (3,3) world->cam, T (3,1).  Row-vector convention matrices like the reference."""
    K = torch.as_tensor(K, dtype=torch.float32)
    R = torch.as_tensor(R, dtype=torch.float32)
    T = torch.as_tensor(T, dtype=torch.float32).reshape(3, 1)
    fx, fy = float(K[0, 0]), float(K[1, 1])
    FoVx = 2.0 * math.atan(W / (2.0 * fx))
    FoVy = 2.0 * math.atan(H / (2.0 * fy))
    w2v = torch.eye(4)
    w2v[:3, :3] = R
    w2v[:3, 3:] = T
    tanx, tany = math.tan(FoVx / 2), math.tan(FoVy / 2)
    top, right = tany * n, tanx * n
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * n / (2 * right)
    Pm[1, 1] = 2.0 * n / (2 * top)
    Pm[3, 2] = 1.0
    Pm[2, 2] = f / (f - n)
    Pm[2, 3] = -(f * n) / (f - n)
    wvt = w2v.t().contiguous()
    proj = Pm.t().contiguous()
    cam = SimpleNamespace(
        image_height=int(H), image_width=int(W), K=K.to(device), R=R.to(device), T=T.to(device),
        FoVx=FoVx, FoVy=FoVy, tanfovx=math.tan(FoVx * 0.5), tanfovy=math.tan(FoVy * 0.5),
        world_view_transform=wvt.to(device), projection_matrix=proj.to(device),
        full_proj_transform=(wvt @ proj).to(device), camera_center=(-R.t() @ T)[:, 0].to(device),
        znear=float(n), zfar=float(f))
    return cam


def orbit_camera(view, n_views=8, radius=4.0, H=800, W=800, fx=1111.1, n=2.0, f=6.0, device="cpu"):
    """View `view` of `n_views` on a radius-`radius` sphere looking at the origin."""
    az = 2.0 * math.pi * view / n_views
    el = math.radians(20.0 + 10.0 * (view % 3))
    c = torch.tensor([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)])
    fwd = -c / c.norm()                                   # camera +z looks at the origin
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up); right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    R = torch.stack([right, down, fwd], dim=0)            # rows = camera axes in world coords
    T = -(R @ c).reshape(3, 1)
    K = torch.tensor([[fx, 0, W / 2.0], [0, fx, H / 2.0], [0, 0, 1.0]])
    return make_camera(K, R, T, H, W, n, f, device)


def base_gaussians(P, seed=0, sh_coeffs=16, device="cpu"):
    """Base surfels: xyz~U([-1.3,1.3]^3), scale=exp(U(ln .004, ln .04)), quat=normalize(N), opacity=sigmoid(N(0,1.5))."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * 1.3
    ls = math.log(0.004) + torch.rand(P, 2, generator=g) * (math.log(0.04) - math.log(0.004))
    scales = torch.exp(ls)
    q = torch.randn(P, 4, generator=g)
    rots = q / q.norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5)
    f_dc = torch.rand(P, 1, 3, generator=g) * 3.0 - 1.5
    f_rest = torch.randn(P, sh_coeffs - 1, 3, generator=g) * 0.1
    shs = torch.cat([f_dc, f_rest], dim=1)
    spec = torch.sigmoid(torch.randn(P, 1, generator=g) - 2.0)
    rough = torch.full((P, 1), 0.5)
    out = dict(means3D=xyz, scales=scales, rotations=rots, opacities=opac, shs=shs, specular=spec, roughness=rough)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def env_gaussians(P, seed=1, bound=50.0, sh_coeffs=16, device="cpu"):
    """Environment surfels over +-bound (the reference initialises 32^3*5 random points in the scene bounds,
    easyvolcap/models/samplers/envgs_sampler.py:194-207); scales follow the 3-NN spacing of that density."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(P, 3, generator=g) * 2 - 1) * bound
    keep_out = xyz.norm(dim=-1) < 3.0                       # keep the env set off the object volume
    xyz[keep_out] = xyz[keep_out] / xyz[keep_out].norm(dim=-1, keepdim=True) * (3.0 + 40.0 * torch.rand(int(keep_out.sum()), 1, generator=g))
    spacing = 2 * bound / (P ** (1.0 / 3.0))
    scales = spacing * torch.exp(torch.rand(P, 2, generator=g) * 1.0 - 0.7)
    q = torch.randn(P, 4, generator=g)
    rots = q / q.norm(dim=-1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5 - 1.0)
    f_dc = torch.rand(P, 1, 3, generator=g) * 3.0 - 1.5
    f_rest = torch.randn(P, sh_coeffs - 1, 3, generator=g) * 0.1
    shs = torch.cat([f_dc, f_rest], dim=1)
    out = dict(means3D=xyz, scales=scales, rotations=rots, opacities=opac, shs=shs)
    return {k: v.to(device).contiguous() for k, v in out.items()}


def build_rotation(q):
    q = q / q.norm(dim=-1, keepdim=True)
    r, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def splat2world(means3D, scales, rotations, scale_modifier=1.0):
    """(P,4,4) row-vector matrix: rows 0,1 = scaled tangent axes, row 2 = normal, row 3 = centre."""
    R = build_rotation(rotations)
    s3 = torch.cat([scales * scale_modifier, torch.ones_like(scales[:, :1])], dim=-1)
    L = (R * s3[:, None, :]).transpose(1, 2)
    T = torch.zeros(means3D.shape[0], 4, 4, dtype=means3D.dtype, device=means3D.device)
    T[:, :3, :3] = L
    T[:, 3, :3] = means3D
    T[:, 3, 3] = 1
    return T


def transmat_python(cam, means3D, scales, rotations, scale_modifier=1.0):
    """The (P,9) cov3D_precomp / transMat of gaussian2d_utils.py:1050-1061."""
    s2w = splat2world(means3D, scales, rotations, scale_modifier)
    W, H = cam.image_width, cam.image_height
    n, f = cam.znear, cam.zfar
    ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, f - n, n], [0, 0, 0, 1]],
                           dtype=torch.float32, device=means3D.device).T
    world2pix = cam.full_proj_transform.to(means3D.device) @ ndc2pix
    return (s2w[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9).contiguous()


def get_disks(means3D, scales, rotations):
    """Surfel -> 3-sigma quad: v (4P,3) f32, f (2P,3) i32 (optix_utils.py:39-69).

    The reference forms the 4 corners with a (4P,4,4)@(4P,4,1) batched matmul; on this stack hipBLASLt turns that into an
    8.7 ms GEMM per step, so the same corners  mu + 3*(su*a*(+-1) + sv*b*(+-1))  are written elementwise (pinned against the
    reference's own output by tests/test_golden.py::test_get_disks)."""
    R = build_rotation(rotations)
    a3 = R[:, :, 0] * (3.0 * scales[:, 0:1])
    b3 = R[:, :, 1] * (3.0 * scales[:, 1:2])
    v = torch.stack([means3D - a3 + b3, means3D - a3 - b3, means3D + a3 + b3, means3D + a3 - b3], dim=1).reshape(-1, 3)
    P = means3D.shape[0]
    idx = torch.arange(0, 4 * P, device=means3D.device).reshape(P, 4)
    f = torch.stack([idx[:, :3], idx[:, 1:]], dim=1).reshape(-1, 3).int()
    return v.contiguous(), f.contiguous()


def get_rays(cam):
    """Camera rays (H,W,3): origin = camera centre, direction with camera-z = 1 (z_depth), pixel centre +0.5."""
    H, W = cam.image_height, cam.image_width
    dev = cam.K.device
    i, j = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
    xy1 = torch.stack([j + 0.5, i + 0.5, torch.ones_like(i)], dim=-1)
    pix_cam = xy1 @ torch.linalg.inv(cam.K).T
    pix_world = (pix_cam - cam.T[:, 0]) @ cam.R
    ray_o = (-cam.R.T @ cam.T)[:, 0]
    ray_d = pix_world - ray_o
    return ray_o.expand_as(ray_d).contiguous(), ray_d.contiguous()
