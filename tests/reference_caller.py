"""The UNCHANGED EasyVolcap caller's own expression forms, restated for measurement and tests only (they do not belong in the
package that ships).  `bench.py --caller reference` times a step through them; tests pin them against the reference's own outputs
(tests/golden/boundary_golden.npz) and use them as the torch twin the fused HIP glue is compared with.

    get_disks_reference_form   easyvolcap/utils/optix_utils.py:39-69     (4P,4,4) @ (4P,4,1) batched matmul
    dpt2norm, surface_maps     easyvolcap/utils/gaussian2d_utils.py:1125-1142, 1158-1206

install() registers them with envgs_amd.envgs_step.REFERENCE_FORMS (the package holds only the two slots)."""
import math

import torch

from envgs_amd.synth import splat2world


def get_disks_reference_form(means3D, scales, rotations):
    """The SAME quads in the reference's own expression form (optix_utils.py:39-69): splat2world^T with the normal column zeroed, applied to
    the four 3-sigma uv corners as a (4P,4,4) @ (4P,4,1) batched matmul.  Only bench.py --caller reference uses it: it is what the unchanged
    EasyVolcap caller executes every training step before it calls the tracer (8.7 ms through hipBLASLt, 1.0 ms through rocBLAS on MI355X)."""
    T = splat2world(means3D, scales, rotations).permute(0, 2, 1).clone()
    T[..., 2] = 0
    P = T.shape[0]
    sigma3 = torch.as_tensor([[-1., 1.], [-1., -1.], [1., 1.], [1., -1.]], device=T.device) * 3
    sigma3 = torch.cat([sigma3, torch.ones_like(sigma3)], dim=-1)[None].repeat(P, 1, 1)
    v = T[:, None].expand(-1, 4, -1, -1).reshape(-1, 4, 4) @ sigma3.reshape(-1, 4, 1)
    v = v[..., :3, 0]
    idx = torch.arange(0, v.shape[0], device=T.device).reshape(P, 4)
    f = torch.stack([idx[:, :3], idx[:, 1:]], dim=1).reshape(-1, 3).int()
    return v.contiguous(), f.contiguous()


def dpt2norm(cam, dpt):
    """dpt2xyz + dpt2norm of gaussian2d_utils.py:1158-1206 (torch, any device): depth (1,H,W) -> pseudo surface normals (H,W,3), zero border."""
    dev = dpt.device
    c2w = torch.linalg.inv(cam.world_view_transform.T)
    W, H = cam.image_width, cam.image_height
    fx = W / (2 * math.tan(cam.FoVx / 2.)); fy = H / (2 * math.tan(cam.FoVy / 2.))
    K = torch.tensor([[fx, 0., W / 2.], [0., fy, H / 2.], [0., 0., 1.0]], dtype=torch.float32, device=dev)
    u, v = torch.meshgrid(torch.arange(W, dtype=torch.float32, device=dev), torch.arange(H, dtype=torch.float32, device=dev), indexing='xy')
    pix = torch.stack([u, v, torch.ones_like(u)], dim=-1).reshape(-1, 3)
    ray_d = pix @ torch.linalg.inv(K).mT @ c2w[:3, :3].mT
    xyz = (dpt.reshape(-1, 1) * ray_d + c2w[:3, 3]).reshape(H, W, 3)
    out = torch.zeros_like(xyz)
    dx = xyz[2:, 1:-1] - xyz[:-2, 1:-1]
    dy = xyz[1:-1, 2:] - xyz[1:-1, :-2]
    out[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return out


def surface_maps(cam, allmap, depth_ratio=0.0):
    """surf_depth (1,H,W), surf_normal (3,H,W): the regulariser maps of render()'s tail (gaussian2d_utils.py:1125-1142), torch expressions
    (envgs_amd.fused.surface_normal is the one-kernel form)."""
    alpha = allmap[1:2]
    median = torch.nan_to_num(allmap[5:6], 0, 0)
    expect = torch.nan_to_num(allmap[0:1] / alpha, 0, 0)
    depth = expect * (1 - depth_ratio) + median * depth_ratio
    normal = dpt2norm(cam, depth).permute(2, 0, 1) * alpha.detach()
    return depth, normal


def install():
    from envgs_amd import envgs_step
    envgs_step.REFERENCE_FORMS["get_disks"] = get_disks_reference_form
    envgs_step.REFERENCE_FORMS["surface_maps"] = surface_maps
