"""Shared helpers for the test-suite: small seeded scenes in the layout the reference call sites use."""
import numpy as np
import torch

from envgs_amd import synth


def small_scene(P=400, H=64, W=80, seed=0, view=1, C=3, sh=True, spread=1.0, scale_mul=4.0):
    """A small scene whose surfels are big enough on a tiny image to overlap heavily."""
    g = synth.base_gaussians(P, seed=seed)
    g["means3D"] = g["means3D"] * spread
    g["scales"] = g["scales"] * scale_mul
    cam = synth.orbit_camera(view, H=H, W=W, fx=1111.1 * W / 800.0)
    if not sh:
        gen = torch.Generator().manual_seed(seed + 7)
        g["colors_precomp"] = torch.rand(P, C, generator=gen)
    return g, cam


def cam_args(cam):
    return dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
                W=cam.image_width, H=cam.image_height)


def npy(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


def rel_err(a, b, eps=1e-8):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + eps))
