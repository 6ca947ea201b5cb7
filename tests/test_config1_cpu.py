"""CPU: BASELINE configs[0] -- 2 k surfels, 256x256, PyTorch-eager CPU project + alpha-blend (no extension; plumbing only).
The eager restatement (oracle/eager.py, fp32) and the C oracle must agree on this case; it is also what bench.py's CPU figures rest on."""
import numpy as np
import torch

from envgs_amd import synth
from oracle import eager, raster as orc
from tests.util import assert_close_frac


def test_config1_eager_vs_c_oracle():
    P, H, W = 2000, 256, 256
    g = synth.base_gaussians(P, seed=0)
    g["scales"] = g["scales"] * 2.0
    cam = synth.orbit_camera(0, H=H, W=W, fx=1111.1 * W / 800.0)
    bg = torch.ones(3)
    with torch.no_grad():
        color, radii, allmap, weight = eager.rasterize(g["means3D"], g["opacities"], cam.world_view_transform, cam.full_proj_transform,
                                                       cam.camera_center, W, H, scales=g["scales"], rotations=g["rotations"], shs=g["shs"],
                                                       sh_degree=3, bg=bg, pix_chunk=8192)
    ref = orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                             cam.camera_center.numpy(), W, H, scales=g["scales"].numpy(), rotations=g["rotations"].numpy(),
                             shs=g["shs"].numpy(), sh_degree=3, bg=bg.numpy())
    assert (ref["radii"] > 0).sum() > 1500 and ref["N"] > 3000
    np.testing.assert_array_equal(radii.numpy(), ref["radii"])
    assert_close_frac(color.numpy(), ref["out_color"], 1e-4, max_bad_frac=1e-3, flip_bound=0.05, what="color")
    for ch in (0, 1, 2, 3, 4):
        assert_close_frac(allmap[ch].numpy(), ref["allmap"][ch], 1e-4, max_bad_frac=1e-3, flip_bound=0.05, what="allmap%d" % ch)
    assert_close_frac(weight.numpy(), ref["weight"], 1e-4, max_bad_frac=1e-3, flip_bound=0.05, what="weight")


def test_visibility_filter_semantics():
    """SURVEY 8 row a9 (optix_utils.py:203-213): wet > 0, OR-ed with the in-image / z >= 0.2 test only when tracing starts at the camera."""
    import torch
    from envgs_amd.envgs_step import visibility_filter
    wet = torch.tensor([[0.0], [0.3], [0.0], [0.0], [0.0]])
    K = torch.tensor([[100.0, 0, 50], [0, 100.0, 40], [0, 0, 1]])
    R = torch.eye(3); T = torch.zeros(3, 1)
    means = torch.tensor([[0.0, 0.0, 1.0],      # centre of the image, z = 1         -> visible by projection
                          [9.0, 9.0, 1.0],      # off-image but blended by a ray     -> visible by wet
                          [0.0, 0.0, 0.1],      # too close (z < 0.2)
                          [0.6, 0.0, 1.0],      # u = 110 > W = 100                  -> outside
                          [-0.5, -0.4, 1.0]])   # u = 0, v = 0: the inclusive border -> visible
    assert visibility_filter(wet).tolist() == [False, True, False, False, False]
    got = visibility_filter(wet, means, K, R, T, H=80, W=100, start_from_first=True)
    assert got.tolist() == [True, True, False, False, True] and got.dtype == torch.bool
