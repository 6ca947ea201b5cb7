"""CPU: pin the in-tree boundary pieces against golden vectors generated from the reference itself
(tests/golden/make_golden.py).  Covers SURVEY.md section 8(c) items 1-5."""
import numpy as np
import torch

from envgs_amd import synth
from oracle import raster as orc
from tests import reference_caller


def _cam(g):
    return synth.make_camera(g["K"], g["R"], g["T"], int(g["H"]), int(g["W"]), float(g["n"]), float(g["f"]))


def test_camera_matrices(golden):
    cam = _cam(golden)
    np.testing.assert_allclose(cam.world_view_transform.numpy(), golden["world_view_transform"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cam.projection_matrix.numpy(), golden["projection_matrix"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), golden["full_proj_transform"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cam.camera_center.numpy(), golden["camera_center"], rtol=1e-6, atol=1e-6)
    assert abs(cam.FoVx - float(golden["FoVx"])) < 1e-6 and abs(cam.FoVy - float(golden["FoVy"])) < 1e-6


def test_rotation_and_splat2world(golden):
    q = torch.from_numpy(golden["quat"])
    np.testing.assert_allclose(synth.build_rotation(q).numpy(), golden["rotmat"], rtol=1e-5, atol=1e-6)
    s2w = synth.splat2world(torch.from_numpy(golden["xyz"]), torch.from_numpy(golden["scales"]), q)
    np.testing.assert_allclose(s2w.numpy(), golden["splat2world"], rtol=1e-5, atol=1e-7)


def test_transmat_python_twin(golden):
    cam = _cam(golden)
    args = [torch.from_numpy(golden[k]) for k in ("xyz", "scales", "quat")]
    np.testing.assert_allclose(synth.transmat_python(cam, *args).numpy(), golden["transmat"], rtol=2e-5, atol=1e-4)
    np.testing.assert_allclose(synth.transmat_python(cam, *args, scale_modifier=0.7).numpy(), golden["transmat_mod07"], rtol=2e-5, atol=1e-4)


def test_oracle_transmat_matches_reference(golden):
    """R1 of the oracle (in-kernel transMat from scales/rotations) == the reference's python transMat."""
    cam = _cam(golden)
    P = golden["xyz"].shape[0]
    for mod, key in ((1.0, "transmat"), (0.7, "transmat_mod07")):
        out = orc.raster_forward(golden["xyz"], np.ones((P, 1), np.float32), cam.world_view_transform.numpy(),
                                 cam.full_proj_transform.numpy(), cam.camera_center.numpy(), cam.image_width,
                                 cam.image_height, scales=golden["scales"], rotations=golden["quat"],
                                 colors_precomp=np.ones((P, 3), np.float32), scale_modifier=mod)
        vis = out["radii"] > 0
        assert vis.sum() > P // 2
        ref = golden[key]
        err = np.abs(out["transmat"][vis] - ref[vis]).max() / np.abs(ref[vis]).max()
        assert err < 1e-5, err
        # precomputed transMat path must give the identical projection (radius, tile rect, centre)
        out2 = orc.raster_forward(golden["xyz"], np.ones((P, 1), np.float32), cam.world_view_transform.numpy(),
                                  cam.full_proj_transform.numpy(), cam.camera_center.numpy(), cam.image_width,
                                  cam.image_height, transmat_precomp=out["transmat"],
                                  colors_precomp=np.ones((P, 3), np.float32), scale_modifier=mod)
        np.testing.assert_array_equal(out2["radii"], out["radii"])
        np.testing.assert_array_equal(out2["tiles_touched"], out["tiles_touched"])


def test_oracle_sh_matches_reference(golden):
    cam = _cam(golden)
    P = golden["xyz"].shape[0]
    for deg in range(4):
        out = orc.raster_forward(golden["xyz"], np.ones((P, 1), np.float32), cam.world_view_transform.numpy(),
                                 cam.full_proj_transform.numpy(), cam.camera_center.numpy(), cam.image_width,
                                 cam.image_height, scales=golden["scales"], rotations=golden["quat"],
                                 shs=golden["shs"], sh_degree=deg)
        vis = out["radii"] > 0
        np.testing.assert_allclose(out["rgb"][vis], golden[f"colors_deg{deg}"][vis], rtol=1e-4, atol=2e-6)


def test_get_disks(golden):
    v, f = synth.get_disks(*[torch.from_numpy(golden[k]) for k in ("xyz", "scales", "quat")])
    np.testing.assert_allclose(v.numpy(), golden["disks_v"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(f.numpy(), golden["disks_f"])
    assert f.dtype == torch.int32 and f[:2].tolist() == [[0, 1, 2], [1, 2, 3]]


def test_get_rays(golden):
    cam = _cam(golden)
    ro, rd = synth.get_rays(cam)
    np.testing.assert_allclose(ro.numpy(), golden["ray_o"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rd.numpy(), golden["ray_d"], rtol=1e-4, atol=1e-5)
    # z_depth convention: camera-space z of every direction is exactly 1
    z = (rd.reshape(-1, 3) @ cam.R.T)[:, 2]
    assert torch.allclose(z, torch.ones_like(z), atol=1e-5)


def test_dpt2norm_twin(golden):
    """tests/reference_caller.dpt2norm (the torch re-derivation the fused surface_normal kernel is tested against) vs the reference's own dpt2norm."""
    from envgs_amd import envgs_step
    cam = _cam(golden)
    out = reference_caller.dpt2norm(cam, torch.tensor(golden["dpt"])[None])
    ref = golden["dpt2norm"]
    assert out.shape == ref.shape and float(np.abs(ref[1:-1, 1:-1]).sum()) > 0 and float(np.abs(ref[0]).sum()) == 0
    np.testing.assert_allclose(out.numpy(), ref, rtol=1e-4, atol=2e-5)
