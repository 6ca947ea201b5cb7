"""GPU, end to end: one full EnvGS step -- ch05 base raster -> reflect -> env trace -> blend -> backward -- through the SAME caller code
(envgs_amd/envgs_step.py, the re-derivation of EnvGSSampler.forward) over the HIP extensions and over the CPU oracle.  This is the
test that the two extensions COMPOSE: the tracer's ray gradients must flow back through the reflected-ray construction into the
rasterizer's normal / depth gradients (easyvolcap/models/samplers/envgs_sampler.py:420-455 with detach=False)."""
import numpy as np
import pytest
import torch

from envgs_amd import envgs_step, synth
from tests.util import check_close, record, floor_rel_err

pytestmark = pytest.mark.gpu


def _scene(dev):
    H, W = 48, 64
    base = synth.base_gaussians(1500, seed=3)
    base["scales"] = base["scales"] * 5.0
    base["opacities"] = torch.sigmoid(torch.randn(1500, 1, generator=torch.Generator().manual_seed(1)) + 1.5)
    env = synth.env_gaussians(800, seed=4, bound=12.0)
    cam = synth.orbit_camera(1, H=H, W=W, fx=1111.1 * W / 800.0)
    mv = lambda d: {k: v.to(dev).clone().requires_grad_(True) for k, v in d.items()}
    camd = synth.orbit_camera(1, H=H, W=W, fx=1111.1 * W / 800.0, device=dev)
    return mv(base), mv(env), camd


def _run(pkg, tpkg, tracer, dev, keep=None):
    base, env, cam = _scene(dev)
    rays = synth.get_rays(cam)
    bg = torch.zeros(3, device=dev); env_bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays, base, env, bg, env_bg, torch.tensor([2], device=dev))
    gen = torch.Generator().manual_seed(7)
    H, W = cam.image_height, cam.image_width
    dcol = (torch.randn(H, W, 3, generator=gen) / (H * W)).to(dev)
    dall = (torch.randn(7, H, W, generator=gen) / (H * W)).to(dev); dall[5:] = 0
    if keep is not None:                                 # upstream gradient only where the oracle's audits call the pixel / its reflected ray determined
        k = keep.to(dev)
        dcol = dcol * k[..., None]; dall = dall * k[None]
    loss = (out["rgb"] * dcol).sum() + (out["base"]["allmap"] * dall).sum()
    loss.backward()
    g = {("base." + k): v.grad for k, v in base.items() if v.grad is not None}
    g.update({("env." + k): v.grad for k, v in env.items() if v.grad is not None})
    return out, g, (base, env, cam)


# Gradients of the COMPOSED step (raster -> reflect -> trace -> blend): no per-element noise scale exists for a chain of two oracles under
# autograd, and the reflected rays of the two runs differ by fp32 rounding of the base pass (so a grazing ray may hit a different surfel set
# although the audit of the oracle's own rays calls it determined).  Plain elementwise bound, floor = the tensor's mean magnitude.
STEP_GRAD_TOL = 3e-3


@pytest.mark.parametrize("fused_glue", [False, True])
def test_full_envgs_step_matches_oracle_end_to_end(fused_glue):
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    from oracle import raster as orc, trace as otr
    from tests.oracle_packages import make_raster_pkg, make_trace_pkg
    test = "envgs_step[%s]" % ("fused" if fused_glue else "torch")
    # 1. the oracle run + its audits: fragile base pixels, fragile reflected rays
    opkg, otpkg = make_raster_pkg(5), make_trace_pkg()
    rec = {}
    F = opkg.GaussianRasterizer.forward
    out_o, _, (base, env, cam) = _run(opkg, otpkg, otpkg.SurfelTracer(), torch.device("cpu"))
    n = lambda t: t.detach().cpu().numpy()
    b = out_o["base"]
    H, W = cam.image_height, cam.image_width
    from envgs_amd.envgs_step import C0                                           # the colours the base pass handed to the rasterizer
    ref = orc.raster_forward(n(base["means3D"]), n(base["opacities"]), n(cam.world_view_transform), n(cam.full_proj_transform), n(cam.camera_center), W, H,
                             scales=n(base["scales"]), rotations=n(base["rotations"]),
                             colors_precomp=np.zeros((base["means3D"].shape[0], 5), np.float32), bg=np.zeros(3, np.float32))
    frag_px = orc.raster_audit(ref)["fragile"]
    ra = otr.trace_audit(n(out_o["ref_o"]).reshape(-1, 3), n(out_o["ref_d"]).reshape(-1, 3), n(env["means3D"]), n(env["scales"]), n(env["rotations"]),
                         n(env["opacities"]), start_from_first=False)
    keep = torch.from_numpy(~(frag_px | ra["fragile"].reshape(H, W)))
    record(test, "excluded_pixels", 1.0 - float(keep.float().mean()))
    # 2. both runs with the upstream gradient masked to the determined pixels
    envgs_step.FUSED["on"] = fused_glue              # HIP run: torch glue or the fused HIP glue (envgs_amd.fused)
    try:
        out_h, g_h, _ = _run(pkg, tpkg, tpkg.SurfelTracer(), torch.device("cuda:0"), keep)
    finally:
        envgs_step.FUSED["on"] = False
    out_o, g_o, _ = _run(opkg, otpkg, otpkg.SurfelTracer(), torch.device("cpu"), keep)
    k = keep.numpy()
    # 3. images: every determined pixel within 1e-4 -- except rays that flipped because the two base passes differ by fp32 rounding (counted, <= 0.5 %)
    for nm in ("rgb", "rgb_env"):
        a, r = n(out_h[nm])[k], n(out_o[nm])[k]
        bad = (floor_rel_err(a, r)[0] > 1e-4).any(axis=-1)
        record(test, nm + ".flipped_rays", float(bad.mean()))
        assert bad.mean() <= 5e-3
        check_close(test, nm, a[~bad], r[~bad])
    assert float(out_o["rgb_env"].detach().abs().mean()) > 0.05 and float(out_o["base"]["spec"].detach().mean()) > 0.01        # the env pass matters
    assert set(g_h) == set(g_o) and {"base.means3D", "base.rotations", "base.specular", "env.shs", "env.means3D"} <= set(g_h)
    for kk in sorted(g_h):
        check_close(test, kk, n(g_h[kk]), n(g_o[kk]), tol=STEP_GRAD_TOL)
    # the ray-gradient path is really exercised: base geometry gets gradient THROUGH the env colour
    assert float(g_o["base.rotations"].abs().max()) > 0
