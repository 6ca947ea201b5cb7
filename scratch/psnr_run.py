"""Round 5: a larger held-out-view PSNR run than tests/test_train_convergence.py can afford -- ground truth of ALL views rendered by the CPU oracle
(brute-force tracer: rays x surfels), the HIP path trained on the training views only (fused L1+SSIM, sparse fused Adam), PSNR
(easyvolcap/utils/metric_utils.py:21-24: -10 log10 mse) on training and held-out views every few hundred steps, fp32 and fp16 feature storage.
   python scratch/psnr_run.py [--res 256] [--base 30000] [--env 10000] [--views 16] [--steps 3000]      ->  profiles/r05_psnr_run.txt"""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from envgs_amd import envgs_step, synth
from envgs_amd.loss import l1_ssim_loss
from envgs_amd.optim import FusedAdam
from tests import oracle_packages
from tests.test_train_convergence import _raw, _act, _psnr

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256); ap.add_argument("--base", type=int, default=30000); ap.add_argument("--env", type=int, default=10000)
ap.add_argument("--views", type=int, default=16); ap.add_argument("--steps", type=int, default=3000)
a = ap.parse_args()
import diff_surfel_rasterization_wet_ch05 as pkg, diff_surfel_tracing as tpkg
dev = torch.device("cuda:0")
H = W = a.res
NV = a.views
HELD = tuple(range(3, NV, 4)); TRAIN = tuple(v for v in range(NV) if v not in HELD)
gt_b = synth.base_gaussians(a.base, seed=3); gt_b["scales"] = gt_b["scales"] * (2.5 * (300000 / a.base) ** (1 / 3) / 2.15)
gt_b["opacities"] = torch.sigmoid(torch.randn(a.base, 1, generator=torch.Generator().manual_seed(1)) + 1.5)
gt_b["specular"] = torch.sigmoid(torch.randn(a.base, 1, generator=torch.Generator().manual_seed(2)))
gt_e = synth.env_gaussians(a.env, seed=4, bound=12.0)
cams_c = [synth.orbit_camera(v, n_views=NV, H=H, W=W, fx=1111.1 * W / 800.0) for v in range(NV)]
cams = [synth.orbit_camera(v, n_views=NV, H=H, W=W, fx=1111.1 * W / 800.0, device=dev) for v in range(NV)]
rays = [synth.get_rays(c) for c in cams]
bg_c, env_bg_c, deg_c = torch.zeros(3), torch.tensor([0.1, 0.2, 0.3]), torch.tensor([2])
bg, env_bg, deg = bg_c.to(dev), env_bg_c.to(dev), deg_c.to(dev)
print("# scene: %d base + %d env surfels, %dx%d, %d views (%d trained, %d held out: %s), SH degree 2; ground truth = CPU oracle (%d threads)" % (
    a.base, a.env, H, W, NV, len(TRAIN), len(HELD), list(HELD), torch.get_num_threads()), flush=True)
opkg, otpkg = oracle_packages.make_raster_pkg(5), oracle_packages.make_trace_pkg()
envgs_step.FUSED["on"] = False
raw_b, raw_e = _raw(gt_b, "cpu"), _raw(gt_e, "cpu")
t0 = time.time()
with torch.no_grad():
    target = [envgs_step.envgs_forward(opkg, otpkg, otpkg.SurfelTracer(), cams_c[v], synth.get_rays(cams_c[v]), _act(raw_b), _act(raw_e), bg_c, env_bg_c, deg_c)["rgb"].to(dev) for v in range(NV)]
print("# oracle ground truth: %.1f s for %d views" % (time.time() - t0, NV), flush=True)
envgs_step.FUSED["on"] = True
for storage in ("f32", "f16"):
    envgs_step.FEATURE_F16["on"] = storage == "f16"
    tracer = tpkg.SurfelTracer()
    render = lambda b, e, v: envgs_step.envgs_forward(pkg, tpkg, tracer, cams[v], rays[v], _act(b), _act(e), bg, env_bg, deg)["rgb"]
    gtb, gte = _raw(gt_b, dev), _raw(gt_e, dev)
    with torch.no_grad():
        fw = [_psnr(render(gtb, gte, v), target[v]) for v in range(NV)]
    print("%s: HIP render of the ground-truth parameters vs the oracle's images: %.1f dB worst view, %.1f dB mean" % (storage, min(fw), sum(fw) / NV), flush=True)
    g = torch.Generator().manual_seed(11)
    noise = lambda t, s: (torch.randn(t.shape, generator=g) * s).to(dev)
    base = {k: v.clone() for k, v in gtb.items()}; env = {k: v.clone() for k, v in gte.items()}
    base["shs"] += noise(base["shs"], 0.6); env["shs"] += noise(env["shs"], 0.6)
    base["opacities"] += noise(base["opacities"], 0.7); env["opacities"] += noise(env["opacities"], 0.7)
    base["specular"] += noise(base["specular"], 0.7)
    base["means3D"] += noise(base["means3D"], 0.002); env["means3D"] += noise(env["means3D"], 0.03)
    for d in (base, env):
        for t in d.values(): t.requires_grad_(True)
    lr = dict(means3D=1e-4, shs=1e-2, opacities=2e-2, scales=2e-3, rotations=1e-3, specular=2e-2, roughness=1e-2)
    groups = [{"params": [t], "lr": lr[k], "name": k} for k, t in base.items()] + [{"params": [t], "lr": lr[k] * (10 if k == "means3D" else 1), "name": "env_" + k} for k, t in env.items()]
    opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    def score(views):
        with torch.no_grad():
            return sum(_psnr(render(base, env, v), target[v]) for v in views) / len(views)
    t0 = time.time()
    for it in range(a.steps + 1):
        if it % max(1, a.steps // 6) == 0:
            torch.cuda.synchronize()
            print("%s step %5d: PSNR train %.2f dB, held-out %.2f dB   (%.1f s)" % (storage, it, score(TRAIN), score(HELD), time.time() - t0), flush=True)
        if it == a.steps: break
        v = TRAIN[it % len(TRAIN)]
        l1_ssim_loss(render(base, env, v).permute(2, 0, 1), target[v].permute(2, 0, 1)).backward()
        opt.step(); opt.zero_grad(set_to_none=True)
envgs_step.FUSED["on"] = False; envgs_step.FEATURE_F16["on"] = None
