#!/usr/bin/env python
"""bench.py -- one JSON line for the EnvGS render-and-trace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one training iteration's pass of the hot path over one camera view per GPU, exactly as the
reference's loop drives it (easyvolcap/runners/volumetric_video_runner.py:406-448): forward through the drop-in
`GaussianRasterizer` autograd.Function (and, for the `envgs` workload, the `SurfelTracer`), a synthetic upstream
gradient N(0,1)/HW on colour and allmap, `loss.backward()`; with N > 1 the 8-view batch is sharded over ranks and
the flat per-Gaussian gradient buffer is all-reduced once per step over RCCL/xGMI (weak scaling: one view per GPU).
Inputs are synthetic and resident in HBM before the timed region starts.

Besides the driver's contract fields the line carries
  roofline     : dominant kernel (most time per step): DEDUPLICATED algorithmic HBM bytes (what one launch must move at minimum: every
                 input structure once, every output once) / HIP-event launch time vs the 8 TB/s HBM peak -- every frac <= 1 and the sum over
                 the step stays below the peak; `traffic` = measured HBM bytes (PMC), `issue` = the instruction-issue roofline of the same
                 kernel (VALU / SALU wave-instructions per second against the issue peaks) from the tracked SQ counter summary in profiles/
  cpu_baseline : the CPU oracle (oracle/, OpenMP over the host cores) on the same scene + the PyTorch-eager config-1 figure, rank 0, N=1 only
  kernels      : per-kernel ms/launch and achieved GB/s from HIP events on the launch stream
  caller       : --caller fused (default: HIP glue) | twin (torch glue) | reference (the expression forms the unchanged EasyVolcap caller runs)
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # N > 1: the host driver only supports dmabuf IPC (RCCL's buffer exchange fails with the legacy mode)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(kernel, P, N, HW, C, sh_in_kernel):
    """BASELINE.md section 4 / SURVEY.md section 8(d): per-launch algorithmic HBM bytes (fp32)."""
    if kernel == "project_surfels":
        return P * (112 + (204 if sh_in_kernel else 4 * C))
    if kernel == "bin_tile_pairs":        # two walks over (centre, radius, depth) of every surfel, one 8 B pair written per tile instance
        return 2 * 16 * P + 8 * N
    if kernel == "sort_tile_lists":       # pair read, surfel id written (SURVEY's 164 B x N is the reference's device-wide radix sort: not done here)
        return (8 + 4) * N
    if kernel == "composite_fwd":
        return N * (64 + 4 * C) + HW * (48 + 4 * C) + 4 * P
    if kernel == "composite_bwd":
        return N * ((64 + 4 * C) + 8 * (15 + C)) + HW * (48 + 4 * C)
    if kernel == "project_surfels_bwd":
        return P * (4 * (15 + C) + 80 + (384 if sh_in_kernel else 0))
    return 0


def trace_algorithmic_bytes(kernel, tc, P_env, R, entries, others=False, rgb_only=True, colour_state=False):
    """Deduplicated algorithmic HBM bytes of the tracer kernels PER STEP: what the kernel must move at minimum -- each input structure
    read ONCE (BVH nodes 64 B + wide nodes 128 B + surfel record 64 B + SH block 192 B per env surfel, 24 B + 4 B order per ray), each
    list / state / record element read or written once -- however often the implementation re-fetches them from L2 / MALL.
    (SURVEY.md 8(d)'s per-RAY units -- 64 B per node visit of every ray -- counted a node once per ray although a 64-ray packet fetches it
    once, which put 'achieved' above the HBM peak; those units are kept only as the `per_ray_model_MB` diagnostic.)
    hits = composited hits, found = collected hits, entries = distinct (batch, surfel) pairs of the backward.
    rgb_only: the colour is the only traced output the loss uses (this bench, the EnvGS step): the backward then reads plane 0 of the per-hit
    state alone, 16 B per hit (batch_surfel_bwd<true>).  colour_state: the forward was told so (set_colour_only_backward) and WRITES plane 0 alone."""
    hits, found = tc["hits"], tc["found"]
    st = 16 if colour_state else (40 if others else 32)
    if kernel == "trace.collect_hits":
        return P_env * (64 + 128 + 64) + R * (24 + 4 + 4) + found * 8
    if kernel == "trace.sort_composite_fwd":
        return found * 8 + hits * (8 + st) + P_env * (64 + 192) + R * (24 + 4 + 4 + 4 * (3 + 1 + 1 + 3 + 1 + 2 + 16 + 1))
    if kernel == "trace.register_hits":
        return hits * (8 + 4) + entries * 8 + R * 4 + P_env * 8 * 16
    if kernel == "trace.batch_surfel_bwd":
        return hits * ((16 if (rgb_only and not others) else st) + 4) + entries * (8 + 256) + P_env * (64 + 192) + R * (24 + 48 + 48 + 24)
    if kernel == "trace.reduce_surfel_records":
        return entries * 256 + P_env * (32 + 192 + 64)
    if kernel == "bvh_build":
        return P_env * (48 + 24 + 6 * 16 + 64 + 64 + 128)
    return 0


def trace_per_ray_model_bytes(kernel, tc, R):
    """Round 1's per-RAY byte model (SURVEY.md 8(d) units x the kernel's counters), kept as a diagnostic only: it is NOT a roofline input."""
    hits, visits, found = tc["hits"], tc["node_visits"], tc["found"]
    if kernel == "trace.collect_hits":
        return visits * 64 + found * (64 + 8) + R * 24
    if kernel == "trace.sort_composite_fwd":
        return found * 16 + hits * (8 + 64 + 192 + 48) + R * (24 + 4 * (11 + 16 + 1))
    if kernel == "trace.batch_surfel_bwd":
        return hits * (8 + 48 + 64 + 192 + 8 * 63) + R * (24 + 4 * 12 + 4 * 12 + 24)
    return 0


# instruction-issue peaks (MI355X_MICROARCH.md): a 64-lane VALU instruction occupies a SIMD-32 for 2 cycles -> 4 SIMDs x 0.5 = 2 VALU
# wave-instructions per cycle per CU; one scalar unit per CU -> 1 SALU instruction per cycle per CU; 256 CUs x 2.4 GHz
VALU_PEAK_GINST = 256 * 2 * 2.4
VALU_MEASURED_GINST = 898.0     # what the chip sustains: independent v_fma_f32 / mixed VALU, 8 waves per SIMD (scratch/valu_peak.hip -> profiles/r02_valu_peak.txt)
SALU_PEAK_GINST = 256 * 1 * 2.4
PMC_TAG = "r06"
STEP_INVENTORY = os.path.join("profiles", "r06_step_inventory.txt")     # rocprofv3 kernel trace of this workload's step (scratch/step_inventory.py)


def workload_key(args, H, W):
    """Which tracked counter summary (profiles/<tag>_pmc_<key>.json, produced by profiles/collect_profiles.sh <key>) describes THIS run's workload --
    or None: the line then carries `issue` / `traffic` = null instead of another workload's counters."""
    base = args.gaussians == 300000 and args.caller == "fused" and args.feature_dtype == "f32"
    if args.workload == "raster":
        return "raster" if (args.gaussians == 300000 and H == 800 and W == 800) else None
    if args.workload == "base_trace":
        return ("base_trace_d%d" % args.trace_depth) if (args.gaussians == 300000 and H == 800 and W == 800 and args.trace_depth in (0, 2)) else None
    if H == 1200 and W == 1600 and args.trace_depth == 2 and args.channels == 7 and args.gaussians == 300000 and args.env_gaussians == 163840 and args.caller == "fused":
        return "config5"
    if not (base and H == 800 and W == 800 and args.trace_depth == 0 and args.channels == 5 and not args.no_colour_only_state):
        return None
    return {163840: "envgs", 700000: "env700k"}.get(args.env_gaussians)


def step_inventory():
    """Kernels per step and idle GPU time per step, from the committed kernel trace of the default EnvGS workload (a kernel trace cannot be
    taken from inside the timed run; the file names the command it came from)."""
    import re
    try:
        txt = open(os.path.join(ROOT, STEP_INVENTORY)).read()
    except OSError:
        return None
    m = re.search(r"step: (\d+) kernels", txt)
    per = [(int(k), float(i)) for i, k in re.findall(r"^step [0-9.]+ ms, idle ([0-9.]+) ms, (\d+) kernels", txt, flags=re.M)]
    if not m or not per:
        return None
    return {"kernels_per_step": int(m.group(1)), "kernels_per_step_sampled": [k for k, _ in per],
            "idle_gpu_ms_per_step": round(sum(i for _, i in per) / len(per), 3), "source": STEP_INVENTORY}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="envgs", choices=["raster", "envgs", "base_trace"],
                    help="envgs = BASELINE configs[2] (ch05 raster + env trace, the config the metric is quoted on); raster = configs[1]; base_trace = the reference's OTHER "
                         "tracer call (envgs_sampler.py:508-521 use_base_tracing, gaussian2d_sampler.py:391-449 use_optix_tracing): camera rays over the BASE set, "
                         "start_from_first=True, SH in-kernel, others_precomp = (specular, roughness), every traced output differentiated; --trace-depth 0 | 2")
    ap.add_argument("--specular-threshold", type=float, default=None, help="bounce threshold of --trace-depth > 0 (default: 0.5 for the envgs workload's random `others`, "
                    "0.0 = the reference's default for base_trace)")
    ap.add_argument("--gaussians", type=int, default=300000)
    ap.add_argument("--env-gaussians", type=int, default=163840)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--height", type=int, default=0, help="image height if not square (default: --res)")
    ap.add_argument("--width", type=int, default=0, help="image width if not square (default: --res)")
    ap.add_argument("--channels", type=int, default=5, choices=[5, 7], help="envgs workload: -ch05 (1 specular channel, EnvGS) or -ch07 (3, BASELINE configs[4])")
    ap.add_argument("--trace-depth", type=int, default=0, help="specular bounces of the env trace (EnvGS: 0; BASELINE configs[4]: 2); uses random 'others'")
    ap.add_argument("--caller", default="fused", choices=["fused", "twin", "reference"],
                    help="caller-side glue of the envgs workload: fused = envgs_amd.fused HIP kernels (SURVEY 8(f).1); twin = torch expressions "
                         "(envgs_amd/envgs_step.py, pinned against the reference's own render() / render_gaussians() by tests/test_caller_contract.py); "
                         "reference = twin + the expression forms the UNCHANGED EasyVolcap caller executes (batched-matmul get_disks, render()'s "
                         "regulariser maps) -- the step a drop-in user pays")
    ap.add_argument("--torch-glue", action="store_true", help="alias of --caller twin")
    ap.add_argument("--feature-dtype", default="f32", choices=["f32", "f16"],
                    help="storage of the per-surfel feature arrays the extensions read (base colours / SH, env SH): f16 = BASELINE configs[4]'s storage variant "
                         "(fp32 master parameters in the optimizer, a half copy per step for the render path; arithmetic and gradients stay fp32)")
    ap.add_argument("--prefer-rocblas", action="store_true", help="opt in to envgs_amd.prefer_rocblas() for the whole run (torch's tiny-K batched matmuls of the unchanged caller's get_disks, INTEGRATION.md section 5); the default run leaves torch's BLAS choice alone and times the reference-caller form under BOTH settings")
    ap.add_argument("--debug-trace", type=int, default=0, help="ENVGS_DBG_TRACE diagnostic switch mask (include/envgs_raster.h); reported in the JSON line")
    ap.add_argument("--diag", action="store_true", help="load the diagnostic build (A/B kernels of --debug-trace 8 / 16 / 512 / 2048; the product library rejects those switches)")
    ap.add_argument("--debug-raykey", type=int, default=0, help="ENVGS_DBG_RAYKEY diagnostic switch: value - 1 = direction-only rounds of the ray coherence key (csrc/ray_key.h); 0 = default")
    ap.add_argument("--debug-collect-wgs", type=int, default=0, help="ENVGS_DBG_COLLECT_WGS diagnostic switch: workgroups per CU of the collection's persistent grid (default 8)")
    ap.add_argument("--debug-segments", type=int, default=0, help="ENVGS_DBG_SEGMENTS diagnostic switch; reported in the JSON line")
    ap.add_argument("--optim", default="fused", choices=["fused", "torch", "none"],
                    help="optimizer step inside the timed iteration: sparse fused Adam (envgs_amd.optim, SURVEY 8(f).2), torch.optim.Adam, or none")
    ap.add_argument("--no-overlap-allreduce", action="store_true", help="N > 1: exchange the gradient buckets after backward() instead of launching them from backward hooks")
    ap.add_argument("--exchange", default="auto", choices=["auto", "direct", "allreduce"],
                    help="N > 1: direct reduce-scatter + all-gather over the xGMI mesh (all_to_all + local sum + all_gather), or one all_reduce per bucket")
    ap.add_argument("--no-render", action="store_true", help="skip the forward-only render timing (profiling runs: keeps the kernel statistics to the training steps)")
    ap.add_argument("--bvh-rebuild-every", type=int, default=16, help="the tracer's build-or-refit policy (every caller form): a full LBVH build at least on every K-th request, refits "
                    "(same topology, new boxes; exact) on the others while the tree's measured surface-area cost has not grown; 1 = full build on every request (the literal OptiX behaviour)")
    ap.add_argument("--no-colour-only-state", action="store_true", help="fused caller: let the env trace keep the full per-hit state (as for a caller that may differentiate "
                    "its depth / accumulation / normal outputs) instead of the colour's plane only (SurfelTracer.set_colour_only_backward)")
    ap.add_argument("--no-deferred-surfel-grads", action="store_true", help="fused caller, N = 1: keep the tracer backward's record sums on the step's stream (the reference's "
                    "semantics) instead of letting them run beside the base pass's backward (SurfelTracer.set_deferred_surfel_gradients; joined before the optimizer)")
    ap.add_argument("--no-prebuild", action="store_true", help="fused caller: build the environment structure inside the traced call (as the reference caller does) instead of ahead, under the base pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--live-scopes", default="two", choices=["two", "all", "none"], help="which of the library's HIP-event scopes record INSIDE the timed regions: the two kernels that took most of the warm-up "
                    "(default; the others are timed in a separate pass after the regions), all of them (0.17 ms per EnvGS step of barrier packets), or none (diagnostic: no per-kernel times)")
    ap.add_argument("--repeats", type=int, default=10, help="the timed region of EXACTLY --steps steps (barrier + synchronize on both sides) is run this many times "
                    "back to back; ms_per_step / value are the MEDIAN region, ms_per_step_spread has min / max (VERDICT r4: 20 steps = 0.17 s was a thin sample)")
    ap.add_argument("--no-reference-caller", action="store_true", help="skip the extra few steps that time the unchanged-EasyVolcap-caller form of the step (config.reference_caller_ms_per_step)")
    ap.add_argument("--step-times", type=int, default=0, help="diagnostics: after the timed region, run this many extra steps one by one (synchronised) and print their wall times and the allocator statistics to stderr")
    ap.add_argument("--stage-counts", action="store_true", help="diagnostics: after the timed region, one more step with every traced call's counters (rays, hits found / composited, entries) printed to stderr")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--cpu-rays", type=int, default=2048, help="rays of the same view traced by the brute-force CPU oracle (bounded sample)")
    args = ap.parse_args()

    if args.torch_glue and args.caller == "fused":
        args.caller = "twin"
    if args.prefer_rocblas:
        import envgs_amd as _ea
        _ea.prefer_rocblas()
    from envgs_amd import dist as edist, synth, raster, tracing, _lib
    rank, world, local = edist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dev = torch.device("cuda", local % torch.cuda.device_count())     # (a functional test may run 2 gloo ranks on one GPU)
    torch.cuda.set_device(dev)
    if args.diag:
        _lib.select("diag")                        # libenvgs_hip_diag.so: the product kernels + the superseded A/B kernels behind --debug-trace
    lib = _lib.load()
    lib.envgs_debug_set(0, args.debug_trace); lib.envgs_debug_set(1, args.debug_segments); lib.envgs_debug_set(2, args.debug_collect_wgs)
    lib.envgs_debug_set(4, args.debug_raykey)
    import torch.distributed as dist

    P, H, W = args.gaussians, (args.height or args.res), (args.width or args.res)
    HW = H * W
    envgs = args.workload == "envgs"
    btrace = args.workload == "base_trace"
    C = args.channels if envgs else 3
    g = synth.base_gaussians(P, seed=0, device=dev)
    cams = [synth.orbit_camera(v, n_views=8, H=H, W=W, fx=1111.1 * W / 800.0, device=dev) for v in range(8)]
    bg = torch.ones(3, device=dev) if not (envgs or btrace) else torch.zeros(3, device=dev)
    gen = torch.Generator().manual_seed(1)
    dcol = (torch.randn(C, H, W, generator=gen) / HW).to(dev)
    dall = (torch.randn(7, H, W, generator=gen) / HW).to(dev)
    dall[6] = 0                                   # distortion loss weight is 0 in the shipped configs (envgs.yaml:73)

    names = ["means3D", "shs", "opacities", "scales", "rotations"]
    params = {k: g[k].clone().requires_grad_(True) for k in names}
    env_params = {}
    ge = None
    if envgs:
        params["specular"] = g["specular"].repeat(1, C - 4).contiguous().clone().requires_grad_(True)      # (P,1) for -ch05, (P,3) for -ch07
        params["roughness"] = g["roughness"].clone().requires_grad_(True)
        ge = synth.env_gaussians(args.env_gaussians, seed=1, device=dev)
        env_params = {k: ge[k].clone().requires_grad_(True) for k in names}
        if args.trace_depth > 0:                   # bounces need a per-surfel specular value on the env set (others_precomp[:, 0]); half of the rays bounce
            from envgs_amd import envgs_step as _es
            _es.TRACE.update(depth=args.trace_depth, specular_threshold=(0.5 if args.specular_threshold is None else args.specular_threshold))
            env_others = torch.rand(args.env_gaussians, 2, generator=torch.Generator().manual_seed(3)).to(dev)

    if envgs:
        import importlib
        pkg = importlib.import_module("diff_surfel_rasterization_wet_ch0%d" % C)
        import diff_surfel_tracing as tpkg
        from envgs_amd import envgs_step
        mode = {"caller": args.caller}

        def set_caller(c):
            mode["caller"] = c
            envgs_step.FUSED["on"] = c == "fused"
            envgs_step.REFERENCE_FORMS["on"] = c == "reference"
            if c == "reference":
                from tests import reference_caller          # the unchanged caller's expression forms: measurement material, kept outside the package
                reference_caller.install()
        set_caller(args.caller)
        envgs_step.PREBUILD["on"] = not args.no_prebuild
        envgs_step.REFIT["every"] = max(1, args.bvh_rebuild_every)
        envgs_step.COLOUR_ONLY["on"] = not args.no_colour_only_state
        # N > 1 accumulates into the exchange's flat .grad views inside backward(): the surfel gradients must be complete on the step's stream there
        envgs_step.DEFER["on"] = world == 1 and not args.no_deferred_surfel_grads
        dnorm_hw = (torch.randn(3, H, W, generator=torch.Generator().manual_seed(2)) / HW).to(dev)
        tracer = tpkg.SurfelTracer()
        rays = [synth.get_rays(c) for c in cams]
        env_bg = torch.zeros(3, device=dev)
        dcol_hw3 = dcol[:3].permute(1, 2, 0).contiguous()
    elif btrace:
        import diff_surfel_tracing as tpkg
        from envgs_amd import fused as _fused
        params["specular"] = g["specular"].clone().requires_grad_(True)
        params["roughness"] = g["roughness"].clone().requires_grad_(True)
        tracer = tpkg.SurfelTracer()
        tracer.set_structure_policy("adaptive" if args.bvh_rebuild_every > 1 else "rebuild", max_age=max(1, args.bvh_rebuild_every - 1))
        # (each parameter's gradient comes from this call alone; joined after backward().  Bounce-free there is nothing for the tail to run beside: 6.62 against 6.55 ms)
        tracer.set_deferred_surfel_gradients(world == 1 and args.trace_depth > 0 and not args.no_deferred_surfel_grads)
        rays = [synth.get_rays(c) for c in cams]
        gen_t = torch.Generator().manual_seed(7)
        d_out = {k: (torch.randn(H, W, c_, generator=gen_t) / HW).to(dev) for k, c_ in (("rgb", 3), ("dpt", 1), ("acc", 1), ("norm", 3), ("aux", 2))}
        bt_thr = 0.0 if args.specular_threshold is None else args.specular_threshold
        pkg = None
    else:
        import diff_surfel_rasterization_wet as pkg
    sh_degree = torch.tensor([3], device=dev)

    env_in = dict(env_params)
    if envgs and args.trace_depth > 0:
        env_in["others"] = env_others
    half = args.feature_dtype == "f16"
    import envgs_amd
    envgs_amd.set_feature_storage("f16" if half else "f32")      # half copies inside the autograd nodes: fp32 parameters in, fp32 gradients out

    def base_trace_forward(cam, ray):
        """HardwareRendering.render_gaussians (optix_utils.py:87-201) as the base samplers call it: quads + rebuild request, SH in-kernel,
        others_precomp = (specular, roughness), start_from_first=True."""
        v, f = _fused.surfel_quads(params["means3D"], params["scales"], params["rotations"])
        tracer.build_acceleration_structure(v, f, rebuild=True)
        ts = tpkg.SurfelTracingSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
                                        viewmatrix=cam.world_view_transform.contiguous(), projmatrix=cam.full_proj_transform.contiguous(), sh_degree=sh_degree,
                                        campos=cam.camera_center.contiguous(), prefiltered=False, debug=False, max_trace_depth=int(args.trace_depth),
                                        specular_threshold=float(bt_thr))
        grads3D = torch.zeros_like(params["means3D"]).requires_grad_(True)
        others = torch.cat([params["specular"], params["roughness"]], dim=-1)
        return tracer(ray[0], ray[1], v, means3D=params["means3D"], grads3D=grads3D, shs=params["shs"], colors_precomp=None, others_precomp=others,
                      opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"], cov3D_precomp=None, tracer_settings=ts,
                      start_from_first=True)

    def settings(cam):
        return pkg.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree,
            campos=cam.camera_center, prefiltered=False, debug=False)

    n_acc = {"N": 0, "steps": 0}
    last_rays = [None, None]
    all_params = list(params.values()) + list(env_params.values())
    # 3DGS learning-rate ratios (configs/base/gaussian2d.yaml-style groups) scaled by 1e-2: same optimizer work per step, but the seeded
    # scene stays statistically the one the committed profiles describe over any --steps
    lr_of = {"means3D": 1.6e-6, "shs": 2.5e-5, "opacities": 5e-4, "scales": 5e-5, "rotations": 1e-5, "specular": 2.5e-5, "roughness": 2.5e-5}
    groups = [{"params": [v], "lr": lr_of[k], "name": k} for k, v in params.items()] + \
             [{"params": [v], "lr": lr_of[k], "name": "env_" + k} for k, v in env_params.items()]
    opt = None
    if args.optim == "fused":
        from envgs_amd.optim import FusedAdam
        opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    elif args.optim == "torch":
        opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    last_grads = []
    # N > 1: the gradient exchange starts inside backward() -- the environment bucket (final once the tracer's backward is done) is in
    # flight over xGMI while the base pass is still differentiating; --no-overlap-allreduce = one flat bucket after backward()
    reducer = None
    if world > 1:
        reducer = edist.GradExchange(lambda: [list(env_params.values()), list(params.values())], average=True, algo=args.exchange,
                                     overlap=not args.no_overlap_allreduce)

    def step(it):
        vi = (it * world + rank) % 8
        cam = cams[vi]
        if reducer is not None:
            reducer.begin_step()                   # .grad = zeroed views of the persistent flat buffers (the message itself)
        if envgs:
            out = envgs_step.envgs_forward(pkg, tpkg, tracer, cam, rays[vi], params, env_in, bg, env_bg, sh_degree)
            last_rays[0], last_rays[1] = out["ref_o"].detach(), out["ref_d"].detach()
            allmap = out["base"]["allmap"]
            loss = (out["rgb"] * dcol_hw3).sum() + (allmap * dall).sum()
            if mode["caller"] == "reference":     # the normal-consistency term consumes render()'s regulariser maps (volumetric_video_supervisor)
                loss = loss + (out["base"]["surf_normal"] * dnorm_hw).sum()
        elif btrace:
            rgb_, dpt_, acc_, norm_, dist_, aux_, mid_, wet_ = base_trace_forward(cam, rays[vi])
            loss = (rgb_ * d_out["rgb"]).sum() + (dpt_ * d_out["dpt"]).sum() + (acc_ * d_out["acc"]).sum() + (norm_ * d_out["norm"]).sum() + (aux_ * d_out["aux"]).sum()
        else:
            means2D = torch.zeros_like(params["means3D"], requires_grad=True)
            color, radii, allmap, weight = pkg.GaussianRasterizer(raster_settings=settings(cam))(
                means3D=params["means3D"], means2D=means2D, shs=params["shs"], colors_precomp=None,
                opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"], cov3D_precomp=None)
            loss = (color * dcol).sum() + (allmap * dall).sum()
        n_acc["N"] += (raster.LAST_STATS["N"] if not btrace else 0); n_acc["steps"] += 1
        loss.backward()
        tracing.join_deferred_gradients()         # (no-op unless the env surfels' gradients were left to finish beside the base pass's backward)
        nbytes = reducer.finish() if reducer is not None else 0
        if opt is not None:
            opt.step()
        last_grads[:] = [p_.grad for p_ in all_params]
        if reducer is None:
            for p_ in all_params:
                p_.grad = None
        return nbytes

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The library's HIP-event scopes are barrier packets on their streams: with every scope of an EnvGS step on, the step is 0.17 ms (2.3 %) longer than with
    # none (measured, --live-scopes all / none).  The warm-up runs with all of them and says which TWO kernels take most of the step (the collection and
    # the sort / composite pass swap places from run to run); those two stay on inside the timed regions -- the dominant kernel's launch time in
    # `roofline` is measured there -- and every other kernel is timed in a separate pass of a few steps right after them.
    NK = 0
    while lib.envgs_prof_kernel_name(NK): NK += 1
    KNAMES = [lib.envgs_prof_kernel_name(k).decode() for k in range(NK)]
    ALL_SCOPES = (1 << 64) - 1

    def drain_prof():
        for k in range(NK):
            t_, c_ = ctypes.c_double(0), ctypes.c_int(0)
            lib.envgs_prof_read(k, ctypes.byref(t_), ctypes.byref(c_))
    lib.envgs_prof_select(ALL_SCOPES)
    lib.envgs_prof_enable(1)
    drain_prof()
    for it in range(args.warmup):
        step(it)
    live = None
    if args.live_scopes == "two" and args.warmup > 0:
        tot = {}
        for k in range(NK):
            t_, c_ = ctypes.c_double(0), ctypes.c_int(0)
            lib.envgs_prof_read(k, ctypes.byref(t_), ctypes.byref(c_))
            if c_.value > 0 and KNAMES[k] not in ("trace_fwd", "trace_bwd", "bvh_build", "fused_adam_multi"):
                tot[k] = t_.value
        live = sorted(tot, key=lambda k: -tot[k])[:2]
    lib.envgs_prof_enable(0)
    exch_tune = None
    if reducer is not None and args.exchange == "auto":
        exch_tune = reducer.autotune()             # N > 1: both exchange forms measured on this machine between warm-up and the timed regions; the faster one is used
    n_acc.update(N=0, steps=0)
    lib.envgs_prof_select(sum(1 << k for k in live) if live else ALL_SCOPES)
    lib.envgs_prof_enable(0 if args.live_scopes == "none" else 1)
    drain_prof()                                  # anything recorded during warm-up
    ar_bytes = 0
    regions = []
    for rep in range(max(1, args.repeats)):       # every region: EXACTLY --steps steps between two (barrier + synchronize) pairs
        sync_all()
        t0 = time.perf_counter()
        for it in range(args.steps):
            ar_bytes = step(args.warmup + rep * args.steps + it)
        sync_all()
        regions.append(time.perf_counter() - t0)
    lib.envgs_prof_enable(0)
    n_timed = dict(n_acc)                          # (the kernel timers cover the timed steps only; later steps -- the reference-caller form, --step-times -- must not dilute the per-launch figures)
    if world > 1:
        tt = torch.tensor(regions, device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)  # per region: the slowest rank
        regions = [float(x) for x in tt.tolist()]
    elapsed = sorted(regions)[len(regions) // 2] if len(regions) % 2 else sorted(regions)[len(regions) // 2 - 1]      # the median region (lower middle of an even count)
    steps_done = args.steps * len(regions)

    EXT_KERNELS = ("project_surfels", "scan_tiles_touched", "bin_tile_pairs", "sort_tile_lists", "composite_fwd", "composite_bwd", "project_surfels_bwd",
                   "bvh_build", "trace_fwd", "trace_bwd")     # the two extensions' own launches (trace_fwd / trace_bwd are whole-call scopes: their inner kernels are not added again)

    def read_prof(nsteps):
        """Drain the library's HIP-event timers: {name: (ms per launch, launches)} and the extensions' own share of a step."""
        out = {}
        for k in range(NK):
            t_, c_ = ctypes.c_double(0), ctypes.c_int(0)
            lib.envgs_prof_read(k, ctypes.byref(t_), ctypes.byref(c_))
            if c_.value > 0:
                out[lib.envgs_prof_kernel_name(k).decode()] = (t_.value / c_.value, c_.value, t_.value)
        ext = sum(v[2] for n_, v in out.items() if n_ in EXT_KERNELS) / max(nsteps, 1)
        return out, ext
    timed_prof, ext_ms = read_prof(steps_done)
    kernel_pass_steps = 0
    if live is not None:
        # the pass for the kernels whose scopes were off: same steps, every scope on, outside the timed regions.  Launch counts are scaled to the timed step
        # count so that the per-step arithmetic below holds for both sources; the two live kernels keep their in-region figures
        kernel_pass_steps = max(4, min(args.steps, 10))
        lib.envgs_prof_select(ALL_SCOPES)
        lib.envgs_prof_enable(1)
        drain_prof()
        for it in range(kernel_pass_steps):
            step(args.warmup + steps_done + it)
        sync_all()
        lib.envgs_prof_enable(0)
        pass_prof, ext_ms = read_prof(kernel_pass_steps)
        scale = steps_done / kernel_pass_steps
        merged = {n_: (v[0], max(1, int(round(v[1] * scale))), v[2] * scale) for n_, v in pass_prof.items()}
        merged.update(timed_prof)
        timed_prof = merged
    lib.envgs_prof_select(ALL_SCOPES)

    # The drop-in number next to the headline: the SAME step with the expression forms the UNCHANGED EasyVolcap caller executes (batched-matmul
    # get_disks, render()'s regulariser maps + normal term, torch SH / reflection / blend) around the same two extensions -- what a user who only
    # swaps the packages pays.  A few steps, outside the timed region, reported in config.reference_caller_ms_per_step.
    ref_caller_ms = None
    ref_caller_by_blas = None
    ref_ext_ms = None
    if envgs and args.caller != "reference" and not args.no_reference_caller:
        set_caller("reference")
        ref_ext = []

        def time_reference_form(base_it):
            for it in range(3):
                step(base_it + it)
            sync_all()
            lib.envgs_prof_enable(1)
            tr_ = time.perf_counter()
            nref = max(4, min(args.steps, 10))
            for it in range(nref):
                step(base_it + 3 + it)
            sync_all()
            ms = (time.perf_counter() - tr_) / nref * 1e3
            lib.envgs_prof_enable(0)
            ref_ext.append(read_prof(nref)[1])
            if world > 1:
                tt = torch.tensor([ms], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ms = float(tt.item())
            return ms
        blas_name = lambda: str(torch.backends.cuda.preferred_blas_library()).split(".")[-1].lower()
        # importing the packages no longer touches torch's BLAS choice: the number a user who ONLY swaps the packages
        # sees is the one under torch's own setting; the other one is what the one-line pin of INTEGRATION.md section 5 buys
        ref_caller_by_blas = {}
        first = blas_name()
        ref_caller_ms = time_reference_form(args.warmup + args.steps)
        ref_caller_by_blas[first] = round(ref_caller_ms, 3)
        try:
            other = "cublas" if "lt" in first or first == "default" else "cublaslt"
            before = torch.backends.cuda.preferred_blas_library()
            torch.backends.cuda.preferred_blas_library(other)
            ref_caller_by_blas[blas_name()] = round(time_reference_form(args.warmup + args.steps + 40), 3)
            torch.backends.cuda.preferred_blas_library(before)
        except Exception as e:                                       # a torch build without the switch
            ref_caller_by_blas["error"] = str(e)[:80]
        ref_ext_ms = ref_ext[0] if ref_ext else None
        set_caller(args.caller)
        step(args.warmup + args.steps + 20)          # (back on the measured caller for the diagnostics below)
        sync_all()

    # N > 1: the gradient exchange on its own, both forms, on the step's own flat buffers (after the timed region): the first hardware run with
    # N > 1 ranks decides between the direct reduce-scatter + all-gather and the library's all_reduce from these fields
    exch = None
    if world > 1 and reducer is not None and reducer.buckets:
        exch = {"bytes_per_step": int(sum(B.flat.numel() * 4 for B in reducer.buckets)), "world": world}
        for algo in ("direct", "allreduce"):
            for _ in range(2):
                for B in reducer.buckets: edist.exchange_flat(B.flat, average=True, algo=algo, recv=B.recv, mine=B.mine)
            sync_all()
            te = time.perf_counter()
            nex = 5
            for _ in range(nex):
                for B in reducer.buckets: edist.exchange_flat(B.flat, average=True, algo=algo, recv=B.recv, mine=B.mine)
            sync_all()
            ms = (time.perf_counter() - te) / nex * 1e3
            tt = torch.tensor([ms], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ms = float(tt.item())
            S = exch["bytes_per_step"]
            # bytes that cross ONE link: direct = S/world out + S/world back on each of the world-1 links; a ring all_reduce pushes 2 (world-1)/world S through every link of the ring
            per_link = (2.0 * S / world) if algo == "direct" else (2.0 * (world - 1) / world * S)
            exch[algo] = {"ms": round(ms, 4), "algorithm_GBps": round(S / 1e9 / (ms / 1e3), 2), "per_link_GBps_model": round(per_link / 1e9 / (ms / 1e3), 2)}

    if args.step_times > 0 and rank == 0:
        ms0 = torch.cuda.memory_stats(dev)
        ts = []
        for it in range(args.step_times):
            torch.cuda.synchronize(dev); a_ = time.perf_counter()
            step(args.warmup + args.steps + it)
            torch.cuda.synchronize(dev); ts.append(round((time.perf_counter() - a_) * 1e3, 2))
        ms1 = torch.cuda.memory_stats(dev)
        keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")
        print("step times (ms): %s; allocator during them: %s; reserved %.1f GB, peak allocated %.1f GB" % (
            ts, {k: ms1.get(k, 0) - ms0.get(k, 0) for k in keys}, ms1.get("reserved_bytes.all.current", 0) / 2**30,
            ms1.get("allocated_bytes.all.peak", 0) / 2**30), file=sys.stderr)

    if args.stage_counts and (envgs or btrace) and rank == 0:
        # diagnostics: one more step with every traced call's counters read back (synchronising) -- rays, hits found / composited, entries per stage
        orig_tf = tracing.trace_forward
        seen = []

        def counting_tf(*a, **k):
            r_ = orig_tf(*a, **k)
            tc_ = tracing.last_trace_counts()
            seen.append(dict(rays=tc_["rays"], found=tc_["found"], hits=tc_["hits"], entries=sum(tracing.last_entry_counts()), cap=tc_["cap"], max_list=tc_["max_list"],
                             rows=tc_["compact_rows"], packet_nodes=tc_["packet_nodes"], packet_leaves=tc_["packet_leaves"], kbuffer_rays=tc_["rays_without_rows"]))
            return r_
        tracing.trace_forward = counting_tf
        try:
            for v_ in range(8):
                seen.append("view %d" % v_)
                step(8 * (args.warmup + args.steps + 100) + v_)
                torch.cuda.synchronize(dev)
        finally:
            tracing.trace_forward = orig_tf
        for c_ in seen:
            print(c_ if isinstance(c_, str) else "  traced call: %s" % json.dumps(c_), file=sys.stderr)

    # R7's LANE OCCUPANCY on this very view: the audit instantiation of the compositing kernel writes, per pixel, which entries of its
    # tile's list it blended; a (wavefront = 8x8 quadrant, splat) pass of composite_bwd runs when any of the quadrant's 64 pixels blended the splat, with
    # exactly those lanes live.  Outside the timed region, rank 0.
    lane_occ = None
    if rank == 0 and not btrace and H % 16 == 0 and W % 16 == 0:
        try:
            with torch.no_grad():
                st_ = (pkg.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cams[0].tanfovx, tanfovy=cams[0].tanfovy, bg=(bg if not envgs else torch.zeros(C, device=dev)), scale_modifier=1.0,
                                                         viewmatrix=cams[0].world_view_transform, projmatrix=cams[0].full_proj_transform, sh_degree=sh_degree,
                                                         campos=cams[0].camera_center, prefiltered=False, debug=False))
                cols_ = torch.zeros(P, C, device=dev)                                    # (the contributor sets do not depend on the colours)
                _, sv_ = raster.rasterize_forward(C, params["means3D"].detach(), None, cols_, params["opacities"].detach(), params["scales"].detach(),
                                                  params["rotations"].detach(), None, st_)
                rg_ = sv_["ranges"].view(-1, 2).long()
                lmax_ = int((rg_[:, 1] - rg_[:, 0]).max())
                if H * W * lmax_ <= (6 << 30):
                    contrib_, _, _ = raster.render_audit(sv_, lmax_)
                    q_ = contrib_.view(H // 8, 8, W // 8, 8, lmax_).permute(0, 2, 4, 1, 3).reshape(H // 8, W // 8, lmax_, 64)
                    live_ = q_.sum(-1, dtype=torch.int32)
                    passes_, lanes_ = int((live_ > 0).sum()), int(live_.sum())
                    lane_occ = {"passes": passes_, "live_lanes": lanes_, "lanes_per_pass": round(lanes_ / max(passes_, 1), 2), "frac": round(lanes_ / max(passes_, 1) / 64.0, 4),
                                "tile_instances": int(sv_["N"]), "note": "(8x8 quadrant, splat) passes of composite_bwd with a live lane and their live lanes on view 0 of THIS run, counted from "
                                "the contributor flags the AUDIT instantiation of the compositing kernel writes (envgs_raster_render_audit), outside the timed region"}
                    del contrib_, q_, live_
                del sv_
        except Exception as e_:
            lane_occ = {"error": repr(e_)[:200]}

    # per-kernel HIP-event times (this rank)
    N_avg = n_timed["N"] / max(n_timed["steps"], 1)
    traced = envgs or btrace
    P_trace = args.env_gaussians if envgs else P                 # the set the tracer walks
    tcounts = tracing.last_trace_counts() if traced else None    # (with --trace-depth > 0: the LAST stage's counters)
    entries = sum(tracing.last_entry_counts()) if traced else 0

    # acceleration structure on its own (outside the timed region): a full build (Morton keys, sort, hierarchy, fit) against a REFIT of the same
    # topology (envgs_bvh_refit: OptiX's "update", build_acceleration_structure(rebuild=False)) over the step's own environment set
    bvh_times = None
    if envgs and rank == 0:
        from envgs_amd import fused as _fz
        with torch.no_grad():
            vq, _ = _fz.surfel_quads(env_params["means3D"], env_params["scales"], env_params["rotations"])
            opq = env_params["opacities"].detach()
            nodes0, _ = tracing.build_bvh(vq, opq)
            def _ms(fn, reps=8):
                fn(); torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): fn()
                e1.record(); torch.cuda.synchronize(dev)
                return e0.elapsed_time(e1) / reps
            b_ms = _ms(lambda: tracing.build_bvh(vq, opq))
            r_ms = _ms(lambda: tracing.build_bvh(vq, opq, refit=nodes0))
        bvh_times = {"surfels": int(env_params["means3D"].shape[0]), "build_ms": round(b_ms, 4), "refit_ms": round(r_ms, 4), "refit_over_build": round(r_ms / max(b_ms, 1e-9), 3),
                     "note": "wall time of the launches on an otherwise idle GPU (incl. the temp allocation); per-kernel split in profiles/ (rocprofv3 kernel stats)"}
        del nodes0, vq

    # the metric's second half, "render Mpix/s": forward only under no_grad (inference: no per-hit state, no entries), outside the timed region
    def render(it):
        vi = (it * world + rank) % 8
        with torch.no_grad():
            if envgs:
                envgs_step.envgs_forward(pkg, tpkg, tracer, cams[vi], rays[vi], params, env_in, bg, env_bg, sh_degree)
            elif btrace:
                base_trace_forward(cams[vi], rays[vi])
            else:
                pkg.GaussianRasterizer(raster_settings=settings(cams[vi]))(
                    means3D=params["means3D"], means2D=torch.zeros_like(params["means3D"]), shs=params["shs"], colors_precomp=None,
                    opacities=params["opacities"], scales=params["scales"], rotations=params["rotations"], cov3D_precomp=None)
    n_render = 0 if args.no_render else max(2, min(args.steps, 10))
    render_s = float("nan")
    if n_render:
        render(0); render(1)
        sync_all()
        tr0 = time.perf_counter()
        for it in range(n_render):
            render(it)
        sync_all()
        render_s = (time.perf_counter() - tr0) / n_render
    if world > 1 and n_render:
        tt = torch.tensor([render_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        render_s = float(tt.item())
    kernels = {}
    for name, (ms, launches, _tot) in timed_prof.items():      # (the timers of the timed regions were drained right after them: timed_prof)
        if launches > 0:
            ab = algorithmic_bytes(name, P, N_avg, HW, C, not envgs) if not btrace else 0
            pr = 0
            if not ab and traced and tcounts:
                ab = trace_algorithmic_bytes(name, tcounts, P_trace, HW, entries, others=(args.trace_depth > 0 or btrace), rgb_only=not btrace,
                                             colour_state=(envgs and args.caller == "fused" and not args.no_colour_only_state and not args.trace_depth))
                pr = trace_per_ray_model_bytes(name, tcounts, HW)
            if name == "fused_adam_multi":        # 28 B per updated element (p,g,m,v in; p,m,v out), 4 B per skipped one (g only)
                nz = sum(int((g_ != 0).sum()) for g_ in last_grads if g_ is not None)
                tot = sum(g_.numel() for g_ in last_grads if g_ is not None)
                ab = 28 * nz + 4 * (tot - nz)
            per_step = max(1, round(launches / max(n_timed["steps"], 1)))      # the tracer forward runs its kernels once per batch segment
            ab = ab / per_step                                               # (2 segments on 2 streams, overlapping): bytes per LAUNCH
            kernels[name] = {"ms": round(ms, 4), "launches": launches, "alg_MB": round(ab / 1e6, 2),
                             "GBps": round(ab / 1e9 / (ms / 1e3), 1) if ab and ms > 0 else None}
            if pr:
                kernels[name]["per_ray_model_MB"] = round(pr / per_step / 1e6, 1)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = args.steps * world / elapsed
        spread = {"repeats": len(regions), "min": round(min(regions) / args.steps * 1e3, 4), "max": round(max(regions) / args.steps * 1e3, 4),
                  "timed_seconds_total": round(sum(regions), 3), "note": "ms_per_step / value = the median of `repeats` back-to-back timed regions of exactly `steps` steps each"}
        leaf = {k: v for k, v in kernels.items() if k not in ("trace_fwd", "trace_bwd") and v["GBps"]}
        dom = max(leaf, key=lambda k: leaf[k]["ms"] * leaf[k]["launches"]) if leaf else None      # most time per step (launches included)
        roof = None
        pm = {}
        wkey = workload_key(args, H, W)
        PMC_SUMMARY = os.path.join("profiles", "%s_pmc_%s.json" % (PMC_TAG, wkey)) if wkey else None
        try:                                   # tracked PMC summary of the SAME workload (profiles/collect_profiles.sh <key> -> profiles/summarize.py; separate --pmc passes)
            pm = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))["kernels"] if PMC_SUMMARY else {}
        except Exception:
            pm, PMC_SUMMARY = {}, None

        def issue_of(name):
            """Instruction-issue roofline of one kernel: wave-instructions per launch (SQ_INSTS_*, from the tracked counter summary) over the
            launch time measured in THIS run, against the issue peaks."""
            r = pm.get(name)
            if not r or "SQ_INSTS_VALU" not in r or name not in kernels:
                return None
            t = kernels[name]["ms"] / 1e3
            valu, salu = r["SQ_INSTS_VALU"], r.get("SQ_INSTS_SALU", 0.0)
            out = {"valu_insts_per_launch": int(valu), "salu_insts_per_launch": int(salu),
                   "valu_ginst_per_s": round(valu / t / 1e9, 1), "valu_peak_ginst_per_s": VALU_PEAK_GINST, "valu_util": round(valu / t / 1e9 / VALU_PEAK_GINST, 4),
                   "salu_ginst_per_s": round(salu / t / 1e9, 1), "salu_peak_ginst_per_s": SALU_PEAK_GINST, "salu_util": round(salu / t / 1e9 / SALU_PEAK_GINST, 4)}
            tot = sum(r.get(k_, 0.0) for k_ in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS"))
            out["all_insts_per_launch"] = int(tot)                       # every instruction type: what an issue-bound kernel pays for (profiles/r05_ab_collect.txt)
            out["all_ginst_per_s"] = round(tot / t / 1e9, 1)
            if r.get("SQ_WAVE_CYCLES"):
                out["wait_share_of_wave_cycles"] = round(r.get("SQ_WAIT_ANY", 0.0) / r["SQ_WAVE_CYCLES"], 3)
            out["valu_measured_peak_ginst_per_s"] = VALU_MEASURED_GINST
            out["valu_util_measured_peak"] = round(valu / t / 1e9 / VALU_MEASURED_GINST, 4)
            out["source"] = PMC_SUMMARY
            return out

        def calibrated(name):
            """Counter traffic by the calibrated reading of profiles/r03_fetch_calibration.txt: FETCH_SIZE counts read REQUESTS x 64 B, so for gather-dominated
            kernels (one request per lane access) the prescribed 2 x FETCH_SIZE over-states the reads; fetch_requests x 64 B + WRITE_SIZE is the lower reading."""
            r = pm.get(name) or {}
            if "fetch_requests" not in r:
                return None
            return int(r["fetch_requests"] * 64 + r.get("WRITE_SIZE_KB", 0.0) * 1024)

        if dom:
            A = kernels[dom]["GBps"]
            traffic = pm.get(dom, {}).get("hbm_bytes")
            roof = {"kernel": dom, "bound": "hbm", "achieved": A, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(A / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_calibrated": calibrated(dom),
                    "alg_bytes_per_launch": int(kernels[dom]["alg_MB"] * 1e6), "ms_per_launch": kernels[dom]["ms"],
                    "tile_instances_N": int(N_avg), "issue": issue_of(dom),
                    "step_total": {"alg_MB": round(sum(v["alg_MB"] * max(1, round(v["launches"] / max(n_timed["steps"], 1))) for v in leaf.values()), 1),
                                   "GBps_over_step": round(sum(v["alg_MB"] * max(1, round(v["launches"] / max(n_timed["steps"], 1))) for v in leaf.values()) / 1e3 / (ms_per_step / 1e3), 1)},
                    "note": "dominant kernel = most HIP-event time per step (duration x launches).  achieved = DEDUPLICATED algorithmic HBM bytes per launch (every "
                            "input structure once, every list / state / record element once: bench.py:trace_algorithmic_bytes; raster: SURVEY.md 8d formulas) / launch time; "
                            "traffic = (2*FETCH_SIZE + WRITE_SIZE) per launch from " + (PMC_SUMMARY or "(no tracked counter summary for this workload: traffic / issue are null)") + " (separate --pmc passes).  The tracer and compositing kernels are "
                            "instruction-issue bound, not HBM bound: `issue` carries their VALU / SALU issue rates against the issue peaks (2 VALU + 1 SALU "
                            "wave-instructions per cycle per CU, 256 CUs, 2.4 GHz)"}
            roof["kernel_timing"] = ({"in_the_timed_regions": [KNAMES[k] for k in live], "other_kernels": "a separate pass of %d steps with every scope on, right after the timed regions" % kernel_pass_steps,
                                      "why": "a HIP-event scope is a barrier packet on its stream: all ~30 scopes of an EnvGS step cost 0.17 ms per step (--live-scopes all / none); "
                                             "the two kernels that took most of the warm-up keep theirs inside the timed regions"}
                                     if live is not None else {"in_the_timed_regions": "every scope" if args.live_scopes != "none" else "none"})
            # the next kernels by time per step, the same way (the dominant one can change from round to round -- in round 3 the collection
            # overtook the sort / composite pass, which got 20 % faster while the collection gave up half of its wavefront slots to it)
            order = sorted(leaf, key=lambda k: -leaf[k]["ms"] * leaf[k]["launches"])
            roof["next_kernels"] = [{"kernel": k, "ms_per_launch": kernels[k]["ms"], "launches_per_step": max(1, round(kernels[k]["launches"] / max(n_timed["steps"], 1))),
                                     "achieved": kernels[k]["GBps"], "frac": round(kernels[k]["GBps"] / HBM_PEAK_GBS, 5), "traffic": pm.get(k, {}).get("hbm_bytes"), "traffic_calibrated": calibrated(k),
                                     "alg_bytes_per_launch": int(kernels[k]["alg_MB"] * 1e6),
                                     "valu_util_measured_peak": (issue_of(k) or {}).get("valu_util_measured_peak")} for k in order[1:4]]
            rb = kernels.get("composite_bwd")
            if rb and rb["GBps"]:
                roof["raster_composite_bwd"] = {"achieved": rb["GBps"], "frac": round(rb["GBps"] / HBM_PEAK_GBS, 5), "ms_per_launch": rb["ms"],
                                                "traffic": pm.get("composite_bwd", {}).get("hbm_bytes"), "issue": issue_of("composite_bwd"),
                                                "lane_occupancy": lane_occ}
                if envgs and world == 1 and args.caller == "fused" and not args.no_deferred_surfel_grads:
                    roof["raster_composite_bwd"]["shares_the_chip"] = ("with the tracer backward's deferred record sums (reduce_surfel_records on the library's stream, 512 workgroups): "
                                                                       "its launch time here includes that; alone (--no-deferred-surfel-grads) it takes ~0.06 ms less")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            if not btrace:
                cpu = cpu_baseline(g, cams[0], bg, dcol, dall, H, W, args.cpu_reps, C,
                                   (ge, last_rays, args.cpu_rays) if envgs else None)
            else:
                cpu = cpu_baseline_trace(g, rays[0], args.cpu_rays, H, W)
        line = {
            "metric": "train iters/s (fwd+bwd of the render hot path + Adam step, one %dx%d view per GPU per iter) + render Mpix/s" % (W, H),
            "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_step_spread": spread, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if not half else "f16-storage/f32-acc (feature arrays stored as half, fp32 arithmetic, accumulation and gradients)"), "data": "synthetic (seeded, BASELINE.md section 3; random-init Gaussians)",
            "config": {"workload": (("Ref-Real sedan-like full EnvGS (ch0%d raster + env LBVH trace)" % C) if envgs else
                                    ("camera rays traced over the base set (use_base_tracing / use_optix_tracing: start_from_first, SH in-kernel, others = specular + roughness, "
                                     "all five traced outputs differentiated), max_trace_depth %d, specular_threshold %g" % (args.trace_depth, bt_thr)) if btrace else
                                    "Ref-NeRF toaster-like base 2DGS raster only (BASELINE configs[1]), SH deg 3 in-kernel"),
                       "pmc_summary": PMC_SUMMARY,
                       "gaussians": P, "env_gaussians": (args.env_gaussians if envgs else 0), "resolution": [H, W], "trace_depth": (args.trace_depth if traced else None), "channels": C, "views": 8,
                       "parallelism": "dp%d (%d ranks seen by torch.distributed over %s, %d GPU(s) visible to this rank; camera batch sharded, %s)" % (world, (dist.get_world_size() if world > 1 else 1), ((dist.get_backend() + (" = RCCL" if dist.get_backend() == "nccl" else "")) if world > 1 else "no process group"), torch.cuda.device_count(), ("env / base flat grad buffers, %s, %s" % ((args.exchange if exch_tune is None else "auto -> %s (direct %.3f ms, allreduce %.3f ms per exchange of both buckets)" % (exch_tune["chosen"], exch_tune["direct"], exch_tune["allreduce"])), "launched from backward hooks" if not args.no_overlap_allreduce else "after backward")) if reducer is not None else "single GPU"),
                       "optimizer": {"fused": "sparse fused Adam, one launch (envgs_amd.optim)", "torch": "torch.optim.Adam", "none": "none"}[args.optim],
                       "caller_glue": ("n/a (one traced call per step; quads by envgs_amd.fused.surfel_quads)" if btrace else "n/a (raster only)" if not envgs else {"fused": "fused HIP (envgs_amd.fused)", "twin": "torch expressions (envgs_amd/envgs_step.py)",
                                       "reference": "the unchanged EasyVolcap caller's expression forms (batched-matmul get_disks, render()'s regulariser maps + normal term)"}[args.caller]),
                       "reference_caller_ms_per_step": (None if ref_caller_ms is None else round(ref_caller_ms, 3)),
                       "reference_caller_ms_by_torch_blas": ref_caller_by_blas,
                       "extension_ms_per_step": round(ext_ms, 3),
                       "reference_caller_extension_ms_per_step": (None if ref_ext_ms is None else round(ref_ext_ms, 3)),
                       "extension_note": "HIP-event time of the two extensions' own launches per step (raster stages, structure build / refit, tracer forward and backward) under the measured caller and under the reference-caller form: what is left of reference_caller_ms_per_step beyond it is torch glue of the caller (get_disks' batched matmuls, SH colours, reflection, regulariser maps, torch.optim-free here), not extension time",
                       "reference_caller_note": "the same step with the UNCHANGED EasyVolcap caller's expression forms around the same extensions (--caller reference), a few steps outside the timed region; reference_caller_ms_per_step is under torch's OWN BLAS choice (importing the packages changes nothing process-wide), reference_caller_ms_by_torch_blas has it under both (cublas = rocBLAS, cublaslt = hipBLASLt; the one-line pin of INTEGRATION.md section 5)",
                       "env_structure": (None if not envgs else ("LBVH rebuilt every step (tracer policy 'rebuild')" if args.bvh_rebuild_every <= 1 else
                                         "every call asks for a rebuild (as the reference's caller does); the tracer serves it with a refit while the tree is young: full LBVH build every %d calls or when the measured surface-area cost grew > 1.25x (SurfelTracer.set_structure_policy, all caller forms)" % args.bvh_rebuild_every)),
                       "env_per_hit_state": (None if not envgs else ("colour plane only: the caller promises a colour-only backward (16 B per hit; another gradient raises)"
                                             if (args.caller == "fused" and not args.no_colour_only_state and not args.trace_depth) else "all planes (32 B per hit, 40 with `others`)")),
                       "surfel_gradients": (None if not (envgs or btrace) else ("finished on the library's stream beside the base pass's backward, joined before the optimizer (SurfelTracer.set_deferred_surfel_gradients)"
                                                if (world == 1 and ((btrace and args.trace_depth > 0) or (envgs and args.caller == "fused")) and not args.no_deferred_surfel_grads) else "on the step's stream")),
                       "torch_blas": str(torch.backends.cuda.preferred_blas_library()).split(".")[-1],
                       "dist_backend": (dist.get_backend() if world > 1 else None),
                       "debug_switches": {"trace": args.debug_trace, "segments": args.debug_segments, "collect_wgs": args.debug_collect_wgs},
                       "allreduce_bytes_per_step": int(ar_bytes)},
            "train_mpix_per_s": round(value * HW / 1e6, 2),
            "render_mpix_per_s": (round(world * HW / render_s / 1e6, 2) if n_render else None),
            "render_ms_per_view": (round(render_s * 1e3, 4) if n_render else None),
            "exchange": exch,
            "launches": (step_inventory() if (envgs and args.caller == "fused" and H == 800 and W == 800 and not args.trace_depth and P == 300000 and args.env_gaussians == 163840 and args.bvh_rebuild_every == 16) else None),
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels, "bvh": bvh_times, "trace_counts": (dict(tcounts, entries=entries) if tcounts else None),
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(g, cam, bg, dcol, dall, H, W, reps, C, env=None):
    """The CPU oracle (C, OpenMP over all host cores) on the SAME scene and view: `reps` x (raster forward + backward);
    for the envgs workload plus the brute-force tracer oracle on a bounded sample of the same reflected rays, scaled to H*W."""
    try:
        from oracle import raster as orc
        import numpy as np
        a = {k: v.detach().cpu().numpy() for k, v in g.items()}
        view = cam.world_view_transform.cpu().numpy(); proj = cam.full_proj_transform.cpu().numpy()
        campos = cam.camera_center.cpu().numpy()
        dc, da = dcol.cpu().numpy(), dall.cpu().numpy()
        if C == 3:
            ckw = dict(shs=a["shs"], sh_degree=3)
        else:                                     # ch05 / ch07: colours precomputed (the python SH of the reference is not timed here)
            rng = np.random.default_rng(0)
            ckw = dict(colors_precomp=np.concatenate([rng.random((a["means3D"].shape[0], 3), dtype=np.float32), np.repeat(a["specular"], C - 4, axis=1), a["roughness"]], 1))
        t0 = time.perf_counter()
        for _ in range(reps):
            fwd = orc.raster_forward(a["means3D"], a["opacities"], view, proj, campos, W, H, scales=a["scales"],
                                     rotations=a["rotations"], bg=bg.cpu().numpy(), **ckw)
            orc.raster_backward(fwd, dc, da)
        dt = (time.perf_counter() - t0) / reps
        sample = "%d x (forward+backward) of the same %d-surfel %dx%d view through oracle/surfel_raster_oracle.c (%.2f s/iter)" % (
            reps, a["means3D"].shape[0], H, W, dt)
        if env is not None:
            from oracle import trace as otr
            ge, last_rays, nr = env
            e = {k: v.detach().cpu().numpy() for k, v in ge.items()}
            idx = np.linspace(0, H * W - 1, nr).astype(np.int64)
            ro = last_rays[0].reshape(-1, 3).cpu().numpy()[idx]; rd = last_rays[1].reshape(-1, 3).cpu().numpy()[idx]
            t1 = time.perf_counter()
            tf = otr.trace_forward(ro, rd, e["means3D"], e["scales"], e["rotations"], e["opacities"], shs=e["shs"], sh_degree=3,
                                   start_from_first=False)
            z = np.zeros
            otr.trace_backward(tf, np.ones((nr, 3), np.float32) / (H * W), z(nr, np.float32), z(nr, np.float32), z((nr, 3), np.float32), z((nr, 2), np.float32))
            dtt = (time.perf_counter() - t1) * (H * W / nr)
            sample += "; + brute-force tracer oracle (oracle/surfel_trace_oracle.c) fwd+bwd on %d of the %d reflected rays x %d env surfels, scaled to the full view (%.1f s/iter)" % (
                nr, H * W, e["means3D"].shape[0], dtt)
            dt += dtt
        out = {"value": round(1.0 / dt, 5), "unit": "iters/s", "cores": os.cpu_count(), "kind": "port",
               "sample": sample + "; OpenMP over all host cores", "raster_s_per_iter": round(dt - (dtt if env is not None else 0.0), 3)}
        if env is not None:
            out["tracer_s_per_iter"] = round(dtt, 1)
            out["tracer_note"] = "BRUTE FORCE: the tracer oracle tests every ray against every surfel (no acceleration structure), %.0f %% of this figure" % (100.0 * dtt / dt)
        out["eager_config1"] = eager_config1()
        if out["eager_config1"].get("value"):       # (also in `sample`, which every consumer of the line keeps)
            out["sample"] += "; PyTorch-eager config-1 (2 000 surfels, 256x256, CPU): " + out["eager_config1"]["sample"]
        return out
    except Exception as e:                       # the baseline is a reported figure, never a reason to lose the GPU number
        return {"value": None, "unit": "iters/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}


def cpu_baseline_trace(g, ray, nr, H, W):
    """base_trace workload: the brute-force tracer oracle (oracle/surfel_trace_oracle.c, OpenMP) forward + backward on a bounded sample of the same
    camera rays over the same base set, scaled to the full view."""
    try:
        import numpy as np
        from oracle import trace as otr
        a = {k: v.detach().cpu().numpy() for k, v in g.items()}
        idx = np.linspace(0, H * W - 1, nr).astype(np.int64)
        ro = ray[0].reshape(-1, 3).cpu().numpy()[idx]; rd = ray[1].reshape(-1, 3).cpu().numpy()[idx]
        oth = np.concatenate([a["specular"], a["roughness"]], 1).astype(np.float32)
        t1 = time.perf_counter()
        tf = otr.trace_forward(ro, rd, a["means3D"], a["scales"], a["rotations"], a["opacities"], shs=a["shs"], sh_degree=3, others=oth, start_from_first=True)
        on = lambda c: np.ones((nr, c), np.float32) / (H * W)
        otr.trace_backward(tf, on(3), on(1)[:, 0], on(1)[:, 0], on(3), on(2))
        dtt = (time.perf_counter() - t1) * (H * W / nr)
        return {"value": round(1.0 / dtt, 5), "unit": "iters/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "brute-force tracer oracle (oracle/surfel_trace_oracle.c: every ray x every surfel, no acceleration structure) fwd+bwd on %d of the %d camera rays x %d base "
                          "surfels, bounce-free, scaled to the full view (%.1f s/iter); OpenMP over all host cores" % (nr, H * W, a["means3D"].shape[0], dtt)}
    except Exception as e:
        return {"value": None, "unit": "iters/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}


def eager_config1(reps=10, warm=2, budget_s=75.0, threads=None):
    """BASELINE configs[0] / north_star: "PyTorch-eager CPU project + alpha-blend" -- 2 000 surfels, 256x256, forward only, oracle/eager.py
    (dense per-pixel blend in pixel chunks, fp32).  SURVEY.md 8(d): 2 warm-ups, then 10 repetitions; median and minimum reported; 16 torch
    threads (see below; `host_cores` says what the box has).  Bounded: the repetitions stop early (never below 3) once `budget_s` seconds are spent, and the count is
    reported -- the default bench.py run has to finish within minutes."""
    try:
        import statistics
        from envgs_amd import synth
        from oracle import eager
        P, H, W = 2000, 256, 256
        g = synth.base_gaussians(P, seed=0)
        g["scales"] = g["scales"] * 2.0
        cam = synth.orbit_camera(0, H=H, W=W, fx=1111.1 * W / 800.0)
        old = torch.get_num_threads()
        # SURVEY.md 8(d) says "all host cores"; on the GPU box's 256 cores the dense eager kernels of this size oversubscribe badly (measured,
        # scratch/eager_threads.py: 16 threads 8.0 s per render, 32: 9.2, 64: 16.3, 128: 34.8, 256: 142) -- the best setting is used and reported
        nthreads = threads or min(os.cpu_count() or 1, 16)
        torch.set_num_threads(nthreads)
        run = lambda: eager.rasterize(g["means3D"], g["opacities"], cam.world_view_transform, cam.full_proj_transform, cam.camera_center, W, H,
                                      scales=g["scales"], rotations=g["rotations"], shs=g["shs"], sh_degree=3, bg=torch.ones(3), pix_chunk=8192)
        ts = []
        t_all = time.perf_counter()
        with torch.no_grad():
            for _ in range(warm): run()
            for _ in range(reps):
                t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
                if len(ts) >= 3 and time.perf_counter() - t_all > budget_s:
                    break
        torch.set_num_threads(old)
        med, mn = statistics.median(ts), min(ts)
        return {"value": round(1.0 / med, 3), "unit": "renders/s", "best": round(1.0 / mn, 3), "median_s": round(med, 3), "min_s": round(mn, 3),
                "mpix_per_s": round(H * W / med / 1e6, 4), "threads": nthreads, "host_cores": os.cpu_count(), "warmups": warm, "reps": len(ts),
                "sample": "%d warm-ups + %d x forward of 2000 surfels at 256x256 (BASELINE configs[0]), oracle/eager.py, torch eager fp32, %d threads: median %.2f s, min %.2f s"
                          % (warm, len(ts), nthreads, med, mn)}
    except Exception as e:
        return {"value": None, "sample": "failed: %r" % (e,)}


if __name__ == "__main__":
    main()
