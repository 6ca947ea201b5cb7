#!/usr/bin/env python3
"""Dump (inputs, outputs, gradients) of the REAL extensions at the drop-in boundary -- the fixture that closes "parity unpinned".

The kernel bodies of diff_surfel_rasterization_wet{,_ch05,_ch07} and diff_surfel_tracing are not in /root/reference (empty submodules,
.gitmodules:1-6; install line README.md:69-72), so every constant of the oracle's compositing / tracing arithmetic is this project's reading of
the 2DGS / EnvGS papers.  Whoever has the CUDA/OptiX build on an NVIDIA box runs THIS script there:

    python dump_reference_kernel_golden.py --out reference_kernel_golden.pt            # needs: torch + the four installed packages, nothing else

and copies the file to tests/golden/reference_kernel_golden.pt of this repository; tests/test_reference_kernel_golden.py then feeds the stored
INPUTS to the HIP path and compares with the stored outputs and gradients (radii bit-exact, everything else within 1e-4).  The script is
self-contained on purpose: it imports torch, the four installed extension packages and -- by file path, for the seeded scene generators only --
envgs_amd/synth.py (pure torch; copy that one file next to the script, or pass --synth).  It contains no reference source.

In this repository's own environment the "installed" packages are the HIP ones; `--allow-local` lets the script run over them, which is how
the GPU test checks the script and the loader against each other (schema, argument order, settings fields) before anyone needs it for real.

File schema (torch.save of a dict):
  meta    : {schema: 1, torch, device_name, packages: {name: module file}, seed}
  raster  : [ {name, package, settings: {12 fields of GaussianRasterizationSettings; tensors on CPU}, inputs: {means3D, shs | colors_precomp,
               opacities, scales, rotations}, upstream: {color (C,H,W), allmap (7,H,W)}, outputs: {color, radii, allmap, weight},
               grads: {means3D, means2D, shs | colors_precomp, opacities, scales, rotations}} ]
  tracer  : [ {name, settings: {14 fields of SurfelTracingSettings}, start_from_first, inputs: {ray_o, ray_d, means3D, shs | colors_precomp,
               others_precomp | None, opacities, scales, rotations}, upstream: {rgb, dpt, acc, norm, aux},
               outputs: {rgb, dpt, acc, norm, dist, aux, mid, wet}, grads: {ray_o, ray_d, means3D, grads3D, shs | colors_precomp, others_precomp,
               opacities, scales, rotations}} ]
"""
import argparse
import importlib
import importlib.util
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def load_synth(path):
    cand = [path] if path else [os.path.join(HERE, "synth.py"), os.path.join(REPO, "envgs_amd", "synth.py")]
    for c in cand:
        if c and os.path.exists(c):
            spec = importlib.util.spec_from_file_location("_envgs_synth", c)
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            return m
    raise SystemExit("synth.py not found (copy envgs_amd/synth.py next to this script or pass --synth)")


def import_packages(allow_local):
    """The four extension packages as INSTALLED.  This repository's root carries same-named drop-in packages: unless --allow-local, the root is
    taken off sys.path first and a module that still resolves into the repository is refused."""
    names = ("diff_surfel_rasterization_wet", "diff_surfel_rasterization_wet_ch05", "diff_surfel_rasterization_wet_ch07", "diff_surfel_tracing")
    if not allow_local:
        sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != REPO]
    elif REPO not in sys.path:
        sys.path.insert(0, REPO)
    mods = {}
    for n in names:
        m = importlib.import_module(n)
        f = os.path.abspath(getattr(m, "__file__", "") or "")
        if not allow_local and f.startswith(REPO + os.sep):
            raise SystemExit("%s resolves to %s inside this repository: run where the real extension is installed (or pass --allow-local for a self-test)" % (n, f))
        mods[n] = m
    return mods


def cpu(t):
    return None if t is None else t.detach().to("cpu").clone()


def raster_cases(synth, seed):
    """Small seeded scenes in the three packages' forms: SH in-kernel (wet), 5 and 7 precomputed channels, a ragged image, long per-tile lists."""
    g = lambda s: torch.Generator().manual_seed(seed * 1000 + s)
    specs = [dict(name="wet_sh3_64x80", package="diff_surfel_rasterization_wet", P=400, H=64, W=80, C=3, sh=True, deg=3, s=0, bg=[0.2, 0.5, 0.9], mul=4.0),
             dict(name="wet_sh1_ragged_70x90", package="diff_surfel_rasterization_wet", P=600, H=70, W=90, C=3, sh=True, deg=1, s=1, bg=[0.0, 0.0, 0.0], mul=4.0),
             dict(name="ch05_64x64_bg3", package="diff_surfel_rasterization_wet_ch05", P=500, H=64, W=64, C=5, sh=False, deg=0, s=2, bg=[1.0, 1.0, 1.0], mul=4.0),
             dict(name="ch07_48x100", package="diff_surfel_rasterization_wet_ch07", P=500, H=48, W=100, C=7, sh=False, deg=0, s=3, bg=[0.0, 0.0, 0.0], mul=4.0),
             dict(name="wet_sh3_long_lists_128", package="diff_surfel_rasterization_wet", P=3000, H=128, W=128, C=3, sh=True, deg=3, s=4, bg=[1.0, 1.0, 1.0], mul=6.0)]
    for sp in specs:
        gs = synth.base_gaussians(sp["P"], seed=seed * 100 + sp["s"])
        gs["scales"] = gs["scales"] * sp["mul"]
        cam = synth.orbit_camera(1 + sp["s"], H=sp["H"], W=sp["W"], fx=1111.1 * sp["W"] / 800.0)
        if not sp["sh"]:
            gs["colors_precomp"] = torch.rand(sp["P"], sp["C"], generator=g(sp["s"]))
        up = dict(color=torch.randn(sp["C"], sp["H"], sp["W"], generator=g(10 + sp["s"])) / (sp["H"] * sp["W"]),
                  allmap=torch.randn(7, sp["H"], sp["W"], generator=g(20 + sp["s"])) / (sp["H"] * sp["W"]))
        up["allmap"][5] = 0.0                                  # the median-depth channel is a selection, not a sum: no gradient flows through it
        yield sp, gs, cam, up


def run_raster(mods, synth, dev, seed):
    out = []
    for sp, gs, cam, up in raster_cases(synth, seed):
        mod = mods[sp["package"]]
        st_cpu = dict(image_height=sp["H"], image_width=sp["W"], tanfovx=float(cam.tanfovx), tanfovy=float(cam.tanfovy), bg=torch.tensor(sp["bg"]),
                      scale_modifier=1.0, viewmatrix=cam.world_view_transform.clone(), projmatrix=cam.full_proj_transform.clone(),
                      sh_degree=torch.tensor([sp["deg"]]), campos=cam.camera_center.clone(), prefiltered=False, debug=False)
        st = mod.GaussianRasterizationSettings(**{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in st_cpu.items()})
        names = ("means3D", "shs" if sp["sh"] else "colors_precomp", "opacities", "scales", "rotations")
        leaves = {k: gs[k].to(dev).clone().requires_grad_(True) for k in names}
        m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
        color, radii, allmap, weight = mod.GaussianRasterizer(raster_settings=st)(
            means3D=leaves["means3D"], means2D=m2, shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"), opacities=leaves["opacities"],
            scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
        ((color * up["color"].to(dev)).sum() + (allmap * up["allmap"].to(dev)).sum()).backward()
        torch.cuda.synchronize()
        grads = {k: cpu(v.grad) for k, v in leaves.items()}
        grads["means2D"] = cpu(m2.grad)
        out.append(dict(name=sp["name"], package=sp["package"], settings=st_cpu, inputs={k: cpu(v) for k, v in leaves.items()},
                        upstream={k: v.clone() for k, v in up.items()},
                        outputs=dict(color=cpu(color), radii=cpu(radii), allmap=cpu(allmap), weight=cpu(weight)), grads=grads))
        print("raster %-28s ok: color %s, %d visible" % (sp["name"], tuple(color.shape), int((radii > 0).sum())))
    return out


def tracer_cases(synth, seed):
    specs = [dict(name="camera_rays_sh3", P=600, deg=3, sh=True, others=True, sff=True, cam=True, HW=(20, 24), s=0),
             dict(name="reflected_like_rays_sh2", P=800, deg=2, sh=True, others=False, sff=False, cam=False, R=512, s=1),
             dict(name="precomputed_colours", P=500, deg=0, sh=False, others=True, sff=False, cam=False, R=384, s=2)]
    for sp in specs:
        g = torch.Generator().manual_seed(seed * 1000 + 500 + sp["s"])
        e = synth.base_gaussians(sp["P"], seed=seed * 100 + 50 + sp["s"])
        e["scales"] = e["scales"] * 6.0
        if not sp["sh"]:
            e["colors_precomp"] = torch.rand(sp["P"], 3, generator=g)
        e["others"] = torch.rand(sp["P"], 2, generator=g) if sp["others"] else None
        if sp["cam"]:
            H, W = sp["HW"]
            cam = synth.orbit_camera(2, H=H, W=W, fx=1111.1 * W / 800.0)
            ro, rd = synth.get_rays(cam)
        else:
            cam = synth.orbit_camera(3, H=8, W=8)
            R = sp["R"]
            ro = (torch.rand(1, R, 3, generator=g) * 2 - 1) * 1.2
            rd = torch.randn(1, R, 3, generator=g)
            rd = rd / rd.norm(dim=-1, keepdim=True) * (0.5 + torch.rand(1, R, 1, generator=g))          # NOT normalised (optix_utils.py:125-127)
        lead = tuple(ro.shape[:-1])
        up = {k: torch.randn(lead + (c,), generator=g) / max(1, ro[..., 0].numel()) for k, c in (("rgb", 3), ("dpt", 1), ("acc", 1), ("norm", 3), ("aux", 2))}
        yield sp, e, cam, ro.contiguous(), rd.contiguous(), up


def run_tracer(mods, synth, dev, seed):
    mod = mods["diff_surfel_tracing"]
    out = []
    for sp, e, cam, ro, rd, up in tracer_cases(synth, seed):
        st_cpu = dict(image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=float(cam.tanfovx), tanfovy=float(cam.tanfovy),
                      bg=torch.tensor([0.1, 0.2, 0.3]), scale_modifier=1.0, viewmatrix=cam.world_view_transform.contiguous().clone(),
                      projmatrix=cam.full_proj_transform.contiguous().clone(), sh_degree=torch.tensor([sp["deg"]]), campos=cam.camera_center.contiguous().clone(),
                      prefiltered=False, debug=False, max_trace_depth=0, specular_threshold=0.0)
        st = mod.SurfelTracingSettings(**{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in st_cpu.items()})
        names = ["means3D", "shs" if sp["sh"] else "colors_precomp", "opacities", "scales", "rotations"] + (["others"] if sp["others"] else [])
        leaves = {k: e[k].to(dev).clone().requires_grad_(True) for k in names}
        o = ro.to(dev).clone().requires_grad_(True); d = rd.to(dev).clone().requires_grad_(True)
        g3 = torch.zeros_like(leaves["means3D"], requires_grad=True)
        v, f = synth.get_disks(leaves["means3D"].detach(), leaves["scales"].detach(), leaves["rotations"].detach())
        tracer = mod.SurfelTracer()
        tracer.build_acceleration_structure(v.detach().clone(), f.detach().clone(), rebuild=True)
        outs = tracer(o, d, v, means3D=leaves["means3D"], grads3D=g3, shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"),
                      others_precomp=leaves.get("others"), opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                      cov3D_precomp=None, tracer_settings=st, start_from_first=sp["sff"])
        rgb, dpt, acc, norm, dist, aux, mid, wet = outs
        loss = sum((x * up[k].to(dev)).sum() for k, x in (("rgb", rgb), ("dpt", dpt), ("acc", acc), ("norm", norm), ("aux", aux)))
        loss.backward()
        torch.cuda.synchronize()
        grads = {("others_precomp" if k == "others" else k): cpu(t.grad) for k, t in leaves.items()}
        grads.update(ray_o=cpu(o.grad), ray_d=cpu(d.grad), grads3D=cpu(g3.grad))
        inputs = {("others_precomp" if k == "others" else k): cpu(t) for k, t in leaves.items()}
        inputs.update(ray_o=ro.clone(), ray_d=rd.clone())
        if not sp["others"]:
            inputs["others_precomp"] = None
        out.append(dict(name=sp["name"], settings=st_cpu, start_from_first=sp["sff"], inputs=inputs, upstream={k: t.clone() for k, t in up.items()},
                        outputs=dict(rgb=cpu(rgb), dpt=cpu(dpt), acc=cpu(acc), norm=cpu(norm), dist=cpu(dist), aux=cpu(aux), mid=cpu(mid), wet=cpu(wet)), grads=grads))
        print("tracer %-28s ok: %d rays, mean acc %.3f" % (sp["name"], rgb[..., 0].numel(), float(acc.mean())))
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=os.path.join(HERE, "reference_kernel_golden.pt"))
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--synth", default=None, help="path of envgs_amd/synth.py (default: next to this script, then the repository's)")
    ap.add_argument("--allow-local", action="store_true", help="self-test: use this repository's own drop-in packages as the 'installed' ones")
    args = ap.parse_args()
    synth = load_synth(args.synth)
    mods = import_packages(args.allow_local)
    dev = torch.device(args.device)
    data = dict(meta=dict(schema=1, torch=torch.__version__, device_name=torch.cuda.get_device_name(dev), seed=args.seed, local_packages=bool(args.allow_local),
                          packages={n: os.path.abspath(getattr(m, "__file__", "") or "") for n, m in mods.items()}),
                raster=run_raster(mods, synth, dev, args.seed), tracer=run_tracer(mods, synth, dev, args.seed))
    torch.save(data, args.out)
    print("wrote %s (%d raster cases, %d tracer cases, %.1f KB)" % (args.out, len(data["raster"]), len(data["tracer"]), os.path.getsize(args.out) / 1024.0))


if __name__ == "__main__":
    main()
