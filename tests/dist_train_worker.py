"""Data-parallel EnvGS training worker for the N > 1 tests (launched with `python -m torch.distributed.run ... tests/dist_train_worker.py`):
one rank per GPU over RCCL (`nccl`) on a multi-GPU node, or several ranks sharing one GPU over gloo (ENVGS_DIST_BACKEND=gloo) on the 1-GPU
test box -- the same code either way.  It is the reference's multi-GPU entry (easyvolcap/scripts/main.py:240-275: one process per device,
every rank a full replica) restated over this project's pieces:

    camera sharding (envgs_amd.dist.shard order of bench.py)  ->  envgs_step.envgs_forward (the two drop-in extensions + fused glue)
    ->  backward with GradExchange (persistent flat buffers, direct reduce-scatter + all-gather launched from hooks)  ->  FusedAdam
    ->  densification statistics summed over ranks, then SurfelSet.densify_and_prune with identically seeded split offsets (the base set
        grows, the env set is pruned: both buckets are re-bound)

After `--steps` steps every rank compares ALL its parameters and Adam moments with rank 0's, bit for bit; rank 0 prints one JSON line."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _logit(p):
    return torch.log(p / (1 - p))


def raw_from(g, with_brdf):
    raw = {"_xyz": g["means3D"], "_features_dc": g["shs"][:, :1].contiguous(), "_features_rest": g["shs"][:, 1:].contiguous(),
           "_scaling": torch.log(g["scales"]), "_rotation": g["rotations"], "_opacity": _logit(g["opacities"].clamp(1e-4, 1 - 1e-4))}
    if with_brdf:
        raw["_specular"] = _logit(g["specular"].clamp(1e-4, 1 - 1e-4)); raw["_roughness"] = _logit(g["roughness"].clamp(1e-4, 1 - 1e-4))
    return raw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--densify-at", type=int, default=2)
    ap.add_argument("--gaussians", type=int, default=20000)
    ap.add_argument("--env-gaussians", type=int, default=8192)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--exchange", default="direct", choices=["direct", "allreduce"])
    args = ap.parse_args()

    import torch.distributed as dist
    from envgs_amd import dist as edist, synth, envgs_step, ckpt, densify
    from envgs_amd.optim import FusedAdam
    import diff_surfel_rasterization_wet_ch05 as pkg
    import diff_surfel_tracing as tpkg
    rank, world, local = edist.init_from_env()
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    H = W = args.res
    g = synth.base_gaussians(args.gaussians, seed=0, device=dev)
    g["scales"] = g["scales"] * 3.0
    ge = synth.env_gaussians(args.env_gaussians, seed=1, device=dev)
    cams = [synth.orbit_camera(v, n_views=8, H=H, W=W, fx=1111.1 * W / 800.0, device=dev) for v in range(8)]
    rays = [synth.get_rays(c) for c in cams]
    lr = {"_xyz": 1.6e-4, "_features_dc": 2.5e-3, "_features_rest": 1.25e-4, "_scaling": 5e-3, "_rotation": 1e-3, "_opacity": 5e-2, "_specular": 2.5e-3, "_roughness": 2.5e-3}
    base_raw = {k: torch.nn.Parameter(v.clone()) for k, v in raw_from(g, True).items()}
    env_raw = {k: torch.nn.Parameter(v.clone()) for k, v in raw_from(ge, False).items()}
    groups = [{"params": [v], "lr": lr[k], "name": "sampler.pcd." + k} for k, v in base_raw.items()] + \
             [{"params": [v], "lr": lr[k], "name": "sampler.env." + k} for k, v in env_raw.items()]
    opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    gen = torch.Generator(device=dev)
    base = densify.SurfelSet(base_raw, opt, "sampler.pcd.", spatial_scale=1.0, generator=gen)
    env = densify.SurfelSet(env_raw, opt, "sampler.env.", spatial_scale=1.0, generator=gen)
    ex = edist.GradExchange(lambda: [list(env.p.values()), list(base.p.values())], average=True, algo=args.exchange, overlap=True)
    tracer = tpkg.SurfelTracer()
    envgs_step.FUSED["on"] = True
    wgen = torch.Generator().manual_seed(5)
    dcol = (torch.randn(H, W, 3, generator=wgen) / (H * W)).to(dev)
    dall = (torch.randn(7, H, W, generator=wgen) / (H * W)).to(dev); dall[5:] = 0
    bg = torch.zeros(3, device=dev); env_bg = torch.zeros(3, device=dev)
    deg = torch.tensor([3], device=dev)
    log = []
    env_first = env.p["_features_dc"].detach()[:64].clone()
    for it in range(args.steps):
        if it == args.densify_at:
            # every rank takes the SAME decisions: statistics summed / maxed over the ranks, split offsets from identically seeded generators
            s = base.stats
            edist.allreduce_densify_stats(s["xyz_gradient_accum"], s["denom"], s["xyz_weight_accum"], s["max_radii2D"])
            gen.manual_seed(1234 + it)
            n0 = base.number
            thr = float(torch.quantile(base.gradient_avg()[base.stats["denom"][:, 0] > 0, 0], 0.7)) if bool((base.stats["denom"] > 0).any()) else 1.0
            base.densify_and_prune(min_opacity=0.02, min_gradient=None, densify_grad_threshold=thr, densify_size_threshold=0.02)
            op = env.opacity()[:, 0]
            env.remove(op < float(torch.quantile(op, 0.05)))
            log.append(dict(step=it, base_before=n0, base_after=base.number, env_after=env.number, events=[list(e) for e in base.log]))
        vi = (it * world + rank) % 8
        ex.begin_step()
        b_act, e_act = ckpt.activate(base.p), ckpt.activate(env.p)
        out = envgs_step.envgs_forward(pkg, tpkg, tracer, cams[vi], rays[vi], b_act, e_act, bg, env_bg, deg)
        loss = (out["rgb"] * dcol).sum() + (out["base"]["allmap"] * dall).sum()
        out["base"]["means2D"].retain_grad()
        loss.backward()
        ex.finish()
        opt.step()
        with torch.no_grad():
            radii = out["base"]["radii"]
            base.add_densification_stats(out["base"]["means2D"].grad, radii > 0, out["base"]["weight"], radii)
    torch.cuda.synchronize(dev)
    # bit-for-bit comparison with rank 0: parameters and both Adam moments of both sets
    same = True
    shapes = []
    for sset in (base, env):
        for k, gr, prm, st in sset._groups():
            for t in [prm.data] + ([st["exp_avg"], st["exp_avg_sq"]] if st is not None and "exp_avg" in st else []):
                n = torch.tensor([t.numel()], device=dev, dtype=torch.int64)
                dist.broadcast(n, 0)
                ref = t.detach().clone().contiguous().reshape(-1) if rank == 0 else torch.empty(int(n.item()), dtype=t.dtype, device=dev)
                dist.broadcast(ref, 0)                                       # (every rank takes part in every collective, whatever it holds)
                same = same and ref.numel() == t.numel() and bool(torch.equal(ref, t.detach().reshape(-1)))
            shapes.append(list(prm.shape))
    flag = torch.tensor([1 if same else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    finite = all(bool(torch.isfinite(p).all()) for sset in (base, env) for p in sset.p.values())
    moved = not torch.equal(env.p["_features_dc"].detach()[:64], env_first)          # (rows 0..63 survive the 5 % prune only by chance: compare loosely)
    if rank == 0:
        print(json.dumps(dict(identical_on_all_ranks=bool(int(flag.item())), finite=finite, world=world, backend=dist.get_backend(), steps=args.steps,
                              base=base.number, env=env.number, densify=log, exchange=args.exchange, trained=moved)))
    ex.remove()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
