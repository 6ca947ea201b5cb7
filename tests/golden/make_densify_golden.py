"""Generate tests/golden/densify_golden.pt by RUNNING the reference's GaussianModel densify / prune schedule on CPU (authoring container only;
/root/reference does not exist on the GPU box).  For each scenario the fixture holds the inputs (raw parameters, Adam moments after two
optimizer steps, densification statistics, thresholds, the RNG seed set right before the call) and what the reference left behind (parameters,
moments, statistics).  Data only.  Re-run:  python tests/golden/make_densify_golden.py"""
import json
import os
import sys
from unittest.mock import MagicMock

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_specular", "_roughness")
STATS = ("xyz_gradient_accum", "denom", "max_radii2D", "xyz_weight_accum")
PREFIX = "sampler.pcd."


def snapshot(m, opt):
    out = {"params": {k: getattr(m, k).detach().clone() for k in NAMES}, "stats": {k: getattr(m, k).detach().clone() for k in STATS}, "m": {}, "v": {}}
    for g in opt.param_groups:
        st = opt.state[g["params"][0]]
        out["m"][g["name"][len(PREFIX):]] = st["exp_avg"].clone()
        out["v"][g["name"][len(PREFIX):]] = st["exp_avg_sq"].clone()
    return out


def main():
    for mod in ("pdbr", "pdbr.utils", "ruamel", "ruamel.yaml", "plyfile", "diff_surfel_tracing"):
        sys.modules[mod] = MagicMock()
    sys.modules["ujson"] = json
    sys.path.insert(0, "/root/reference")
    from easyvolcap.utils import gaussian2d_utils as g2d

    def model(P, seed, max_gs, max_gs_threshold, spatial_scale):
        torch.manual_seed(seed)
        m = g2d.GaussianModel(xyz=torch.rand(P, 3) * 2 - 1, colors=torch.rand(P, 3), init_occ=0.1, init_scale=torch.log(torch.rand(P, 2) * 0.095 + 0.005), sh_degree=1,
                              init_sh_degree=1, render_reflection=True, xyz_lr_scheduler=None, max_gs=max_gs, max_gs_threshold=max_gs_threshold,
                              spatial_scale=spatial_scale)
        with torch.no_grad():
            for k in NAMES:
                if k not in ("_xyz", "_scaling"):
                    getattr(m, k).add_(torch.randn_like(getattr(m, k)))
        opt = torch.optim.Adam([{"params": [getattr(m, k)], "lr": 1e-3, "name": PREFIX + k} for k in NAMES], lr=0.0, eps=1e-15)
        for _ in range(2):
            for k in NAMES:
                getattr(m, k).grad = torch.randn_like(getattr(m, k))
            opt.step()
        with torch.no_grad():
            denom = torch.randint(0, 6, (P, 1)).float()
            m.denom.set_(denom)
            m.xyz_gradient_accum.set_(torch.rand(P, 1) * denom)
            m.max_radii2D.set_(torch.rand(P) * 50)
            m.xyz_weight_accum.set_(torch.rand(P, 1) * 3 * denom)
        return m, opt

    scenarios = {
        "all_branches": dict(P=200, seed=1, max_gs=120, max_gs_threshold=0.9, spatial_scale=1.0, rng=123,
                             args=dict(min_opacity=0.05, min_gradient=0.05, densify_grad_threshold=0.5, densify_size_threshold=0.03, split_screen_threshold=30.0,
                                       max_scene_threshold=0.04, max_screen_threshold=40.0, min_weight_threshold=0.3, prune_visibility=True, prune_large_gs=True)),
        "clone_split_prune": dict(P=120, seed=2, max_gs=10 ** 6, max_gs_threshold=1.0, spatial_scale=2.0, rng=7,
                                  args=dict(min_opacity=0.04, min_gradient=None, densify_grad_threshold=0.4, densify_size_threshold=0.02)),
        "nothing_selected": dict(P=40, seed=3, max_gs=10 ** 6, max_gs_threshold=1.0, spatial_scale=1.0, rng=9,
                                 args=dict(min_opacity=None, min_gradient=None, densify_grad_threshold=1e9, densify_size_threshold=0.02)),
    }
    out = {}
    for name, sc in scenarios.items():
        m, opt = model(sc["P"], sc["seed"], sc["max_gs"], sc["max_gs_threshold"], sc["spatial_scale"])
        before = snapshot(m, opt)
        torch.manual_seed(sc["rng"])
        m.densify_and_prune(optimizer=opt, prefix=PREFIX, **sc["args"])
        after = snapshot(m, opt)
        out[name] = {"config": {k: v for k, v in sc.items() if k != "args"}, "args": sc["args"], "before": before, "after": after}
        print(name, sc["P"], "->", after["params"]["_xyz"].shape[0])
    # the two resets
    m, opt = model(60, 5, 10 ** 6, 1.0, 1.0)
    before = snapshot(m, opt)
    m.reset_opacity(0.01, opt, PREFIX)
    m.reset_specular(0.001, opt, PREFIX)
    out["resets"] = {"before": before, "after": snapshot(m, opt)}
    torch.save(out, os.path.join(HERE, "densify_golden.pt"))


if __name__ == "__main__":
    main()
