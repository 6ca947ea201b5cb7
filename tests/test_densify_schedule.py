"""Host logic of envgs_amd.densify.SurfelSet against the reference's own GaussianModel.densify_and_prune, run on CPU by
tests/golden/make_densify_golden.py: same inputs, same RNG seed -> the same surviving / cloned / split surfels, Adam moments and statistics.
On CPU the row gather is `torch_rows` (a test-only stand-in); the GPU test at the bottom runs the same schedule over the HIP compaction kernels
and requires the identical result."""
import os

import pytest
import torch

from envgs_amd import densify

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify_golden.pt")
PREFIX = "sampler.pcd."
NAMES = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_specular", "_roughness")


def _build(before, cfg, device, row_ops):
    params = {k: torch.nn.Parameter(before["params"][k].to(device).clone()) for k in NAMES}
    opt = torch.optim.Adam([{"params": [params[k]], "lr": 1e-3, "name": PREFIX + k} for k in NAMES], lr=0.0, eps=1e-15)
    for k in NAMES:
        opt.state[params[k]] = {"step": torch.tensor(2.0), "exp_avg": before["m"][k].to(device).clone(), "exp_avg_sq": before["v"][k].to(device).clone()}
    s = densify.SurfelSet(params, opt, PREFIX, spatial_scale=cfg.get("spatial_scale", 1.0), max_gs=cfg.get("max_gs"),
                          max_gs_threshold=cfg.get("max_gs_threshold", 1.0), row_ops=row_ops)
    for k in s.STATS:
        s.stats[k] = before["stats"][k].to(device).clone()
    return s, opt


def _state(s, opt):
    out = {"params": {k: s.p[k].detach().cpu() for k in NAMES}, "m": {}, "v": {}}
    for g in opt.param_groups:
        k = g["name"][len(PREFIX):]
        assert g["params"][0] is s.p[k]                                    # the optimizer trains the surfel set's current parameters
        st = opt.state[g["params"][0]]
        out["m"][k], out["v"][k] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
    assert len(opt.state) == len(NAMES)                                    # no stale entries of replaced parameters
    return out


def _same(got, exp, what):
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    assert torch.allclose(got, exp, rtol=1e-6, atol=1e-7, equal_nan=True), (what, float((got - exp).abs().max()))


@pytest.mark.parametrize("name", ["all_branches", "clone_split_prune", "nothing_selected"])
def test_schedule_matches_reference_run(name):
    sc = torch.load(GOLD, weights_only=True)[name]
    s, opt = _build(sc["before"], sc["config"], "cpu", densify.torch_rows)
    torch.manual_seed(sc["config"]["rng"])
    s.densify_and_prune(**sc["args"])
    got = _state(s, opt)
    for k in NAMES:
        _same(got["params"][k], sc["after"]["params"][k], k)
        _same(got["m"][k], sc["after"]["m"][k], "exp_avg " + k)
        _same(got["v"][k], sc["after"]["v"][k], "exp_avg_sq " + k)
    for k in s.STATS:
        _same(s.stats[k], sc["after"]["stats"][k], k)                      # reset by the schedule, at the new size
    if name == "all_branches":
        ev = dict(s.log)
        assert ev["clone"] > 0 and ev["split"] > 0 and ev["prune_occ_grad"] > 0 and ev["prune_large"] > 0 and ev["split_large"] > 0 and ev["prune_visibility"] > 0


def test_resets_match_reference_run():
    sc = torch.load(GOLD, weights_only=True)["resets"]
    s, opt = _build(sc["before"], {}, "cpu", densify.torch_rows)
    s.reset_opacity(0.01)
    s.reset_specular(0.001)
    got = _state(s, opt)
    for k in NAMES:
        _same(got["params"][k], sc["after"]["params"][k], k)
        _same(got["m"][k], sc["after"]["m"][k], "exp_avg " + k)
        _same(got["v"][k], sc["after"]["v"][k], "exp_avg_sq " + k)
    assert float(got["m"]["_opacity"].abs().sum()) == 0 and float(got["m"]["_xyz"].abs().sum()) > 0


def test_statistics_accumulate_like_the_reference():
    """add_densification_stats (gaussian2d_utils.py:901-909) + the radius update of gaussian2d_sampler.py:330-332."""
    raw = {k: v for k, v in torch.load(GOLD, weights_only=True)["resets"]["before"]["params"].items()}
    s = densify.SurfelSet(raw, None, row_ops=densify.torch_rows)
    P = s.number
    g = torch.Generator().manual_seed(0)
    grad = torch.randn(P, 3, generator=g); vis = torch.rand(P, generator=g) > 0.4; w = torch.rand(P, 1, generator=g); radii = torch.randint(0, 30, (P,), generator=g)
    s.add_densification_stats(grad, vis, w, radii)
    s.add_densification_stats(grad * 2, vis, w, radii // 2)
    assert torch.equal(s.stats["denom"][:, 0], vis.float() * 2)
    assert torch.allclose(s.stats["xyz_gradient_accum"][:, 0], grad.norm(dim=-1) * 3 * vis)
    assert torch.allclose(s.stats["xyz_weight_accum"], w * 2 * vis[:, None]) and torch.equal(s.stats["max_radii2D"], radii.float() * vis)
    avg = s.gradient_avg()
    assert torch.allclose(avg[vis, 0], grad.norm(dim=-1)[vis] * 1.5) and float(avg[~vis].abs().sum()) == 0          # 0/0 -> 0


def test_default_row_ops_refuse_cpu_tensors():
    raw = {k: v for k, v in torch.load(GOLD, weights_only=True)["resets"]["before"]["params"].items()}
    s = densify.SurfelSet(raw, None)
    with pytest.raises(RuntimeError):
        s.remove(torch.zeros(s.number, dtype=torch.bool))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["all_branches", "clone_split_prune"])
def test_schedule_over_hip_compaction_equals_torch_indexing(name):
    sc = torch.load(GOLD, weights_only=True)[name]
    res = []
    for ops in (None, densify.torch_rows):                                 # None = the HIP prune_rows
        s, opt = _build(sc["before"], sc["config"], "cuda:0", ops)
        torch.manual_seed(sc["config"]["rng"]); torch.cuda.manual_seed(sc["config"]["rng"])
        s.densify_and_prune(**sc["args"])
        res.append((_state(s, opt), {k: v.cpu() for k, v in s.stats.items()}, list(s.log)))
    (a, sa, la), (b, sb, lb) = res
    assert la == lb and a["params"]["_xyz"].shape[0] != sc["before"]["params"]["_xyz"].shape[0]
    for k in NAMES:
        assert torch.equal(a["params"][k], b["params"][k]) and torch.equal(a["m"][k], b["m"][k]) and torch.equal(a["v"][k], b["v"][k]), k
    for k in sa:
        assert torch.equal(sa[k], sb[k])
    # same decisions as the CPU reference run (the split offsets come from a different generator on the GPU, so only the counts are comparable)
    assert a["params"]["_xyz"].shape[0] == sc["after"]["params"]["_xyz"].shape[0] or name == "all_branches"


def test_split_draws_follow_the_generator():
    """Data parallelism: ranks that build their SurfelSet with identically seeded generators draw identical split offsets,
    whatever happened to the global RNG in between."""
    sc = torch.load(GOLD, weights_only=True)["clone_split_prune"]
    outs = []
    for junk in (0, 5):
        s, opt = _build(sc["before"], sc["config"], "cpu", densify.torch_rows)
        s.generator = torch.Generator().manual_seed(1234)
        torch.manual_seed(junk); torch.rand(junk + 1)                 # the global generator differs between the two "ranks"
        s.densify_and_prune(**sc["args"])
        outs.append(s.p["_xyz"].detach().clone())
    assert outs[0].shape == outs[1].shape and torch.equal(outs[0], outs[1])
