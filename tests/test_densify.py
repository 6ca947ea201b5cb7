"""Prune compaction and the 3-NN helper (include/envgs_densify.h) against their torch definitions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,frac", [(1, 1.0), (7, 0.5), (1000, 0.0), (300000, 0.8), (4097, 1.0)])
def test_prune_rows_is_boolean_indexing(P, frac):
    from envgs_amd import densify
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(P)
    keep = (torch.rand(P, generator=gen) < frac).to(dev)
    tensors = [torch.randn(P, 3, generator=gen), torch.randn(P, 16, 3, generator=gen), torch.randn(P, 1, generator=gen), torch.randn(P, generator=gen),
               torch.randint(0, 1000, (P, 2), generator=gen, dtype=torch.int32), torch.randn(P, 4, generator=gen)]
    tensors = [t.to(dev) for t in tensors]
    outs = densify.prune_rows(tensors, keep)
    for t, o in zip(tensors, outs):
        assert o.dtype == t.dtype and torch.equal(o, t[keep])                        # bit-exact, order kept


def test_prune_and_cat_optimizer_follow_the_reference_contract():
    from envgs_amd import densify
    from envgs_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    P = 5000
    names = {"_xyz": 3, "_features_dc": 3, "_opacity": 1, "_scaling": 2, "_rotation": 4}
    params = {k: torch.nn.Parameter(torch.randn(P, c, device=dev)) for k, c in names.items()}
    opt = FusedAdam([{"params": [v], "lr": 1e-3, "name": k} for k, v in params.items()], lr=0.0, eps=1e-15)
    for v in params.values():
        v.grad = torch.randn_like(v)
    opt.step()
    before = {k: (v.detach().clone(), opt.state[v]["exp_avg"].clone(), opt.state[v]["exp_avg_sq"].clone(), opt.state[v]["step"]) for k, v in params.items()}
    keep = torch.rand(P, device=dev) < 0.7
    new = densify.prune_optimizer(opt, keep)
    assert set(new) == set(names)
    for k, prm in new.items():
        assert isinstance(prm, torch.nn.Parameter) and prm.requires_grad and prm is opt.param_groups[list(names).index(k)]["params"][0]
        st = opt.state[prm]
        assert torch.equal(prm.detach(), before[k][0][keep]) and torch.equal(st["exp_avg"], before[k][1][keep]) and torch.equal(st["exp_avg_sq"], before[k][2][keep])
        assert st["step"] is before[k][3]
    assert len(opt.state) == len(names)                                               # the old parameters' state is gone
    n = int(keep.sum())
    ext = {k: torch.randn(11, c, device=dev) for k, c in names.items()}
    new2 = densify.cat_tensors_to_optimizer(ext, opt)
    for k, prm in new2.items():
        st = opt.state[prm]
        assert prm.shape[0] == n + 11 and torch.equal(prm.detach()[n:], ext[k]) and not st["exp_avg"][n:].any() and st["exp_avg"].shape == prm.shape
    for v in new2.values():                                                           # and the optimizer still steps
        v.grad = torch.randn_like(v)
    opt.step()


@pytest.mark.parametrize("P", [1, 2, 3, 4, 257, 5000])
def test_knn3_mean_dist2(P):
    from envgs_amd import densify
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(P)
    x = torch.randn(P, 3, generator=gen).to(dev)
    got = densify.knn3_mean_dist2(x)
    d = torch.cdist(x.double(), x.double()) ** 2
    d.fill_diagonal_(float("inf"))
    k = min(3, P - 1)
    ref = d.topk(k, dim=1, largest=False).values.mean(1) if k > 0 else torch.zeros(P, device=dev, dtype=torch.double)
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-7)              # tolerance: fp32 distance arithmetic
