"""Sparse fused Adam (include/envgs_optim.h) against the C restatement of easyvolcap/utils/src/fused_adam.cu:4-32 and against
torch.optim.Adam where no gradient is zero (the two coincide there)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import raster as orc


def oracle_adam(p, g, m, v, step, beta1, beta2, lr, eps):
    lib = orc.lib()
    f = lib.orc_fused_adam
    f.restype = None
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_float] * 5 + [ctypes.c_int64]
    p, m, v = (np.ascontiguousarray(a, np.float32).copy() for a in (p, m, v))
    g = np.ascontiguousarray(g, np.float32)
    f(p.ctypes.data, g.ctypes.data, m.ctypes.data, v.ctypes.data, step, beta1, beta2, lr, eps, p.size)
    return p, m, v


def test_oracle_adam_matches_torch_adam_when_dense():
    rng = np.random.default_rng(0)
    p0 = rng.standard_normal(1000).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-2, betas=(0.9, 0.999), eps=1e-15)
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = rng.standard_normal(1000).astype(np.float32) + 3.0     # never exactly zero
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m, v = oracle_adam(p, g, m, v, float(step), 0.9, 0.999, 1e-2, 1e-15)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-5, atol=1e-7)


def test_oracle_adam_skips_zero_gradients():
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal(64).astype(np.float32)
    m0 = rng.standard_normal(64).astype(np.float32); v0 = rng.random(64).astype(np.float32)
    g = rng.standard_normal(64).astype(np.float32); g[::3] = 0
    p, m, v = oracle_adam(p0, g, m0, v0, 7.0, 0.9, 0.999, 1e-2, 1e-15)
    assert np.array_equal(p[::3], p0[::3]) and np.array_equal(m[::3], m0[::3]) and np.array_equal(v[::3], v0[::3])
    assert (p[1::3] != p0[1::3]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[1], [5, 1024, 1027], [300000 * 3, 300000 * 48, 300000, 7], [3] * 30])
def test_fused_adam_matches_oracle(sizes):
    from envgs_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(len(sizes))
    ps = [torch.nn.Parameter(torch.randn(n, generator=gen).to(dev)) for n in sizes]
    lrs = [1e-2 / (i + 1) for i in range(len(sizes))]
    opt = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(ps, lrs)], lr=0.0, betas=(0.9, 0.999), eps=1e-15)
    ref = [(p.detach().cpu().numpy().copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)) for p, n in zip(ps, sizes)]
    for step in range(1, 4):
        for i, p in enumerate(ps):
            g = torch.randn(sizes[i], generator=gen)
            g[torch.rand(sizes[i], generator=gen) < 0.4] = 0
            if sizes[i] > 4096:
                g[1024:3072] = 0                                        # whole untouched quads / chunks
            p.grad = g.to(dev)
            ref[i] = oracle_adam(ref[i][0], g.numpy(), ref[i][1], ref[i][2], float(step), 0.9, 0.999, lrs[i], 1e-15)
        before = [(p.detach().clone(), {k: (v.clone() if torch.is_tensor(v) and v.is_cuda else v) for k, v in opt.state[p].items()}) for p in ps]
        opt.step()
        torch.cuda.synchronize()
        for i, p in enumerate(ps):
            st = opt.state[p]
            zero = ps[i].grad.cpu().numpy() == 0
            got = p.detach().cpu().numpy()
            np.testing.assert_allclose(got, ref[i][0], rtol=1e-5, atol=1e-7)                    # tolerance: powf / sqrt / div ulps
            np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), ref[i][1], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), ref[i][2], rtol=1e-6, atol=1e-12)
            zt = torch.from_numpy(zero).to(dev)                                                  # skipped elements: bit-exact, untouched
            assert torch.equal(p.detach()[zt], before[i][0][zt])
            if "exp_avg" in before[i][1]:
                assert torch.equal(st["exp_avg"][zt], before[i][1]["exp_avg"][zt]) and torch.equal(st["exp_avg_sq"][zt], before[i][1]["exp_avg_sq"][zt])
            else:
                assert not st["exp_avg"][zt].any() and not st["exp_avg_sq"][zt].any()


@pytest.mark.gpu
def test_fused_adam_state_is_torch_adam_compatible():
    from envgs_amd.optim import FusedAdam
    dev = torch.device("cuda", 0)
    p = torch.nn.Parameter(torch.randn(100, device=dev))
    q = torch.nn.Parameter(p.detach().clone())
    a, b = FusedAdam([p], lr=1e-3, eps=1e-15), torch.optim.Adam([q], lr=1e-3, eps=1e-15)
    for _ in range(3):
        g = torch.randn(100, device=dev) + 5
        p.grad, q.grad = g.clone(), g.clone()
        a.step(); b.step()
    torch.testing.assert_close(p, q, rtol=2e-5, atol=1e-7)
    b.load_state_dict(a.state_dict())                                                           # same state layout
    c = torch.nn.Parameter(torch.randn(4))                                                      # CPU tensor: no fallback, loud error
    c.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        FusedAdam([c], lr=1e-3).step()
