"""GPU parity of the hand-written binning (csrc/raster_bin.hip: R2 scan, per-tile histograms, scatter, per-tile LDS sort) against the
oracle's restatement of the reference's pipeline (scan -> emit -> STABLE sort of (tile id << 32 | depth bits) keys -> ranges): the sorted
key buffer, the surfel list and the tile ranges are index work and must be bit-exact -- in every regime of the per-tile sort (lists that
fit one 32 KB LDS sort, lists for the 128 KB one, lists sorted in HBM), with more tiles than one LDS histogram band holds, with equal
depths (ties fall back to surfel-index order = the stable sort's emission order), and for the scan at sizes around its workgroup granularity."""
import numpy as np
import pytest
import torch

from tests.util import small_scene, cam_args

pytestmark = pytest.mark.gpu


def _run(g, cam, H, W):
    from envgs_amd import raster
    from oracle import raster as orc
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    bg = torch.zeros(3)
    st = mod.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=bg.to(dev), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
        sh_degree=torch.tensor([0], device=dev), campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
    gd = {k: v.to(dev) for k, v in g.items()}
    raster._N_GUESS.pop((dev.index, g["means3D"].shape[0], H, W), None)
    outs, saved = raster.rasterize_forward(3, gd["means3D"], None, gd["colors_precomp"], gd["opacities"], gd["scales"], gd["rotations"],
                                           None, st, keep_binning=True)
    torch.cuda.synchronize()
    ca = cam_args(cam)
    ref = orc.raster_forward(g["means3D"].numpy(), g["opacities"].numpy(), ca["viewmatrix"].numpy(), ca["projmatrix"].numpy(),
                             ca["campos"].numpy(), W, H, bg=bg.numpy(), scales=g["scales"].numpy(), rotations=g["rotations"].numpy(),
                             colors_precomp=g["colors_precomp"].numpy())
    return outs, saved, ref


def _assert_lists_equal(saved, ref):
    N = ref["N"]
    assert saved["N"] == N
    np.testing.assert_array_equal(saved["offsets"].cpu().numpy().view(np.uint32), ref["offsets"])
    np.testing.assert_array_equal(saved["ranges"].cpu().numpy().view(np.uint32), ref["ranges"])
    np.testing.assert_array_equal(saved["keys_sorted"].cpu().numpy().view(np.uint64)[:N], ref["keys_sorted"])
    np.testing.assert_array_equal(saved["point_list"].cpu().numpy().view(np.uint32)[:N], ref["point_list"])


@pytest.mark.parametrize("P, lo, hi", [(1500, 0, 4096), (6000, 4096, 8192), (12000, 8192, 16384), (30000, 16384, 32768), (150000, 131072, 1 << 30)])
def test_long_tile_lists(P, lo, hi):
    """Four tiles, every surfel large: lists of ~P/1.1 entries per tile -- one per-tile LDS sort (4096 / 8192 entries), the 16 384-entry
    LDS sort of the long-list kernel, and segments sorted chunk by chunk (2 chunks / 9 chunks, four merge stages wider than a chunk)."""
    H = W = 32
    g, cam = small_scene(P=P, H=H, W=W, seed=5, C=3, sh=False, scale_mul=40.0)
    outs, saved, ref = _run(g, cam, H, W)
    r = ref["ranges"].astype(np.int64)
    lmax = int((r[:, 1] - r[:, 0]).max())
    assert lo < lmax <= hi, lmax
    _assert_lists_equal(saved, ref)
    # and the image composited from those lists (the deepest lists the compositing kernel sees anywhere in the suite)
    d = np.abs(outs[0].cpu().numpy() - ref["out_color"])
    assert np.isfinite(d).all() and (d > 2e-4).mean() < 2e-2, (d.max(), (d > 2e-4).mean())      # (threshold flips are audited in test_raster_parity.py)


def test_a_few_long_lists_among_short_ones():
    """Most tiles short (the per-tile LDS array is sized from the AVERAGE list: 2048 entries here), four tiles with ~3500 entries: those
    are handed to the long-list kernel through the device-side work list."""
    H = W = 128
    g, cam = small_scene(P=5000, H=H, W=W, seed=8, C=3, sh=False, scale_mul=1.0)
    gen = torch.Generator().manual_seed(3)
    g["means3D"][:3500] = 0.02 * torch.randn(3500, 3, generator=gen)
    g["scales"][:3500] *= 0.3
    outs, saved, ref = _run(g, cam, H, W)
    r = ref["ranges"].astype(np.int64)
    ln = r[:, 1] - r[:, 0]
    assert ref["N"] * 8 // (5 * len(ln)) <= 2048 and int((ln > 2048).sum()) >= 2 and ln.max() <= 16384
    _assert_lists_equal(saved, ref)


def test_equal_depths_fall_back_to_surfel_order():
    """Surfels with identical centres share their view depth bit for bit: the reference's stable sort keeps them in emission order
    (= surfel-index order inside a tile); the per-tile sort orders by (depth bits, surfel id), which is the same list."""
    H = W = 64
    g, cam = small_scene(P=2000, H=H, W=W, seed=9, C=3, sh=False, scale_mul=6.0)
    g["means3D"] = g["means3D"][:500].repeat(4, 1)                        # four surfels (different scales / rotations) on every centre
    outs, saved, ref = _run(g, cam, H, W)
    ks = ref["keys_sorted"]
    assert int((ks[1:] == ks[:-1]).sum()) > 1000                          # many exact (tile, depth) ties
    _assert_lists_equal(saved, ref)


def test_more_tiles_than_one_histogram_band():
    """2064 x 2064 = 16 641 tiles: the LDS histogram covers 16 384 tiles, so every slice of surfels is walked for two bands of tiles."""
    H = W = 2064
    g, cam = small_scene(P=3000, H=H, W=W, seed=6, C=3, sh=False, scale_mul=4.0)
    outs, saved, ref = _run(g, cam, H, W)
    assert ref["ranges"].shape[0] == 129 * 129 > 16384
    r = ref["ranges"].astype(np.int64)
    assert (r[16384:, 1] - r[16384:, 0]).sum() > 0                         # the second band is not empty
    _assert_lists_equal(saved, ref)


def test_capacity_below_the_count_leaves_every_range_empty():
    """The kernels re-derive the instance count; a capacity below it must not write out of bounds and must leave nothing to composite
    (rasterize_forward then repeats the call: test_speculative_instance_count_is_exact_and_recovers_from_a_small_guess)."""
    from envgs_amd import raster, _lib
    import diff_surfel_rasterization_wet as mod
    dev = torch.device("cuda:0")
    H = W = 96
    g, cam = small_scene(P=3000, H=H, W=W, seed=2, C=3, sh=False)
    outs, saved, ref = _run(g, cam, H, W)
    N = saved["N"]
    lib = _lib.load()
    cap = N // 3
    guard = 4096
    pairs = torch.full((cap + guard,), -1, dtype=torch.int64, device=dev)
    plist = torch.full((cap + guard,), -1, dtype=torch.int32, device=dev)
    ranges = torch.full((36, 2), -1, dtype=torch.int32, device=dev)
    nb = lib.envgs_raster_sort_temp_bytes(cap, W, H)
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    out_color = torch.empty(3, H, W, **f32); allmap = torch.empty(7, H, W, **f32); final_T = torch.empty(3, H, W, **f32)
    ncon = torch.empty(2, H, W, dtype=torch.int32, device=dev); weight = torch.empty(3000, 1, **f32)
    p = _lib.ptr
    _lib.check(lib.envgs_raster_bin_and_render(saved["cfg"], cap, p(saved["geom"]), p(saved["radii"]), p(saved["colors"]), p(saved["bg"]),
                                               p(pairs), None, p(plist), p(temp), nb, p(ranges), p(out_color), p(allmap), p(final_T),
                                               p(ncon), p(weight), None, raster._stream(dev)), "bin_and_render")
    torch.cuda.synchronize()
    assert int(ranges.abs().sum()) == 0
    assert bool((pairs[cap:] == -1).all()) and bool((plist[cap:] == -1).all())
    assert float(allmap[1].abs().max()) == 0.0 and float(weight.abs().max()) == 0.0        # nothing composited


@pytest.mark.parametrize("P", [1, 3, 1023, 1024, 1025, 4099, 262144, 300001, 1800000])
def test_scan_around_its_workgroup_granularity(P):
    """R2's two-launch prefix sum (1024 counters per workgroup), in place, through envgs_compact_scan: positions of the kept rows."""
    from envgs_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(P)
    keep = (torch.rand(P, generator=gen) < 0.37)
    k8 = keep.to(torch.uint8).to(dev)
    pos = torch.empty(P, dtype=torch.int32, device=dev)
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    tb = lib.envgs_compact_temp_bytes(P)
    temp = torch.empty(tb, dtype=torch.uint8, device=dev)
    _lib.check(lib.envgs_compact_scan(P, _lib.ptr(k8), _lib.ptr(pos), _lib.ptr(nk), _lib.ptr(temp), tb, None), "compact_scan")
    torch.cuda.synchronize()
    want = np.cumsum(keep.numpy().astype(np.int64))
    assert int(nk.item()) == int(want[-1])
    got = pos.cpu().numpy()[keep.numpy()]
    np.testing.assert_array_equal(got, (want - 1)[keep.numpy()])


def _numpy_ray_keys(ro, rd, lead=4):
    """envgs_amd/csrc/ray_key.h restated (float32; a ray exactly on a quantisation boundary may land one cell off: the test allows a handful):
    octahedral direction 2 x 8 bits, origin cell 3 x 5 bits inside the bounding box of the ray origins, interleaved from the top -- `lead` rounds
    of (u, v), then rounds of (u, v, x, y, z), then what is left of the direction."""
    f = np.float32
    inv = f(1) / (np.abs(rd).sum(1, dtype=f) + f(1e-30))
    u, v = rd[:, 0] * inv, rd[:, 1] * inv
    neg = rd[:, 2] < 0
    uu = (f(1) - np.abs(v)) * np.where(u >= 0, f(1), f(-1)); vv = (f(1) - np.abs(u)) * np.where(v >= 0, f(1), f(-1))
    u = np.where(neg, uu, u); v = np.where(neg, vv, v)
    q = lambda x: np.clip((x * f(0.5) + f(0.5)) * f(256), 0, 255).astype(np.uint32)
    qu, qv = q(u), q(v)
    lo, hi = ro.min(0), ro.max(0)
    ext = (hi - lo).astype(f)
    sc = np.where(ext > 0, f(32) / np.where(ext > 0, ext, f(1)), f(0)).astype(f)
    qo = [np.clip((ro[:, c] - lo[c]) * sc[c], 0, 31).astype(np.uint32) for c in range(3)]
    key = np.zeros(len(rd), np.uint32)
    di, oi = 7, 4
    for k in range(13):
        if di >= 0:
            key = (key << np.uint32(2)) | (((qv >> di) & 1) << 1) | ((qu >> di) & 1); di -= 1
        if k >= lead and oi >= 0:
            key = (key << np.uint32(3)) | (((qo[2] >> oi) & 1) << 2) | (((qo[1] >> oi) & 1) << 1) | ((qo[0] >> oi) & 1); oi -= 1
    return key.astype(np.uint32)


@pytest.mark.parametrize("R, kind", [(1, "random"), (63, "random"), (5000, "random"), (200000, "random"), (640000, "cone"), (300000, "parallel"), (500000, "clumps")])
def test_ray_coherence_order_is_the_stable_key_sort(R, kind):
    """The tracer's ray order (raster_bin.hip: launch_ray_sort -- buckets by the key's top bits, one LDS sort per bucket) is the order a
    stable sort of the 31-bit keys gives: ray ids by (key, id).  'cone': camera-like directions (few direction cells, long buckets);
    'parallel': every ray the same direction, i.e. ONE bucket: the long-list kernel sorts its 300 000 entries chunk by chunk, the whole grid
    on each phase; 'clumps': ten buckets of 17 000 - 100 000 rays (the bounce stages of the 1200x1600 configuration have dozens of them): the grid
    walks the phases of all of them together (phase-major items).  The scratch buffer is EXACTLY envgs_trace_ray_sort_temp_bytes(R) with a canary
    region behind it (the origin-bounds partials used to be written past the reported size)."""
    from envgs_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(R)
    ro = (torch.rand(R, 3, generator=gen) * 2 - 1) * 0.98
    if kind == "random": rd = torch.randn(R, 3, generator=gen)
    elif kind == "cone": rd = torch.cat([0.35 * (torch.rand(R, 2, generator=gen) * 2 - 1), torch.ones(R, 1)], 1)
    elif kind == "clumps":
        rd = torch.randn(R, 3, generator=gen)
        sizes = [17000, 20000, 24000, 30000, 33000, 40000, 50000, 65536, 80000, 100000]
        at = 0
        for i, n_ in enumerate(sizes):                                     # one direction cell + one origin octant each; the low key bits still differ
            dc = torch.tensor([0.9 * np.cos(0.6 * i), 0.9 * np.sin(0.6 * i), 0.4 + 0.05 * i], dtype=torch.float32)
            rd[at:at + n_] = dc + 0.004 * torch.randn(n_, 3, generator=gen)
            ro[at:at + n_] = 0.3 + 0.1 * torch.rand(n_, 3, generator=gen)
            at += n_
    else: rd = torch.tensor([[0.3, -0.2, 0.9]]).repeat(R, 1)
    rod, rdd = ro.to(dev).contiguous(), rd.to(dev).contiguous()
    pairs = torch.zeros(R, dtype=torch.int64, device=dev)
    order = torch.full((R,), -1, dtype=torch.int32, device=dev)
    tb = lib.envgs_trace_ray_sort_temp_bytes(R)
    CANARY = 1 << 16
    temp = torch.full((tb + CANARY,), 0xA5, dtype=torch.uint8, device=dev)
    p = _lib.ptr
    _lib.check(lib.envgs_trace_ray_order(R, p(rod), p(rdd), None, 0, p(pairs), p(order), p(temp), tb, None), "envgs_trace_ray_order")
    torch.cuda.synchronize()
    assert bool((temp[tb:] == 0xA5).all()), "launch_ray_sort wrote past envgs_trace_ray_sort_temp_bytes(R)"
    pr = pairs.cpu().numpy().view(np.uint64)
    ids = (pr & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(R))                                   # every ray placed exactly once
    keys = np.empty(R, np.uint32); keys[ids] = (pr >> np.uint64(32)).astype(np.uint32)
    assert int((keys != _numpy_ray_keys(ro.numpy(), rd.numpy())).sum()) <= max(2, R // 20000)      # the device keys are the documented function
    want = np.lexsort((np.arange(R), keys))                                              # stable sort by key
    np.testing.assert_array_equal(order.cpu().numpy().view(np.uint32).astype(np.int64), want)
