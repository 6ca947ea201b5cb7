"""Host logic (CPU): the bucketed sizes of the tracer's large per-call scratch (envgs_amd/tracing.py:_scratch)."""
import torch

from envgs_amd import tracing


def test_small_requests_are_exact():
    t = tracing._scratch((1000, 64, 2), torch.int32, "cpu")
    assert t.shape == (1000, 64, 2) and t.is_contiguous() and t.untyped_storage().nbytes() == 1000 * 64 * 2 * 4


def test_large_requests_share_a_handful_of_block_sizes():
    sizes = set()
    for n in range(9_000_000, 40_000_000, 777_777):              # 36 MB .. 160 MB of float32: what changing ray counts / capacities produce
        t = tracing._scratch((n,), torch.float32, "cpu")
        assert t.shape == (n,) and t.is_contiguous()
        blk = t.untyped_storage().nbytes() // 4
        assert n <= blk <= n * 1.25 + 1                          # at most a quarter wasted
        sizes.add(blk)
    assert len(sizes) <= 10                                      # 40 different requests, a few distinct allocations
    for blk in sizes:                                            # {1, 1.25, 1.5, 1.75} x 2^k
        k = blk.bit_length() - 1
        assert (blk << 2) % (1 << k) == 0 and (blk << 2) >> k in (4, 5, 6, 7, 8)


def test_views_keep_their_shape_semantics():
    t = tracing._scratch((50_000, 320, 2), torch.int32, "cpu")   # 128 MB: bucketed
    assert t.shape == (50_000, 320, 2) and t.stride() == (640, 2, 1)
    t[49_999, 319, 1] = 7
    assert int(t.view(-1)[-1]) == 7
