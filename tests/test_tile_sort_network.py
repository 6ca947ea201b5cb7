"""CPU check of the per-tile sorting networks (envgs_amd/csrc/tile_sort.h): the header is plain index arithmetic shared by the HIP kernels
(csrc/raster_bin.hip: sort_tile_lists / sort_long_lists) and this test, which compiles its host entry points with g++ and runs both networks
-- the register-blocked padded bitonic network with the LDS slot layout, and the all-ascending network used in place on HBM segments --
over every list length up to 300, lengths around the powers of two up to 20 000, random keys and keys full of ties."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "envgs_amd", "csrc", "tile_sort.h")


@pytest.fixture(scope="module")
def net():
    d = tempfile.mkdtemp(prefix="tile_sort_")
    src = os.path.join(d, "ts.cpp")
    with open(src, "w") as f:
        f.write('#include "%s"\n' % HDR)
    so = os.path.join(d, "libts.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = ctypes.CDLL(so)
    p64 = ctypes.POINTER(ctypes.c_uint64)
    lib.ts_sort_blocked.argtypes = [p64, ctypes.c_int, p64]
    lib.ts_sort_ascending.argtypes = [p64, ctypes.c_int]
    lib.ts_sort_hybrid.argtypes = [p64, ctypes.c_int, p64, ctypes.c_int]
    return lib


def _keys(n, kind, rng):
    if kind == "random":
        return rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
    if kind == "ties":                                                      # few distinct depths, distinct ids in the low word
        return (rng.integers(0, 5, size=n, dtype=np.uint64) << np.uint64(32)) | rng.permutation(n).astype(np.uint64)
    return np.arange(n, dtype=np.uint64)[::-1].copy()                       # descending


@pytest.mark.parametrize("kind", ["random", "ties", "descending"])
def test_both_networks_sort_every_length(net, kind):
    rng = np.random.default_rng(7)
    p64 = ctypes.POINTER(ctypes.c_uint64)
    lengths = list(range(1, 301)) + [511, 512, 513, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 10000,
                                     16383, 16384, 16385, 20000]
    for n in lengths:
        k = _keys(n, kind, rng)
        want = np.sort(k)
        a = k.copy()
        pad = np.zeros(2 * max(n, 16) + 2 * max(n, 16) // 32 + 64, dtype=np.uint64)
        net.ts_sort_blocked(a.ctypes.data_as(p64), n, pad.ctypes.data_as(p64))
        np.testing.assert_array_equal(a, want, err_msg="register-blocked network, n=%d" % n)
        b = k.copy()
        net.ts_sort_ascending(b.ctypes.data_as(p64), n)
        np.testing.assert_array_equal(b, want, err_msg="all-ascending network, n=%d" % n)


@pytest.mark.parametrize("lc", [4, 5, 7, 14])
def test_hybrid_schedule_for_lists_beyond_one_lds_array(net, lc):
    """sort_long_lists' schedule for segments longer than its LDS array (chunks of 2^lc entries sorted / merged in LDS, only the steps
    whose distance is at least a chunk run on the whole segment) -- with small chunks, so that many stages are 'wide', and at the kernel's
    own chunk size."""
    rng = np.random.default_rng(lc)
    p64 = ctypes.POINTER(ctypes.c_uint64)
    lengths = [1, 2, 15, 16, 17, 31, 33, 100, 255, 256, 257, 777, 1024, 1500, 4097] if lc < 14 else [16385, 40000, 100000]
    for n in lengths:
        for kind in ("random", "ties"):
            k = _keys(n, kind, rng)
            a = k.copy()
            pad = np.zeros((1 << lc) + (1 << lc) // 32 + 64, dtype=np.uint64)
            net.ts_sort_hybrid(a.ctypes.data_as(p64), n, pad.ctypes.data_as(p64), lc)
            np.testing.assert_array_equal(a, np.sort(k), err_msg="hybrid schedule, chunk 2^%d, n=%d" % (lc, n))
