// raster_bin.hip -- R2..R5: prefix sum of tiles_touched, (tile id << 32 | depth bits) key emit,
// stable LSD radix sort of (key, surfel id) pairs, per-tile [start, end) ranges.
// All integer work: results are bit-exact against the oracle (tests/test_raster_parity.py).
// Scan and pair sort use rocPRIM's device-wide primitives (the vendor library plays the role CUB plays
// upstream); emit / ranges are hand-written.  HBM-bound: N * ~164 B.
#include "common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace envgs {

size_t scan_temp_bytes(int n)
{
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)(n > 0 ? n : 1),
                            rocprim::plus<uint32_t>());
    return bytes;
}

int launch_scan(const uint32_t *in, uint32_t *out, int n, void *temp, size_t temp_bytes, hipStream_t stream)
{
    if (n <= 0) return 0;
    ProfScope prof_(K_SCAN, stream);
    hipError_t e = rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), stream);
    return (int)e;
}

size_t sort_temp_bytes(uint32_t n, int end_bit)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)(n > 0 ? n : 1), 0u, (unsigned)end_bit);
    return bytes;
}

// One lane per surfel; each visible surfel writes its tiles_touched instances at offsets[i-1].
// The tile rect is recomputed from the stored centre and INTEGER radius exactly as R1 did.
__global__ void __launch_bounds__(256)
emit_tile_keys(int P, int W, int H, const float *__restrict__ geom, const int32_t *__restrict__ radii,
               const uint32_t *__restrict__ offsets, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t cap)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int rad = radii[i];
    if (rad <= 0) return;
    uint32_t off = (i == 0) ? 0u : offsets[i - 1];
    const float cx = geom[(size_t)i * GEOM + 9], cy = geom[(size_t)i * GEOM + 10];
    const uint32_t dbits = __float_as_uint(geom[(size_t)i * GEOM + 15]);
    const float radius = (float)rad;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int x0 = (int)((cx - radius) / (float)TILE); x0 = x0 < 0 ? 0 : (x0 > gx ? gx : x0);
    int y0 = (int)((cy - radius) / (float)TILE); y0 = y0 < 0 ? 0 : (y0 > gy ? gy : y0);
    int x1 = (int)((cx + radius + (float)(TILE - 1)) / (float)TILE); x1 = x1 < 0 ? 0 : (x1 > gx ? gx : x1);
    int y1 = (int)((cy + radius + (float)(TILE - 1)) / (float)TILE); y1 = y1 < 0 ? 0 : (y1 > gy ? gy : y1);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const uint64_t key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
            if (off < cap) { keys[off] = key; vals[off] = (uint32_t)i; }      // (cap < N only when a speculative capacity was too small: the caller repeats the call)
            off++;
        }
}

// Speculative capacity (launch_bin): the slots [N, cap) of the key buffer sort behind every real key.
__global__ void __launch_bounds__(256)
pad_tile_keys(uint32_t cap, const uint32_t *__restrict__ n_dev, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < cap && j >= *n_dev) { keys[j] = ~0ull; vals[j] = 0u; }
}

__global__ void __launch_bounds__(256)
find_tile_ranges(uint32_t N, const uint32_t *__restrict__ n_dev, const uint64_t *__restrict__ keys_sorted, uint32_t *__restrict__ ranges)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) N = min(N, *n_dev);
    if (j >= N) return;
    const uint32_t t = (uint32_t)(keys_sorted[j] >> 32);
    if (j == 0 || t != (uint32_t)(keys_sorted[j - 1] >> 32)) ranges[2 * t] = j;
    if (j == N - 1 || t != (uint32_t)(keys_sorted[j + 1] >> 32)) ranges[2 * t + 1] = j + 1;
}

int launch_bin(const envgs_raster_cfg *cfg, uint32_t N, const float *geom, const int32_t *radii, const uint32_t *offsets,
               uint64_t *keys_unsorted, uint32_t *vals_unsorted, uint64_t *keys_sorted, uint32_t *point_list,
               void *sort_temp, size_t sort_temp_bytes, uint32_t *ranges, hipStream_t stream, const uint32_t *n_dev)
{
    // n_dev == nullptr: N is the exact number of tile instances.  Otherwise N is a CAPACITY chosen before the count was known on the host
    // (no host sync between projection and binning) and *n_dev the count: the tail [count, N) is padded with keys that sort last (one more
    // key bit makes them larger than any tile id), and the ranges are built from the first `count` sorted entries only.
    const int gx = (cfg->width + TILE - 1) / TILE, gy = (cfg->height + TILE - 1) / TILE;
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy, stream);
    if (e != hipSuccess) return (int)e;
    if (N == 0 || cfg->P <= 0) return 0;
    prof_begin(K_EMIT_KEYS, stream);
    hipLaunchKernelGGL(emit_tile_keys, dim3((cfg->P + 255) / 256), dim3(256), 0, stream, cfg->P, cfg->width, cfg->height,
                       geom, radii, offsets, keys_unsorted, vals_unsorted, N);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    if (n_dev) {
        hipLaunchKernelGGL(pad_tile_keys, dim3((N + 255) / 256), dim3(256), 0, stream, N, n_dev, keys_unsorted, vals_unsorted);
        ENVGS_CHECK_LAUNCH(cfg, stream);
    }
    prof_end(K_EMIT_KEYS, stream);
    const int end_bit = 32 + tile_bits(cfg->width, cfg->height) + (n_dev ? 1 : 0);
    prof_begin(K_SORT, stream);
    size_t need = sort_temp_bytes;
    e = rocprim::radix_sort_pairs(sort_temp, need, keys_unsorted, keys_sorted, vals_unsorted, point_list, (size_t)N, 0u,
                                  (unsigned)end_bit, stream);
    prof_end(K_SORT, stream);
    if (e != hipSuccess) return (int)e;
    ProfScope prof_(K_RANGES, stream);
    hipLaunchKernelGGL(find_tile_ranges, dim3((N + 255) / 256), dim3(256), 0, stream, N, n_dev, keys_sorted, ranges);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return 0;
}

}  // namespace envgs
