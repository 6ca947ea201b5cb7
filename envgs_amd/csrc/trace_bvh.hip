// trace_bvh.hip -- T1: LBVH over the surfel proxies (replaces the OptiX GAS build of the reference).
//
//   quad AABBs + scene bounds -> 30-bit Morton code of the centroid || surfel id (unique 62-bit keys)
//   -> bucket sort on the rasterizer's binning machinery (launch_key_sort, raster_bin.hip: LDS histograms + one LDS bitonic sort per bucket;
//      no library sort is left in the library) -> Karras-2012 hierarchy (one lane per internal node, clz on key pairs)
//   -> sparse table of box unions over the SORTED leaves (st[k][i] = union of leaves [i, i + 2^k)): an LBVH node covers a contiguous
//      run of sorted leaves, so each internal node reads both child boxes as two overlapping power-of-two windows -- fully parallel,
//      no bottom-up walk, no atomics (the bottom-up fit was 1 ms of dependent device-scope round trips; the whole build is now 0.14 ms)
// Node = 64 B with both child boxes inline, so one 64 B fetch during traversal decides both children.
//
// Stands behind SurfelTracer.build_acceleration_structure (easyvolcap/utils/optix_utils.py:71-85): called every
// training iteration with rebuild=True, so the whole build is a handful of HBM-bound passes over P records.
#include "common.h"

#include <cstring>

#include "../../include/envgs_trace.h"

namespace envgs {

constexpr int NODE = ENVGS_NODE_STRIDE;

struct BvhTemp {
    uint64_t *keys_in, *keys_out;
    float *leaf_box;        // (P,6)
    float *partial;         // (nblocks,6)
    float *bounds;          // 6
    float *st;              // (levels, P, 6) sparse table of box unions over the sorted leaves: st[k][i] = union of leaves [i, i + 2^k)
    int levels;
    void *sort_temp;
    size_t sort_bytes;
    size_t total;
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static BvhTemp carve(int P, void *base)
{
    BvhTemp t;
    const int n = P > 0 ? P : 1;
    const int nblocks = (n + 255) / 256;
    const size_t sort_bytes = key_sort_temp_bytes(n);
    char *p = (char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char *r = p ? p + off : nullptr; off += align_up(bytes); return (void *)r; };
    t.keys_in = (uint64_t *)take(sizeof(uint64_t) * n);
    t.keys_out = (uint64_t *)take(sizeof(uint64_t) * n);
    t.leaf_box = (float *)take(sizeof(float) * 6 * n);
    t.partial = (float *)take(sizeof(float) * 6 * nblocks);
    t.bounds = (float *)take(sizeof(float) * 8);
    t.levels = 1;
    while ((2 << (t.levels - 1)) <= n) t.levels++;                       // floor(log2 n) + 1
    t.st = (float *)take(sizeof(float) * 6 * (size_t)n * t.levels);
    t.sort_temp = take(sort_bytes);
    t.sort_bytes = sort_bytes;
    t.total = off;
    return t;
}

// ---- 1. quad AABBs + per-block bounds -----------------------------------------------------------
__global__ void __launch_bounds__(256)
quad_boxes(int P, const float *__restrict__ verts, const float *__restrict__ opac, float *__restrict__ leaf_box,
           float *__restrict__ partial)
{
    __shared__ float s_red[6][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (i < P) {
        const float *v = verts + (size_t)i * 12;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) { const float x = v[k * 3 + c]; mn[c] = fminf(mn[c], x); mx[c] = fmaxf(mx[c], x); }
        if (opac) {
            // Opacity-aware tightening.  A hit needs alpha = o*exp(-(u^2+v^2)/2) >= 1/255, i.e. u^2+v^2 <= tau = 2 ln(255 o): the
            // contributing region is the 3-sigma quad INTERSECTED with the disc of radius sqrt(tau) (<= 3.33).  The disc's AABB has
            // half extent sqrt(tau) * sqrt(a_c^2 + b_c^2) per axis; the quad corners give centre mu, 6a = v2 - v0, 6b = v0 - v1.
            const float o = opac[i];
            const float tau = 2.0f * __logf(255.0f * o);
            if (!(tau > 0.0f)) {                 // can never contribute: shrink the box to the surfel's centre (keeps the tree tidy)
#pragma unroll
                for (int c = 0; c < 3; c++) { const float mu = 0.5f * (v[c] + v[9 + c]); mn[c] = mu; mx[c] = mu; }
            } else {
                const float rr = sqrtf(tau) * (1.0f + 1e-4f);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float mu = 0.5f * (v[c] + v[9 + c]);
                    const float a = (v[6 + c] - v[c]) * (1.0f / 6.0f), b = (v[c] - v[3 + c]) * (1.0f / 6.0f);
                    const float he = rr * sqrtf(a * a + b * b);
                    mn[c] = fmaxf(mn[c], mu - he); mx[c] = fminf(mx[c], mu + he);
                }
            }
        }
        // conservative pad: the hit test is analytic (|u|,|v| <= 3), the box comes from rounded vertices
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float pad = 1e-5f * (fabsf(mn[c]) + fabsf(mx[c]) + (mx[c] - mn[c])) + 1e-7f;
            mn[c] -= pad; mx[c] += pad;
            leaf_box[(size_t)i * 6 + c] = mn[c];
            leaf_box[(size_t)i * 6 + 3 + c] = mx[c];
        }
    }
    // block reduction of the bounds (wave shuffles, then 4 partials through LDS)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
        if (lane == 0) { s_red[c][wave] = a; s_red[3 + c][wave] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float r = s_red[c][0];
        for (int w = 1; w < 4; w++) r = c < 3 ? fminf(r, s_red[c][w]) : fmaxf(r, s_red[c][w]);
        partial[(size_t)blockIdx.x * 6 + c] = r;
    }
}

__global__ void __launch_bounds__(256) bounds_final(int nblocks, const float *__restrict__ partial, float *__restrict__ bounds)
{
    __shared__ float s_red[6][4];
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int c = 0; c < 3; c++) { mn[c] = fminf(mn[c], partial[(size_t)b * 6 + c]); mx[c] = fmaxf(mx[c], partial[(size_t)b * 6 + 3 + c]); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
        if (lane == 0) { s_red[c][wave] = a; s_red[3 + c][wave] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        float r = s_red[c][0];
        for (int w = 1; w < 4; w++) r = c < 3 ? fminf(r, s_red[c][w]) : fmaxf(r, s_red[c][w]);
        bounds[c] = r;
    }
}

// ---- 2. Morton keys -----------------------------------------------------------------------------
__device__ __forceinline__ uint32_t expand10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(256)
morton_keys(int P, const float *__restrict__ leaf_box, const float *__restrict__ bounds, uint64_t *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo = bounds[c], hi = bounds[3 + c];
        const float ctr = 0.5f * (leaf_box[(size_t)i * 6 + c] + leaf_box[(size_t)i * 6 + 3 + c]);
        const float ext = hi - lo;
        float u = ext > 0.f ? (ctr - lo) / ext : 0.f;
        u = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        code |= expand10((uint32_t)u) << (2 - c);
    }
    keys[i] = ((uint64_t)code << 32) | (uint32_t)i;
}

// ---- 3. Karras hierarchy ------------------------------------------------------------------------
__device__ __forceinline__ int delta(const uint64_t *__restrict__ keys, int P, int i, int j)
{
    if (j < 0 || j >= P) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// Box of the sorted leaves [lo, hi] from the sparse table: two overlapping power-of-two windows.
__device__ __forceinline__ void range_box(const float *__restrict__ st, int P, int lo, int hi, float *__restrict__ out)
{
    const int k = 31 - __clz(hi - lo + 1);
    const float *a = st + ((size_t)k * P + lo) * 6, *b = st + ((size_t)k * P + (hi - (1 << k) + 1)) * 6;
#pragma unroll
    for (int c = 0; c < 3; c++) { out[c] = fminf(a[c], b[c]); out[3 + c] = fmaxf(a[3 + c], b[3 + c]); }
}

// level 0 of the table: the leaf boxes in sorted order
__global__ void __launch_bounds__(256)
st_level0(int P, const uint64_t *__restrict__ keys, const float *__restrict__ leaf_box, float *__restrict__ st)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= P) return;
    const uint32_t sid = (uint32_t)(keys[j] & 0xFFFFFFFFu);
#pragma unroll
    for (int c = 0; c < 6; c++) st[(size_t)j * 6 + c] = leaf_box[(size_t)sid * 6 + c];
}

// levels k+1 and k+2 from level k (two per launch: half the launches, the table is tiny next to the launch gaps)
__global__ void __launch_bounds__(256)
st_levels(int P, int k, int levels, float *__restrict__ st)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float *src = st + (size_t)k * P * 6;
    const int h = 1 << k;
    const int i1 = min(i + h, P - 1), i2 = min(i + 2 * h, P - 1), i3 = min(i + 3 * h, P - 1);
    float a[6], b[6];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        a[c] = fminf(src[(size_t)i * 6 + c], src[(size_t)i1 * 6 + c]); a[3 + c] = fmaxf(src[(size_t)i * 6 + 3 + c], src[(size_t)i1 * 6 + 3 + c]);
        b[c] = fminf(src[(size_t)i2 * 6 + c], src[(size_t)i3 * 6 + c]); b[3 + c] = fmaxf(src[(size_t)i2 * 6 + 3 + c], src[(size_t)i3 * 6 + 3 + c]);
    }
    float *d1 = st + ((size_t)(k + 1) * P + i) * 6;
#pragma unroll
    for (int c = 0; c < 6; c++) d1[c] = a[c];
    if (k + 2 < levels) {
        float *d2 = st + ((size_t)(k + 2) * P + i) * 6;
#pragma unroll
        for (int c = 0; c < 3; c++) { d2[c] = fminf(a[c], b[c]); d2[3 + c] = fmaxf(a[3 + c], b[3 + c]); }
    }
}

__global__ void __launch_bounds__(256)
build_hierarchy(int P, const uint64_t *__restrict__ keys, const float *__restrict__ st, float *__restrict__ nodes)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P - 1) return;
    const int d = (delta(keys, P, i, i + 1) - delta(keys, P, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, P, i, i - d);
    int lmax = 2;
    while (delta(keys, P, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, P, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, P, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, P, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    // An LBVH node covers a contiguous run of the sorted leaves, so both child boxes are range unions: no bottom-up pass, no atomics,
    // no chain as long as the tree is deep (the fit used to be 1 ms of dependent device-scope round trips).
    const int left = lo == gamma ? ~(int)(uint32_t)(keys[gamma] & 0xFFFFFFFFu) : gamma;
    const int right = hi == gamma + 1 ? ~(int)(uint32_t)(keys[gamma + 1] & 0xFFFFFFFFu) : gamma + 1;
    float *nd = nodes + (size_t)i * NODE;
    float bl[6], br[6];
    range_box(st, P, lo, gamma, bl);
    range_box(st, P, gamma + 1, hi, br);
    float4 *n4 = reinterpret_cast<float4 *>(nd);
    n4[0] = make_float4(bl[0], bl[1], bl[2], bl[3]);
    n4[1] = make_float4(bl[4], bl[5], br[0], br[1]);
    n4[2] = make_float4(br[2], br[3], br[4], br[5]);
    nd[12] = __int_as_float(left);
    nd[13] = __int_as_float(right);
    if (i == 0) nd[14] = __int_as_float(-1);
    if (left >= 0) nodes[(size_t)left * NODE + 14] = __int_as_float(i);          // parent links (diagnostics)
    if (right >= 0) nodes[(size_t)right * NODE + 14] = __int_as_float(i);
    nd[15] = 0.f;
}

// P == 1: a single node whose left child is the only surfel and whose right child can never be hit.
__global__ void single_leaf_node(const float *__restrict__ leaf_box, float *__restrict__ nodes)
{
    if (threadIdx.x == 0) {
        for (int c = 0; c < 6; c++) nodes[c] = leaf_box[c];
        // a degenerate far-away box: an INVERTED box would pass the min/max slab test, a far point never does
        for (int c = 0; c < 3; c++) { nodes[6 + c] = 1.0e30f; nodes[9 + c] = 1.0e30f; }
        nodes[12] = __int_as_float(~0);
        nodes[13] = __int_as_float(~0);
        nodes[14] = __int_as_float(-1);
        nodes[15] = 0.f;
    }
}

// 4-wide nodes for the packet traversal: node4[i] holds the GRANDCHILDREN of binary node i (a child that is a leaf stays as it is), 4 slots of
// 32 B = 128 B.  Every binary node gets one (the traversal
// only ever follows every other level; the rest is 128 B per surfel of unused memory) so the kernel is a plain map, no compaction.  Empty slots
// are far-away points that no ray's slab test passes.
__global__ void __launch_bounds__(256)
build_wide_nodes(int n_internal, const float *__restrict__ nodes, float *__restrict__ nodes4)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_internal) return;
    float lo[4][3], hi[4][3];
    int ref[4];
    int k = 0;
    const float *nd = nodes + (size_t)i * NODE;
    for (int side = 0; side < 2; side++) {
        const int c = __float_as_int(nd[12 + side]);
        if (c < 0) {
            for (int a = 0; a < 3; a++) { lo[k][a] = nd[side * 6 + a]; hi[k][a] = nd[side * 6 + 3 + a]; }
            ref[k++] = c;
        } else {
            const float *cd = nodes + (size_t)c * NODE;
            for (int s2 = 0; s2 < 2; s2++) {
                for (int a = 0; a < 3; a++) { lo[k][a] = cd[s2 * 6 + a]; hi[k][a] = cd[s2 * 6 + 3 + a]; }
                ref[k++] = __float_as_int(cd[12 + s2]);
            }
        }
    }
    for (; k < 4; k++) {                                   // unused slots (a child that is itself a leaf has no grandchildren): a box no ray reaches,
        for (int a = 0; a < 3; a++) { lo[k][a] = 1.0e30f; hi[k][a] = 1.0e30f; }      // and WIDE_EMPTY as the reference so that the traversal skips the test
        ref[k] = ENVGS_WIDE_EMPTY;
    }
    // per slot 8 floats: [lo.x hi.x lo.y hi.y | lo.z hi.z ref 0] -- (lo, hi) of an axis adjacent, so that they arrive as an aligned scalar
    // register PAIR and the slab test's subtract and multiply run as packed fp32 (v_pk_add_f32 / v_pk_mul_f32: both planes in one instruction)
    float *o = nodes4 + (size_t)i * 32;
    for (int q = 0; q < 4; q++) {
        o[q * 8 + 0] = lo[q][0]; o[q * 8 + 1] = hi[q][0]; o[q * 8 + 2] = lo[q][1]; o[q * 8 + 3] = hi[q][1];
        o[q * 8 + 4] = lo[q][2]; o[q * 8 + 5] = hi[q][2]; o[q * 8 + 6] = __int_as_float(ref[q]); o[q * 8 + 7] = 0.f;
    }
}

}  // namespace envgs

using namespace envgs;

extern "C" {

size_t envgs_bvh_temp_bytes(int32_t P) { return carve(P, nullptr).total; }

size_t envgs_bvh_node_floats(int32_t P) { return (size_t)(NODE + 32) * (size_t)(P > 1 ? P - 1 : 1); }

int envgs_bvh_build(int32_t P, const float *vertices, const float *opacities, float *nodes, void *temp, size_t temp_bytes, int32_t debug,
                    void *stream_)
{
    if (P < 0) return ENVGS_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (!vertices || !nodes || !temp) return ENVGS_ERR_BAD_ARG;
    BvhTemp t = carve(P, temp);
    if (temp_bytes < t.total) return ENVGS_ERR_TEMP_TOO_SMALL;
    hipStream_t stream = (hipStream_t)stream_;
    envgs_raster_cfg dbg; dbg.debug = debug;
    const envgs_raster_cfg *cfg = &dbg;
    ProfScope prof_(K_BVH_BUILD, stream);
    const int nblocks = (P + 255) / 256;
    hipLaunchKernelGGL(quad_boxes, dim3(nblocks), dim3(256), 0, stream, P, vertices, opacities, t.leaf_box, t.partial);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    if (P == 1) {
        hipLaunchKernelGGL(single_leaf_node, dim3(1), dim3(64), 0, stream, t.leaf_box, nodes);
        ENVGS_CHECK_LAUNCH(cfg, stream);
        hipLaunchKernelGGL(build_wide_nodes, dim3(1), dim3(256), 0, stream, 1, nodes, nodes + NODE);
        ENVGS_CHECK_LAUNCH(cfg, stream);
        return 0;
    }
    hipLaunchKernelGGL(bounds_final, dim3(1), dim3(256), 0, stream, nblocks, t.partial, t.bounds);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(morton_keys, dim3(nblocks), dim3(256), 0, stream, P, t.leaf_box, t.bounds, t.keys_in);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    // Morton code << 32 | surfel id, ascending: buckets by the code's top bits, one LDS sort per bucket (raster_bin.hip: launch_key_sort)
    const int rc_sort = launch_key_sort(P, t.keys_in, t.keys_out, 62, t.sort_temp, t.sort_bytes, stream);
    if (rc_sort) return rc_sort;
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(st_level0, dim3(nblocks), dim3(256), 0, stream, P, t.keys_out, t.leaf_box, t.st);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    for (int k = 0; k + 1 < t.levels; k += 2) {
        hipLaunchKernelGGL(st_levels, dim3(nblocks), dim3(256), 0, stream, P, k, t.levels, t.st);
        ENVGS_CHECK_LAUNCH(cfg, stream);
    }
    hipLaunchKernelGGL(build_hierarchy, dim3((P - 1 + 255) / 256), dim3(256), 0, stream, P, t.keys_out, t.st, nodes);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(build_wide_nodes, dim3((P - 1 + 255) / 256), dim3(256), 0, stream, P - 1, nodes, nodes + (size_t)(P - 1) * NODE);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return 0;
}

}  // extern "C"
