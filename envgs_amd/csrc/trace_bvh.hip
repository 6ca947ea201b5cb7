// trace_bvh.hip -- T1: LBVH over the surfel proxies (replaces the OptiX GAS build of the reference).
//
//   quad AABBs + scene bounds -> 30-bit Morton code of the centroid || surfel id (unique 62-bit keys)
//   -> bucket sort on the rasterizer's binning machinery (launch_key_sort, raster_bin.hip: LDS histograms + one LDS bitonic sort per bucket;
//      no library sort is left in the library) -> Karras-2012 hierarchy (one lane per internal node, clz on key pairs)
//   -> box unions over the SORTED leaves: an LBVH node covers a contiguous run of sorted leaves, so each internal node reads both child boxes
//      as range unions -- fully parallel, no bottom-up walk, no atomics (the bottom-up fit was 1 ms of dependent device-scope round trips).
//      Round 4: THREE tiers instead of floor(log2 P) + 1 full levels (18 levels x P x 24 B = 70 MB and nine launches at 163 840 surfels):
//      windows of 1 .. 64 sorted leaves (7 levels), windows of 1 .. 64 BLOCKS of 64 leaves (7 levels of P / 64 entries, built the same way),
//      and a full sparse table over SUPER-blocks of 4096 leaves (a handful of entries); a range = its first and last leaf window + the aligned
//      blocks inside it (first and last block window + the aligned super-blocks inside those).  28 MB, three launches.
//   -> the fit (fit_nodes) reads only the stored TOPOLOGY (child references + the other end of each node's leaf range) and the tables, so the
//      same kernel REFITS an existing tree to moved vertices / changed opacities (envgs_bvh_refit: no Morton keys, no sort, no Karras pass).
// Node = 64 B with both child boxes inline, so one 64 B fetch during traversal decides both children.
//
// Stands behind SurfelTracer.build_acceleration_structure (easyvolcap/utils/optix_utils.py:71-85): called every
// training iteration with rebuild=True, so the whole build is a handful of HBM-bound passes over P records.
#include "common.h"

#include <cstring>

#include "../../include/envgs_trace.h"

namespace envgs {

constexpr int NODE = ENVGS_NODE_STRIDE;

constexpr int ST_LOW = 6;            // the leaf-level sparse table holds windows of 2^0 .. 2^ST_LOW sorted leaves
constexpr int BLK = 1 << ST_LOW;     // leaves per block of the block-level table

struct BvhTemp {
    uint64_t *keys_in, *keys_out;
    float *leaf_box;        // (P,6)
    float *partial;         // (nblocks,12): box of the block's leaf boxes, sum and sum of squares of their centres
    float *bounds;          // scene box (6), mean (3) and 1 / (2.5 sigma) (3) of the leaf-box centres
    float *st;              // (ST_LOW + 1, P, 6): st[k][i] = union of the sorted leaves [i, i + 2^k)
    float *bt;              // (ST_LOW + 1, nblk, 6): bt[m][b] = union of the leaf blocks [b, b + 2^m)
    float *sup;             // (mlev, nsup, 6): full sparse table over the super-blocks (64 blocks each)
    int nblk, nsup, mlev;
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
    t.partial = (float *)take(sizeof(float) * 12 * nblocks);
    t.bounds = (float *)take(sizeof(float) * 16);
    t.st = (float *)take(sizeof(float) * 6 * (size_t)n * (ST_LOW + 1));
    t.nblk = (n + BLK - 1) / BLK;
    t.bt = (float *)take(sizeof(float) * 6 * (size_t)t.nblk * (ST_LOW + 1));
    t.nsup = (t.nblk + BLK - 1) / BLK;
    t.mlev = 1;
    while ((2 << (t.mlev - 1)) <= t.nsup) t.mlev++;                       // floor(log2 nsup) + 1
    t.sup = (float *)take(sizeof(float) * 6 * (size_t)t.nsup * t.mlev);
    t.sort_temp = take(sort_bytes);
    t.sort_bytes = sort_bytes;
    t.total = off;
    return t;
}

// ---- 1. quad AABBs + per-block bounds -----------------------------------------------------------
__device__ __forceinline__ void quad_box(const int i, const float *__restrict__ verts, const float *__restrict__ opac, float *mn, float *mx)
{
    const float *v = verts + (size_t)i * 12;
#pragma unroll
    for (int c = 0; c < 3; c++) { mn[c] = 3.0e38f; mx[c] = -3.0e38f; }
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
    }
}

__global__ void __launch_bounds__(256)
quad_boxes(int P, const float *__restrict__ verts, const float *__restrict__ opac, float *__restrict__ leaf_box,
           float *__restrict__ partial)
{
    __shared__ float s_red[12][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (i < P) {
        quad_box(i, verts, opac, mn, mx);
#pragma unroll
        for (int c = 0; c < 3; c++) { leaf_box[(size_t)i * 6 + c] = mn[c]; leaf_box[(size_t)i * 6 + 3 + c] = mx[c]; }
    }
    // block reduction (wave shuffles, then 4 partials through LDS): the bounds, and the first two moments of the leaf-box centres
    float sm[3] = {0.f, 0.f, 0.f}, sq[3] = {0.f, 0.f, 0.f};
    if (i < P) {
#pragma unroll
        for (int c = 0; c < 3; c++) { const float x = 0.5f * (mn[c] + mx[c]); if (fabsf(x) < 1.0e18f) { sm[c] = x; sq[c] = x * x; } }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = mn[c], b = mx[c], u = sm[c], w = sq[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); u += __shfl_xor(u, o); w += __shfl_xor(w, o); }
        if (lane == 0) { s_red[c][wave] = a; s_red[3 + c][wave] = b; s_red[6 + c][wave] = u; s_red[9 + c][wave] = w; }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int c = threadIdx.x;
        float r = s_red[c][0];
        for (int w = 1; w < 4; w++) r = c < 3 ? fminf(r, s_red[c][w]) : (c < 6 ? fmaxf(r, s_red[c][w]) : r + s_red[c][w]);
        partial[(size_t)blockIdx.x * 12 + c] = r;
    }
}

__global__ void __launch_bounds__(256) bounds_final(int P, int nblocks, const float *__restrict__ partial, float *__restrict__ bounds)
{
    __shared__ float s_red[12][4];
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, sm[3] = {0.f, 0.f, 0.f}, sq[3] = {0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblocks; b += 256)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            mn[c] = fminf(mn[c], partial[(size_t)b * 12 + c]); mx[c] = fmaxf(mx[c], partial[(size_t)b * 12 + 3 + c]);
            sm[c] += partial[(size_t)b * 12 + 6 + c]; sq[c] += partial[(size_t)b * 12 + 9 + c];
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float a = mn[c], b = mx[c], u = sm[c], w = sq[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); u += __shfl_xor(u, o); w += __shfl_xor(w, o); }
        if (lane == 0) { s_red[c][wave] = a; s_red[3 + c][wave] = b; s_red[6 + c][wave] = u; s_red[9 + c][wave] = w; }
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int c = threadIdx.x;
        float r = s_red[c][0];
        for (int w = 1; w < 4; w++) r = c < 3 ? fminf(r, s_red[c][w]) : (c < 6 ? fmaxf(r, s_red[c][w]) : r + s_red[c][w]);
        s_red[c][0] = r;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        bounds[c] = s_red[c][0]; bounds[3 + c] = s_red[3 + c][0];
        const float mean = s_red[6 + c][0] / (float)P;
        const float var = fmaxf(s_red[9 + c][0] / (float)P - mean * mean, 0.f);
        const float ext = s_red[3 + c][0] - s_red[c][0];
        // Morton mapping of this axis (morton_coord).  A scene whose box lies within mean +- 2.5 sigma -- uniform, Gaussian, anything without far
        // outliers -- is mapped LINEARLY onto the whole code range, as in rounds 1-3; otherwise the linear part covers mean +- 2.5 sigma (80 % of
        // the range) and the tails are squeezed into the rest.
        const float hext = fmaxf(s_red[3 + c][0] - mean, mean - s_red[c][0]) * 1.0001f;
        float half = 2.5f * sqrtf(var);
        const bool plain = !(half > 0.f) || !(half < 3.0e37f) || hext <= half;
        if (plain) half = hext;
        bounds[6 + c] = mean;
        bounds[9 + c] = (half > 0.f && ext > 0.f) ? 1.0f / half : 0.f;
        bounds[12 + c] = plain ? 0.5f : 0.4f;
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

// Coordinate -> [0, 1) for the 10-bit Morton cells.  A scene without far outliers (its box within mean +- 2.5 sigma of the leaf-box centres):
// linear over the box, as before.  Otherwise linear over mean +- 2.5 sigma (80 % of the code range), the tails squeezed into the outer 10 % on
// either side, (|z| - 1) / (1 + (|z| - 1)).  Rounds 1-3 mapped the
// scene BOX linearly: a handful of far outlier surfels -- common in Gaussian-splat training -- stretched the box until every other surfel fell
// into a few cells, i.e. nearly equal codes: a tree ordered by surfel id, and one giant bucket for the key sort.  Monotone per axis,
// so the Morton order is as spatially coherent as before.
__device__ __forceinline__ float morton_coord(const float x, const float mean, const float inv_half, const float lin)
{
    const float z = (x - mean) * inv_half;                     // +-1 at the end of the linear part (the scene's half extent, or 2.5 sigma)
    const float az = fabsf(z);
    float u = az <= 1.0f ? lin * az : lin + (0.5f - lin) * ((az - 1.0f) / (1.0f + (az - 1.0f)));
    u = z < 0.f ? 0.5f - u : 0.5f + u;
    return u == u ? u : 0.5f;
}

__global__ void __launch_bounds__(256)
morton_keys(int P, const float *__restrict__ leaf_box, const float *__restrict__ bounds, uint64_t *__restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float ctr = 0.5f * (leaf_box[(size_t)i * 6 + c] + leaf_box[(size_t)i * 6 + 3 + c]);
        float u = morton_coord(ctr, bounds[6 + c], bounds[9 + c], bounds[12 + c]);
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

// Box of the sorted leaves [lo, hi] from three tiers of windows: leaves (1 .. 64), blocks of 64 leaves (1 .. 64 blocks), super-blocks of 64 blocks
// (full sparse table).  A range longer than 64 entries of a tier = its first and last 64-window (they cover the partial coarser units at both
// ends) + the ALIGNED coarser units that lie inside it, looked up one tier up.  min / max only: the result is exactly the union of the leaf
// boxes, whatever the decomposition.
struct Tables { const float *st, *bt, *sup; int P, nblk, nsup; };

__device__ __forceinline__ void box_union(float *__restrict__ out, const float *__restrict__ a)
{
#pragma unroll
    for (int c = 0; c < 3; c++) { out[c] = fminf(out[c], a[c]); out[3 + c] = fmaxf(out[3 + c], a[3 + c]); }
}

__device__ __forceinline__ void range_box(const Tables T, int lo, int hi, float *__restrict__ out)
{
    const int len = hi - lo + 1;
    const int k = len <= BLK ? 31 - __clz(len) : ST_LOW;
    const float *a = T.st + ((size_t)k * T.P + lo) * 6;
#pragma unroll
    for (int c = 0; c < 6; c++) out[c] = a[c];
    box_union(out, T.st + ((size_t)k * T.P + (hi - (1 << k) + 1)) * 6);
    if (len <= BLK) return;
    const int b0 = (lo + BLK - 1) >> ST_LOW, b1 = (hi - (BLK - 1)) >> ST_LOW;        // aligned blocks [64 b, 64 b + 63] inside [lo, hi]
    if (b0 > b1) return;
    const int nb = b1 - b0 + 1;
    const int m = nb <= BLK ? 31 - __clz(nb) : ST_LOW;
    box_union(out, T.bt + ((size_t)m * T.nblk + b0) * 6);
    box_union(out, T.bt + ((size_t)m * T.nblk + (b1 - (1 << m) + 1)) * 6);
    if (nb <= BLK) return;
    const int s0 = (b0 + BLK - 1) >> ST_LOW, s1 = (b1 - (BLK - 1)) >> ST_LOW;        // aligned super-blocks inside [b0, b1]
    if (s0 > s1) return;
    const int q = 31 - __clz(s1 - s0 + 1);
    box_union(out, T.sup + ((size_t)q * T.nsup + s0) * 6);
    box_union(out, T.sup + ((size_t)q * T.nsup + (s1 - (1 << q) + 1)) * 6);
}

// The leaf-level table, all ST_LOW + 1 levels in ONE launch: a workgroup owns 256 consecutive sorted positions, stages their leaf boxes and a halo
// of 63 more in LDS (level 0 = the boxes in sorted order: gathered from the per-surfel boxes in a build, computed straight from the moved quads in
// the STORED leaf order in a refit) and doubles the windows in place, level by level, storing each level as it is formed.  (Rounds 1-3: a launch
// per two levels over the whole array, nine launches for 18 levels.)  Positions beyond the last leaf repeat it (min / max: no effect).
// MODE 0: build (gather by the sorted keys), 1: refit (quads in the stored order), 2: the BLOCK tier -- entry b = the 64-window of the tier
// below that starts at its position 64 b (leaf_box = that tier's top level, P = number of blocks, src_n = entries of the tier below).
template <int MODE>
__global__ void __launch_bounds__(256)
st_table(int P, const uint64_t *__restrict__ keys, const float *__restrict__ leaf_box, const int *__restrict__ order, const float *__restrict__ verts,
         const float *__restrict__ opac, float *__restrict__ st, int src_n)
{
    constexpr int W = 256 + BLK;             // 320 staged positions
    __shared__ float s_box[W][7];            // (7: odd stride, conflict-free row access)
    const int g0 = blockIdx.x * 256;
    for (int j = threadIdx.x; j < W; j += 256) {
        const int pos = min(g0 + j, P - 1);
        float mn[3], mx[3];
        if (MODE == 1) quad_box(order[pos], verts, opac, mn, mx);
        else if (MODE == 2) {
            const float *sp = leaf_box + (size_t)min(pos * BLK, src_n - 1) * 6;
#pragma unroll
            for (int c = 0; c < 3; c++) { mn[c] = sp[c]; mx[c] = sp[3 + c]; }
        } else {
            const uint32_t sid = (uint32_t)(keys[pos] & 0xFFFFFFFFu);
#pragma unroll
            for (int c = 0; c < 3; c++) { mn[c] = leaf_box[(size_t)sid * 6 + c]; mx[c] = leaf_box[(size_t)sid * 6 + 3 + c]; }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) { s_box[j][c] = mn[c]; s_box[j][3 + c] = mx[c]; }
    }
    __syncthreads();
    const int i = g0 + (int)threadIdx.x;
    for (int k = 0; k <= ST_LOW; k++) {
        if (i < P) {
            float *d = st + ((size_t)k * P + i) * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) d[c] = s_box[threadIdx.x][c];
        }
        if (k == ST_LOW) break;
        const int h = 1 << k;
        float v[2][6];
        bool on[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int j = (int)threadIdx.x + 256 * r;
            on[r] = j + h < W;               // (windows that would reach beyond the staged halo are never read by a later level of an owned position)
            if (on[r]) {
#pragma unroll
                for (int c = 0; c < 3; c++) { v[r][c] = fminf(s_box[j][c], s_box[j + h][c]); v[r][3 + c] = fmaxf(s_box[j][3 + c], s_box[j + h][3 + c]); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int j = (int)threadIdx.x + 256 * r;
            if (on[r] && j < W) {
#pragma unroll
                for (int c = 0; c < 6; c++) s_box[j][c] = v[r][c];
            }
        }
        __syncthreads();
    }
}

// The top tier, one workgroup: a full sparse table over the SUPER-blocks (64 blocks = 4096 leaves each; 171 of them at the 700 000-surfel cap):
// level 0 = the block tier's 64-windows at stride 64, then floor(log2 n) doubling levels.  (The first version of the two-level scheme gave the
// whole BLOCK tier to one workgroup: 10 938 entries x 14 levels of dependent L2 round trips, 0.45 ms -- most of the build.)
__global__ void __launch_bounds__(1024)
super_table(int nblk, int nsup, int mlev, const float *__restrict__ bt_top, float *__restrict__ sup)
{
    for (int b = threadIdx.x; b < nsup; b += 1024) {
        const float *srcp = bt_top + (size_t)min(b * BLK, nblk - 1) * 6;
#pragma unroll
        for (int c = 0; c < 6; c++) sup[(size_t)b * 6 + c] = srcp[c];
    }
    __syncthreads();
    for (int m = 1; m < mlev; m++) {
        const float *src = sup + (size_t)(m - 1) * nsup * 6;
        float *dst = sup + (size_t)m * nsup * 6;
        const int h = 1 << (m - 1);
        for (int b = threadIdx.x; b < nsup; b += 1024) {
            const int b1 = min(b + h, nsup - 1);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                dst[(size_t)b * 6 + c] = fminf(src[(size_t)b * 6 + c], src[(size_t)b1 * 6 + c]);
                dst[(size_t)b * 6 + 3 + c] = fmaxf(src[(size_t)b * 6 + 3 + c], src[(size_t)b1 * 6 + 3 + c]);
            }
        }
        __syncthreads();
    }
}

// Karras 2012: the TOPOLOGY only.  Node words 12, 13 = child references (>= 0 internal node, < 0 = ~surfel id), 14 = parent (diagnostics),
// 15 = the other end j of the node's run of sorted leaves (the run is [min(i, j), max(i, j)]; its split is gamma = left >= 0 ? left : min(i, j)).
// Also stores the sorted leaf order behind the node arrays: together with the topology it is all a refit needs.
__global__ void __launch_bounds__(256)
build_hierarchy(int P, const uint64_t *__restrict__ keys, float *__restrict__ nodes, int *__restrict__ order)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P - 1) return;
    order[i] = (int)(uint32_t)(keys[i] & 0xFFFFFFFFu);
    if (i == P - 2) order[P - 1] = (int)(uint32_t)(keys[P - 1] & 0xFFFFFFFFu);
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
    const int left = lo == gamma ? ~(int)(uint32_t)(keys[gamma] & 0xFFFFFFFFu) : gamma;
    const int right = hi == gamma + 1 ? ~(int)(uint32_t)(keys[gamma + 1] & 0xFFFFFFFFu) : gamma + 1;
    float *nd = nodes + (size_t)i * NODE;
    nd[12] = __int_as_float(left);
    nd[13] = __int_as_float(right);
    if (i == 0) nd[14] = __int_as_float(-1);
    if (left >= 0) nodes[(size_t)left * NODE + 14] = __int_as_float(i);          // parent links (diagnostics)
    if (right >= 0) nodes[(size_t)right * NODE + 14] = __int_as_float(i);
    nd[15] = __int_as_float(j);
}

// The fit, build and refit alike: both child boxes of binary node i (range unions) and its 4-wide node -- the GRANDCHILDREN of node i (a child that
// is a leaf stays as it is), read off the children's stored topology, 4 slots of 32 B = 128 B.  Every binary node gets a wide node (the packet
// traversal only ever follows every other level; the rest is 128 B per surfel of unused memory) so the kernel is a plain map, no compaction.
// Unused slots are far-away points that no ray's slab test passes, with WIDE_EMPTY as the reference so that the traversal skips the test.
// per slot 8 floats: [lo.x hi.x lo.y hi.y | lo.z hi.z ref 0] -- (lo, hi) of an axis adjacent, so that they arrive as an aligned scalar
// register PAIR and the slab test's subtract and multiply run as packed fp32 (v_pk_add_f32 / v_pk_mul_f32: both planes in one instruction)
__device__ __forceinline__ void node_ranges(const float *__restrict__ nodes, const int i, int &left, int &right, int &lo, int &gamma, int &hi)
{
    const float *nd = nodes + (size_t)i * NODE;
    left = __float_as_int(nd[12]); right = __float_as_int(nd[13]);
    const int j = __float_as_int(nd[15]);
    lo = min(i, j); hi = max(i, j);
    gamma = left >= 0 ? left : lo;
}

__global__ void __launch_bounds__(256)
fit_nodes(int P, const Tables T, const float *topo, float *nodes, float *__restrict__ nodes4)
{
    // topo: the node array the topology is read from -- `nodes` itself in a build, the PREVIOUS structure in a refit (which writes a fresh one:
    // an earlier forward whose backward is still outstanding may hold the old buffer); topology words and the leaf order are carried over
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P - 1) return;
    int left, right, lo, gamma, hi;
    node_ranges(topo, i, left, right, lo, gamma, hi);
    if (topo != nodes) {
        const float *ti = topo + (size_t)i * NODE;
        float *no = nodes + (size_t)i * NODE;
        no[12] = ti[12]; no[13] = ti[13]; no[14] = ti[14]; no[15] = ti[15];
        const float *oin = topo + (size_t)(P - 1) * (NODE + 32);
        float *oout = nodes + (size_t)(P - 1) * (NODE + 32);
        oout[i] = oin[i];
        if (i == P - 2) oout[P - 1] = oin[P - 1];
    }
    float box[2][6];
    range_box(T, lo, gamma, box[0]);
    range_box(T, gamma + 1, hi, box[1]);
    float4 *n4 = reinterpret_cast<float4 *>(nodes + (size_t)i * NODE);
    n4[0] = make_float4(box[0][0], box[0][1], box[0][2], box[0][3]);
    n4[1] = make_float4(box[0][4], box[0][5], box[1][0], box[1][1]);
    n4[2] = make_float4(box[1][2], box[1][3], box[1][4], box[1][5]);
    float sl[4][6];
    int ref[4];
    int k = 0;
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int c = side ? right : left;
        if (c < 0) {
#pragma unroll
            for (int a = 0; a < 6; a++) sl[k][a] = box[side][a];
            ref[k++] = c;
        } else {
            int cl, cr, clo, cg, chi;
            node_ranges(topo, c, cl, cr, clo, cg, chi);
            range_box(T, clo, cg, sl[k]); ref[k++] = cl;
            range_box(T, cg + 1, chi, sl[k]); ref[k++] = cr;
        }
    }
    for (; k < 4; k++) {
#pragma unroll
        for (int a = 0; a < 6; a++) sl[k][a] = 1.0e30f;
        ref[k] = ENVGS_WIDE_EMPTY;
    }
    float4 *o = reinterpret_cast<float4 *>(nodes4 + (size_t)i * 32);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        o[2 * q] = make_float4(sl[q][0], sl[q][3], sl[q][1], sl[q][4]);
        o[2 * q + 1] = make_float4(sl[q][2], sl[q][5], __int_as_float(ref[q]), 0.f);
    }
}

// Tree quality for the build-or-refit decision of the drop-in module (SurfelTracer.build_acceleration_structure): the surface-area sum of every
// binary node's two child boxes -- the SAH cost of the topology under the CURRENT boxes, up to constants -- next to the root's own area.  A refit
// keeps the topology of positions that have since moved: the ratio of this sum (over the root area) to its value right after the full build says
// how much more box area a ray now has to wade through.  out[0] += sum (one float atomic per workgroup: a heuristic, order does not matter),
// out[1] = root area.
__global__ void __launch_bounds__(256)
tree_area_sum(int P, const float *__restrict__ nodes, float *out)
{
    __shared__ float s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a = 0.f;
    if (i < P - 1) {
        const float4 *n4 = reinterpret_cast<const float4 *>(nodes + (size_t)i * NODE);
        const float4 q0 = n4[0], q1 = n4[1], q2 = n4[2];            // left (lo.xyz hi.xyz), right (lo.xyz hi.xyz)
        const float lx = fmaxf(q0.w - q0.x, 0.f), ly = fmaxf(q1.x - q0.y, 0.f), lz = fmaxf(q1.y - q0.z, 0.f);
        const float rx = fmaxf(q2.y - q1.z, 0.f), ry = fmaxf(q2.z - q1.w, 0.f), rz = fmaxf(q2.w - q2.x, 0.f);
        a = (lx * ly + ly * lz + lz * lx) + (rx * ry + ry * rz + rz * rx);
        if (i == 0) {
            const float ex = fmaxf(q0.w, q2.y) - fminf(q0.x, q1.z), ey = fmaxf(q1.x, q2.z) - fminf(q0.y, q1.w), ez = fmaxf(q1.y, q2.w) - fminf(q0.z, q2.x);
            out[1] = ex * ey + ey * ez + ez * ex;
        }
        if (!(a < 1.0e30f)) a = 0.f;                                 // (a box of far-away sentinel points, NaN: not part of the measure)
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_w[0] + s_w[1] + s_w[2] + s_w[3]);
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

// P == 1: the wide node of the single-surfel tree: slot 0 = the surfel, slot 1 = the unreachable far point, two unused slots
__global__ void single_leaf_wide(const float *__restrict__ nodes, float *__restrict__ nodes4)
{
    if (threadIdx.x == 0) {
        for (int q = 0; q < 4; q++) {
            float *o = nodes4 + q * 8;
            for (int a = 0; a < 3; a++) { o[2 * a] = q < 2 ? nodes[q * 6 + a] : 1.0e30f; o[2 * a + 1] = q < 2 ? nodes[q * 6 + 3 + a] : 1.0e30f; }
            o[6] = q < 2 ? nodes[12 + q] : __int_as_float(ENVGS_WIDE_EMPTY);
            o[7] = 0.f;
        }
    }
}

}  // namespace envgs

using namespace envgs;

extern "C" {

size_t envgs_bvh_temp_bytes(int32_t P) { return carve(P, nullptr).total; }

// binary nodes (16 floats each), 4-wide nodes (32 floats each), then the sorted leaf order (P ints): topology + order = what a refit reads
size_t envgs_bvh_node_floats(int32_t P) { return (size_t)(NODE + 32) * (size_t)(P > 1 ? P - 1 : 1) + (size_t)(P > 0 ? P : 0); }

static int fit_from_table(int P, const BvhTemp &t, const float *topo, float *nodes, const envgs_raster_cfg *cfg, hipStream_t stream)
{
    hipLaunchKernelGGL(st_table<2>, dim3((t.nblk + 255) / 256), dim3(256), 0, stream, t.nblk, (const uint64_t *)nullptr, (const float *)(t.st + (size_t)ST_LOW * P * 6),
                       (const int *)nullptr, (const float *)nullptr, (const float *)nullptr, t.bt, P);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(super_table, dim3(1), dim3(1024), 0, stream, t.nblk, t.nsup, t.mlev, (const float *)(t.bt + (size_t)ST_LOW * t.nblk * 6), t.sup);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    Tables T; T.st = t.st; T.bt = t.bt; T.sup = t.sup; T.P = P; T.nblk = t.nblk; T.nsup = t.nsup;
    hipLaunchKernelGGL(fit_nodes, dim3((P - 1 + 255) / 256), dim3(256), 0, stream, P, T, topo, nodes, nodes + (size_t)(P - 1) * NODE);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return 0;
}

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
        hipLaunchKernelGGL(single_leaf_wide, dim3(1), dim3(64), 0, stream, (const float *)nodes, nodes + NODE);
        ENVGS_CHECK_LAUNCH(cfg, stream);
        if (hipMemsetAsync(nodes + NODE + 32, 0, sizeof(int), stream) != hipSuccess) return ENVGS_ERR_BAD_ARG;      // order = {0}
        return 0;
    }
    hipLaunchKernelGGL(bounds_final, dim3(1), dim3(256), 0, stream, P, nblocks, t.partial, t.bounds);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(morton_keys, dim3(nblocks), dim3(256), 0, stream, P, t.leaf_box, t.bounds, t.keys_in);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    // Morton code << 32 | surfel id, ascending: buckets by the code's top bits, one LDS sort per bucket (raster_bin.hip: launch_key_sort)
    const int rc_sort = launch_key_sort(P, t.keys_in, t.keys_out, 62, t.sort_temp, t.sort_bytes, stream);
    if (rc_sort) return rc_sort;
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(st_table<0>, dim3(nblocks), dim3(256), 0, stream, P, (const uint64_t *)t.keys_out, (const float *)t.leaf_box, (const int *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, t.st, 0);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    hipLaunchKernelGGL(build_hierarchy, dim3((P - 1 + 255) / 256), dim3(256), 0, stream, P, t.keys_out, nodes,
                       reinterpret_cast<int *>(nodes + (size_t)(P - 1) * (NODE + 32)));
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return fit_from_table(P, t, (const float *)nodes, nodes, cfg, stream);
}

int envgs_bvh_refit(int32_t P, const float *vertices, const float *opacities, const float *nodes_prev, float *nodes, void *temp, size_t temp_bytes,
                    int32_t debug, void *stream_)
{
    if (P < 0) return ENVGS_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (!vertices || !nodes || !nodes_prev || !temp) return ENVGS_ERR_BAD_ARG;
    if (P == 1) return envgs_bvh_build(P, vertices, opacities, nodes, temp, temp_bytes, debug, stream_);      // nothing to keep
    BvhTemp t = carve(P, temp);
    if (temp_bytes < t.total) return ENVGS_ERR_TEMP_TOO_SMALL;
    hipStream_t stream = (hipStream_t)stream_;
    envgs_raster_cfg dbg; dbg.debug = debug;
    const envgs_raster_cfg *cfg = &dbg;
    ProfScope prof_(K_BVH_BUILD, stream);
    hipLaunchKernelGGL(st_table<1>, dim3((P + 255) / 256), dim3(256), 0, stream, P, (const uint64_t *)nullptr, (const float *)nullptr,
                       reinterpret_cast<const int *>(nodes_prev + (size_t)(P - 1) * (NODE + 32)), vertices, opacities, t.st, 0);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return fit_from_table(P, t, nodes_prev, nodes, cfg, stream);
}

int envgs_bvh_quality(int32_t P, const float *nodes, float *out2, void *stream_)
{
    if (P < 0 || !out2) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (hipMemsetAsync(out2, 0, 2 * sizeof(float), stream) != hipSuccess) return ENVGS_ERR_BAD_ARG;
    if (P < 2) return 0;
    if (!nodes) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(tree_area_sum, dim3((P - 1 + 255) / 256), dim3(256), 0, stream, P, nodes, out2);
    return (int)hipGetLastError();
}

}  // extern "C"
