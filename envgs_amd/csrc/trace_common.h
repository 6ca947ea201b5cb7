// trace_common.h -- shared device code of the tracer (T2 forward, T3 backward; list path and K-buffer path): constants, the argument block,
// the ray / surfel math, the SH basis, the per-ray backward helpers, the batch scheduling helpers, and the declarations of every kernel
// (defined in trace_kbuffer.hip, trace_collect.hip, trace_lists.hip, trace_surfel_bwd.hip; launched from trace_api.hip).
//
// CDNA4 mapping of the K-buffer path: persistent wavefronts (one 64-lane workgroup each, grid = a few per CU) pull batches of 64 rays
// from a global counter; one lane = one ray.  Traversal keeps the per-lane node stack in LDS ([level][lane], so a
// push/pop is one conflict-free ds_write/ds_read_b32 per wavefront) and the K nearest accepted hits sorted in
// registers; a ray is composited in rounds of K hits, restarting the traversal from (t, id) of the last hit, until
// its transmittance drops below 1e-4 or the scene is exhausted.  A BVH node is one 64 B record holding both child
// boxes; a surfel is one 64 B record (centre, opacity, a/s_u, b/s_v, normal).  The backward re-traces in the
// identical order and uses the stored stage-0 sums for the suffix terms, so no hit list is ever written to HBM.
// The list path (what EnvGS runs) is described where its helpers start, below.
//
// Stands behind SurfelTracer.forward/backward (easyvolcap/utils/optix_utils.py:188-201); semantics are restated in
// oracle/surfel_trace_oracle.c ("parity unpinned": the OptiX sources are not in the reference tree).
#ifndef ENVGS_TRACE_COMMON_H
#define ENVGS_TRACE_COMMON_H

#include "common.h"
#include "prof.h"

#include <cstdlib>
#include <cstring>

#include "../../include/envgs_trace.h"

namespace envgs {

constexpr int KBUF = 16;            // hits buffered per round
constexpr int STACK = 64;           // LBVH depth bound: 62-bit keys
constexpr int NCOPY = 8;            // per-surfel hit counters are replicated NCOPY x (by ray index) to spread same-address atomics
constexpr int LDS_STACK = 24;       // collect_hits keeps this many levels in LDS, the rest in an HBM slab
constexpr int MAX_ROUNDS = 256;     // safety bound: 4096 hits per ray
constexpr float UV_MAX = 3.0f;
constexpr int MID = ENVGS_MID_CHANNELS;
constexpr int SREC = ENVGS_SREC_STRIDE;
constexpr int NODE = ENVGS_NODE_STRIDE;

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
static __device__ __constant__ float tC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f};
static __device__ __constant__ float tC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f};

struct SurfHit { float t, u, v, G, alpha, denom; bool ok; };

__device__ __forceinline__ SurfHit hit_surfel(const float4 s0, const float4 s1, const float4 s2, const float4 s3,
                                              const float ox, const float oy, const float oz, const float dx,
                                              const float dy, const float dz)
{
    SurfHit h;
    {
        // t is the SORT KEY of the per-ray hit lists (index work: bit-exact against the oracle, like the rasterizer's depth keys): fixed
        // evaluation order, no FMA contraction, IEEE division.  The normal in the surfel record is built the same way (make_surfel_records).
#pragma clang fp contract(off)
        h.denom = s3.x * dx + s3.y * dy + s3.z * dz;
        const float num = s3.x * (s0.x - ox) + s3.y * (s0.y - oy) + s3.z * (s0.z - oz);
        h.t = num / h.denom;
    }
    const float qx = ox + h.t * dx - s0.x, qy = oy + h.t * dy - s0.y, qz = oz + h.t * dz - s0.z;
    h.u = s1.x * qx + s1.y * qy + s1.z * qz;
    h.v = s2.x * qx + s2.y * qy + s2.z * qz;
    h.G = __expf(-0.5f * (h.u * h.u + h.v * h.v));
    const float a = s0.w * h.G;
    h.alpha = a < ALPHA_CAP ? a : ALPHA_CAP;
    h.ok = (h.denom != 0.0f) && (fabsf(h.u) <= UV_MAX) && (fabsf(h.v) <= UV_MAX) && (h.alpha >= ALPHA_MIN);
    return h;
}

// The same hit at a KNOWN distance: the sort / composite pass holds t as the high word of the (t, id) key the collection wrote -- computed there by
// hit_surfel's own expression -- so the IEEE division and the numerator's dot product are not repeated (round 5: ~15 VALU per 64-hit chunk less).
__device__ __forceinline__ SurfHit hit_surfel_at(const float4 s0, const float4 s1, const float4 s2, const float4 s3, const float t,
                                                 const float ox, const float oy, const float oz, const float dx, const float dy, const float dz)
{
    SurfHit h;
    {
#pragma clang fp contract(off)
        h.denom = s3.x * dx + s3.y * dy + s3.z * dz;
    }
    h.t = t;
    const float qx = ox + h.t * dx - s0.x, qy = oy + h.t * dy - s0.y, qz = oz + h.t * dz - s0.z;
    h.u = s1.x * qx + s1.y * qy + s1.z * qz;
    h.v = s2.x * qx + s2.y * qy + s2.z * qz;
    h.G = __expf(-0.5f * (h.u * h.u + h.v * h.v));
    const float a = s0.w * h.G;
    h.alpha = a < ALPHA_CAP ? a : ALPHA_CAP;
    h.ok = true;
    return h;
}

__device__ __forceinline__ void sh_basis(int D, float x, float y, float z, float *b)
{
    b[0] = kC0;
    if (D > 0) {
        b[1] = -kC1 * y; b[2] = kC1 * z; b[3] = -kC1 * x;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = tC2[0] * xy; b[5] = tC2[1] * yz; b[6] = tC2[2] * (2.0f * zz - xx - yy); b[7] = tC2[3] * xz; b[8] = tC2[4] * (xx - yy);
            if (D > 2) {
                b[9] = tC3[0] * y * (3.0f * xx - yy); b[10] = tC3[1] * xy * z; b[11] = tC3[2] * y * (4.0f * zz - xx - yy);
                b[12] = tC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); b[13] = tC3[4] * x * (4.0f * zz - xx - yy);
                b[14] = tC3[5] * z * (xx - yy); b[15] = tC3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

__device__ __forceinline__ void sh_basis_grad(int D, float x, float y, float z, float *gx, float *gy, float *gz)
{
#pragma unroll
    for (int k = 0; k < 16; k++) { gx[k] = 0.f; gy[k] = 0.f; gz[k] = 0.f; }
    if (D > 0) {
        gy[1] = -kC1; gz[2] = kC1; gx[3] = -kC1;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            gx[4] = tC2[0] * y; gy[4] = tC2[0] * x;
            gy[5] = tC2[1] * z; gz[5] = tC2[1] * y;
            gx[6] = tC2[2] * -2.f * x; gy[6] = tC2[2] * -2.f * y; gz[6] = tC2[2] * 4.f * z;
            gx[7] = tC2[3] * z; gz[7] = tC2[3] * x;
            gx[8] = tC2[4] * 2.f * x; gy[8] = tC2[4] * -2.f * y;
            if (D > 2) {
                gx[9] = tC3[0] * 6.f * xy; gy[9] = tC3[0] * 3.f * (xx - yy);
                gx[10] = tC3[1] * yz; gy[10] = tC3[1] * xz; gz[10] = tC3[1] * xy;
                gx[11] = tC3[2] * -2.f * xy; gy[11] = tC3[2] * (4.f * zz - xx - 3.f * yy); gz[11] = tC3[2] * 8.f * yz;
                gx[12] = tC3[3] * -6.f * xz; gy[12] = tC3[3] * -6.f * yz; gz[12] = tC3[3] * 3.f * (2.f * zz - xx - yy);
                gx[13] = tC3[4] * (4.f * zz - 3.f * xx - yy); gy[13] = tC3[4] * -2.f * xy; gz[13] = tC3[4] * 8.f * xz;
                gx[14] = tC3[5] * 2.f * xz; gy[14] = tC3[5] * -2.f * yz; gz[14] = tC3[5] * (xx - yy);
                gx[15] = tC3[6] * 3.f * (xx - yy); gy[15] = tC3[6] * -6.f * xy;
            }
        }
    }
}

// t_min of the first stage: camera rays skip the near 0.2 (the rasterizer's near plane), reflected rays start at 0, and the secondary rays of a
// bounce traced as a call of their own start just off the surface they left (1e-3, what the in-kernel bounce stages use)
__device__ __forceinline__ float first_tmin(int start_from_first) { return start_from_first == 1 ? NEAR_N : (start_from_first == 2 ? 1.0e-3f : 0.0f); }

struct TraceArgs {
    int P, R, D, M, ND, start_from_first, has_others, bg_len;
    int f16;            // shs / colors are stored as IEEE half (converted on load; see Feat in common.h)
    float spec_thr;
    const float4 *nodes;
    const float4 *srec;
    const float *shs, *colors, *others, *bg;
    const float *ray_o, *ray_d;
    unsigned *counter;
    unsigned long long *stats;      // [hits, per-ray node visits, rounds, hits found, wide nodes fetched per packet, surfel records fetched per packet]
    // forward outputs
    float *rgb, *dpt, *acc, *norm, *dist, *aux, *mid, *wet, *final_T;
    // backward inputs / outputs
    const float *f_rgb, *f_dpt, *f_acc, *f_norm, *f_aux, *f_T;
    const float *g_rgb, *g_dpt, *g_acc, *g_norm, *g_aux;
    float *geo_rec, *dshs, *dcolors, *dothers, *dray_o, *dray_d;
    float mod;
    // per-ray hit lists (list path): entry = (t, surfel id), [R][cap]
    uint2 *hits;
    int *hit_cnt;       // hits found per ray (may exceed cap: the ray then takes the K-buffer path)
    int *n_used;        // hits composited before termination
    int cap;
    unsigned long long *surf_acc; // (P) packed per-surfel accumulator: low 24 bits hit count, high 40 bits fixed-point weight
    int wfrac;                // fractional bits of that fixed-point weight
    unsigned *surf_cnt;       // (P) composited hits per surfel (list path)
    const unsigned *surf_off; // (P) inclusive scan of surf_cnt
    float *records;           // (num_records, 24) per-hit gradient records grouped by surfel
    unsigned long long num_records;
    const unsigned *order;    // (R) ray permutation (coherence sort) or NULL
    int exp;            // diagnostic switches (envgs_debug_set(ENVGS_DBG_TRACE, ...); 0 in production): 8 = atomic-flush backward instead of records,
                        // 16 = binary packet traversal instead of the 4-wide one,
                        // 64 = no coherence sort of the rays, 512 = per-ray collection kernel even when the rays are sorted,
                        // 1024 = packet stack limited to 2 entries (tests the overflow hand-off)
    int *stack_spill;   // collect_hits: (grid, STACK, 64) ints of stack overflow space
    int only_overflow;  // K-buffer kernels: process only rays whose hit_cnt exceeds cap
    int batch0, batch1; // list-path forward kernels: the range of 64-ray batches this launch owns (segments run on two streams)
    uint4 *sparse;      // (sparse_cap) hits of SPARSE entries (envgs_trace.h: sparse_hits): {sorted ray slot, list position, surfel id, record slot}; counter[64] = how many
    unsigned sparse_cap;
    int sparse_max;     // an entry with at most this many hits is filed per hit instead of becoming an entry of the batch kernel (0 = off)
    int reduce_adds;    // reduce_surfel_records adds its sums to what the buffers hold instead of storing them (the backward's deferred tail: the K-buffer pass ran before it)
    int seg;            // segment index: selects the batch-fetch counters and the stack-spill slab
    int spill_stride;   // stack-spill slabs per segment
    unsigned long long *entries;  // (batches, 64*cap) distinct (batch, surfel) entries, see register_hits
    unsigned *pairs;              // (batches, 64*cap) (lane << 16 | k) of every composited hit, grouped by entry
    int *n_entries;               // (batches, 2) table entries, single entries
    unsigned *long_list; // (R) scratch: slots of the rays whose lists exceed 256 hits, appended by the main sort pass (counter[24 + seg] = how many)
    float4 *state;      // per composited hit, for the backward, as two PLANES: plane 0 (16 B rows, row i at state[i]) = (transmittance before the hit, rgb
                        // prefix sums after it); plane 1 (behind the state_plane rows of plane 0; state_row1) = (depth, normal prefix sums) in 16 B rows, or
                        // -- with `others` -- (depth, normal, the two aux sums) in 24 B rows (round 6: the aux sums used to be a third plane of 8 B rows, a
                        // third gather per hit in the generic backward).  A backward whose only upstream gradient is the colour's -- what EnvGS trains with -- reads plane 0 alone
    size_t state_plane; // rows per plane (compact_rows, or R * cap)
    int colour_state;   // 1: only plane 0 exists (envgs_trace_lists::state_planes == 1: the backward will be the colour-only one)
    // compact per-hit buffers (envgs_trace.h: compact_rows): nullptr = the (R, cap) layouts
    unsigned *batch_cnt;          // (batches) written by the cooperative collection: rows the batch needs (sum of its listed rays' hit counts)
    const unsigned *row_off;      // (R) by sorted slot: the ray's first row of hit_state
    const uint2 *batch_rows;      // (batches) {first row, rows} of the batch in entries / pairs
    const void *shp;    // (P, 48) quad-permuted copy of the SH blocks (permute_sh), same storage type as shs; nullptr = per-lane gathers from shs
};

// K-nearest buffer ordered by (t, id); insertion is a fully unrolled compare-exchange chain (registers only).
struct KBuf {
    float t[KBUF];
    int id[KBUF];
    int n;
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < KBUF; i++) { t[i] = 3.0e38f; id[i] = 0x7fffffff; }
        n = 0;
    }
    __device__ __forceinline__ void insert(float ct, int cid) {
#pragma unroll
        for (int i = 0; i < KBUF; i++) {
            const bool before = (ct < t[i]) || (ct == t[i] && cid < id[i]);
            const float tt = before ? t[i] : ct; const int ii = before ? id[i] : cid;
            t[i] = before ? ct : t[i]; id[i] = before ? cid : id[i];
            ct = tt; cid = ii;
        }
        n = n < KBUF ? n + 1 : KBUF;
    }
};

// One traversal round: collect the K nearest accepted hits with (t,id) > (tlo,idlo).
__device__ __forceinline__ void traverse(const TraceArgs &A, int (*stk)[64], const int lane, const bool active,
                                         const float ox, const float oy, const float oz, const float dx, const float dy,
                                         const float dz, const float tlo, const int idlo, KBuf &kb, unsigned &visits)
{
    kb.reset();
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    int sp = 0;
    int cur = active ? 0 : -1;                  // node 0 is the root; -1 = nothing to do
    float tmax = 3.0e38f;
    while (true) {
        if (cur < 0) {
            if (sp == 0) break;
            cur = stk[--sp][lane];
        }
        const float4 *nd = A.nodes + (size_t)cur * 4;
        visits++;
        const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
        const int lc = __float_as_int(n3.x), rc = __float_as_int(n3.y);
        // slabs: left box min (n0.x,n0.y,n0.z) max (n0.w,n1.x,n1.y); right box min (n1.z,n1.w,n2.x) max (n2.y,n2.z,n2.w)
        float a0 = (n0.x - ox) * ix, a1 = (n0.w - ox) * ix, b0 = (n0.y - oy) * iy, b1 = (n1.x - oy) * iy, c0 = (n0.z - oz) * iz, c1 = (n1.y - oz) * iz;
        float tnL = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
        float tfL = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
        a0 = (n1.z - ox) * ix; a1 = (n2.y - ox) * ix; b0 = (n1.w - oy) * iy; b1 = (n2.z - oy) * iy; c0 = (n2.x - oz) * iz; c1 = (n2.w - oz) * iz;
        float tnR = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
        float tfR = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
        bool hitL = (tnL <= tfL) && (tfL >= tlo) && (tnL <= tmax);
        bool hitR = (tnR <= tfR) && (tfR >= tlo) && (tnR <= tmax);
#pragma unroll
        for (int side = 0; side < 2; side++) {
            const bool hit = side == 0 ? hitL : hitR;
            const int ch = side == 0 ? lc : rc;
            if (hit && ch < 0) {
                const int sid = ~ch;
                const float4 *sr = A.srec + (size_t)sid * 4;
                const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], sr[3], ox, oy, oz, dx, dy, dz);
                const bool after = (h.t > tlo) || (h.t == tlo && sid > idlo);
                const bool fits = (kb.n < KBUF) || (h.t < kb.t[KBUF - 1]) || (h.t == kb.t[KBUF - 1] && sid < kb.id[KBUF - 1]);
                if (h.ok && after && fits) {
                    kb.insert(h.t, sid);
                    if (kb.n == KBUF) tmax = kb.t[KBUF - 1];
                }
            }
        }
        hitL = hitL && lc >= 0;
        hitR = hitR && rc >= 0;
        if (hitL && hitR) {
            const bool leftFirst = tnL <= tnR;
            stk[sp++][lane] = leftFirst ? rc : lc;
            cur = leftFirst ? lc : rc;
        } else if (hitL) cur = lc;
        else if (hitR) cur = rc;
        else cur = -1;
    }
}

struct StageSums { float rgb[3], dpt, acc, nrm[3], dist, aux[2], T, M1, M2; };

// 8 halves (one 16 B load) -> 8 floats
__device__ __forceinline__ void unpack_h8(const uint4 x, float *v)
{
    const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        v[2 * q] = __half2float(__ushort_as_half((unsigned short)(w[q] & 0xFFFFu)));
        v[2 * q + 1] = __half2float(__ushort_as_half((unsigned short)(w[q] >> 16)));
    }
}

// SH block of one surfel into registers (zeros beyond the active degree).  fp16 storage: 96 B per surfel = six 16 B loads instead of twelve.
__device__ __forceinline__ void load_sh(const TraceArgs &A, const int sid, const int nb, float *v)
{
    if (A.M == 16 && A.f16) {
        const uint4 *s8 = reinterpret_cast<const uint4 *>(reinterpret_cast<const __half *>(A.shs) + (size_t)sid * 48);
        const int nq = (nb * 3 + 7) >> 3;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            uint4 x = make_uint4(0u, 0u, 0u, 0u);
            if (q < nq) x = s8[q];
            unpack_h8(x, v + 8 * q);
        }
#pragma unroll
        for (int k = 0; k < 48; k++) if (k >= nb * 3) v[k] = 0.f;
    } else if (A.M == 16) {
        const float4 *s4 = reinterpret_cast<const float4 *>(A.shs + (size_t)sid * 48);
        const int nq = (nb * 3 + 3) >> 2;
#pragma unroll
        for (int q = 0; q < 12; q++) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < nq) x = s4[q];
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
        const Feat sh = Feat{A.shs, A.f16 != 0}.at((size_t)sid * A.M * 3);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const bool in = k < nb;
            v[k * 3] = in ? sh[k * 3] : 0.f; v[k * 3 + 1] = in ? sh[k * 3 + 1] : 0.f; v[k * 3 + 2] = in ? sh[k * 3 + 2] : 0.f;
        }
    }
}

__device__ __forceinline__ void surfel_color(const TraceArgs &A, int sid, const float *basis, float *col, bool *cl)
{
    if (A.M > 0) {
        const int nb = (A.D + 1) * (A.D + 1);
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        // the usual layout (16 coefficients x RGB, 16 B aligned): 12 x 16 B loads (6 with fp16 storage) instead of 48 x 4 B gathers
        float v[48];
        load_sh(A, sid, nb, v);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < nb) { const float b = basis[k]; r0 += b * v[k * 3]; r1 += b * v[k * 3 + 1]; r2 += b * v[k * 3 + 2]; }
        r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
        cl[0] = r0 < 0.f; cl[1] = r1 < 0.f; cl[2] = r2 < 0.f;
        col[0] = cl[0] ? 0.f : r0; col[1] = cl[1] ? 0.f : r1; col[2] = cl[2] ? 0.f : r2;
    } else {
        const Feat c = Feat{A.colors, A.f16 != 0}.at((size_t)sid * 3);
        col[0] = c[0]; col[1] = c[1]; col[2] = c[2];
        cl[0] = cl[1] = cl[2] = false;
    }
}

__device__ __forceinline__ int ray_index(int slot, int R, int rh, int rw)
{
    // 64 consecutive slots = one 8x8 pixel block when the ray tensor is an (H,W) image with H,W % 8 == 0
    if (rh > 0 && (rh & 7) == 0 && (rw & 7) == 0) {
        const int blk = slot >> 6, in = slot & 63;
        const int bw = rw >> 3;
        const int by = blk / bw, bx = blk - by * bw;
        return (by * 8 + (in >> 3)) * rw + bx * 8 + (in & 7);
    }
    return slot;
}

// ------------------------------------------------------------------------------------------ T3 ---
// SH basis k as (l0 + l1 x + l2 y + l3 z) * (q0 + q1 xx + q2 yy + q3 zz + q4 xy + q5 yz + q6 xz): lets lane j evaluate
// "its" basis function (k = j / 3) of ANOTHER lane's ray direction during the cooperative gradient flush.
static __device__ __constant__ float kShForm[16][11] = {
    {1, 0, 0, 0, 0.28209479177387814f, 0, 0, 0, 0, 0, 0},
    {0, 0, 1, 0, -0.4886025119029199f, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 1, 0.4886025119029199f, 0, 0, 0, 0, 0, 0},
    {0, 1, 0, 0, -0.4886025119029199f, 0, 0, 0, 0, 0, 0},
    {1, 0, 0, 0, 0, 0, 0, 0, 1.0925484305920792f, 0, 0},
    {1, 0, 0, 0, 0, 0, 0, 0, 0, -1.0925484305920792f, 0},
    {1, 0, 0, 0, 0, -0.31539156525252005f, -0.31539156525252005f, 2.f * 0.31539156525252005f, 0, 0, 0},
    {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1.0925484305920792f},
    {1, 0, 0, 0, 0, 0.5462742152960396f, -0.5462742152960396f, 0, 0, 0, 0},
    {0, 0, 1, 0, 0, 3.f * -0.5900435899266435f, 0.5900435899266435f, 0, 0, 0, 0},
    {0, 0, 0, 1, 0, 0, 0, 0, 2.890611442640554f, 0, 0},
    {0, 0, 1, 0, 0, 0.4570457994644658f, 0.4570457994644658f, 4.f * -0.4570457994644658f, 0, 0, 0},
    {0, 0, 0, 1, 0, -3.f * 0.3731763325901154f, -3.f * 0.3731763325901154f, 2.f * 0.3731763325901154f, 0, 0, 0},
    {0, 1, 0, 0, 0, 0.4570457994644658f, 0.4570457994644658f, 4.f * -0.4570457994644658f, 0, 0, 0},
    {0, 0, 0, 1, 0, 1.445305721320277f, -1.445305721320277f, 0, 0, 0, 0},
    {0, 1, 0, 0, 0, -0.5900435899266435f, 3.f * 0.5900435899266435f, 0, 0, 0, 0},
};

constexpr int NFLD = 22;   // LDS hand-off fields per lane: sid, dc[3], geo[15], dir[3]
constexpr int GEO = ENVGS_GEOREC_STRIDE;

// Per-ray constants of the backward pass (upstream gradients and the stored stage-0 sums).
struct BwdRay {
    float ox, oy, oz, dx, dy, dz;
    float gR0, gR1, gR2, gD, gA, gN0, gN1, gN2, gX0, gX1;
    float fT, bgdot, fr0, fr1, fr2, fD, fA, fN0, fN1, fN2, fX0, fX1;
    float dl2, il, ux, uy, uz;
};
// Running prefix sums and the ray-gradient accumulators.
struct BwdAcc {
    float T, c0, c1, c2, cD, cA, cN0, cN1, cN2, cX0, cX1;
    float dO0, dO1, dO2, dD0, dD1, dD2;
    float Sk[16];
};

__device__ __forceinline__ void bwd_load_ray(const TraceArgs &A, int r, BwdRay &B)
{
    B.ox = A.ray_o[3 * r]; B.oy = A.ray_o[3 * r + 1]; B.oz = A.ray_o[3 * r + 2];
    B.dx = A.ray_d[3 * r]; B.dy = A.ray_d[3 * r + 1]; B.dz = A.ray_d[3 * r + 2];
    // an upstream gradient the caller did not pass (NULL) is zero: no buffer of zeros has to be made for outputs the loss does not use
    B.gR0 = B.gR1 = B.gR2 = B.gD = B.gA = B.gN0 = B.gN1 = B.gN2 = B.gX0 = B.gX1 = 0.f;
    if (A.g_rgb) { B.gR0 = A.g_rgb[3 * r]; B.gR1 = A.g_rgb[3 * r + 1]; B.gR2 = A.g_rgb[3 * r + 2]; }
    if (A.g_dpt) B.gD = A.g_dpt[r];
    if (A.g_acc) B.gA = A.g_acc[r];
    if (A.g_norm) { B.gN0 = A.g_norm[3 * r]; B.gN1 = A.g_norm[3 * r + 1]; B.gN2 = A.g_norm[3 * r + 2]; }
    if (A.g_aux) { B.gX0 = A.g_aux[2 * r]; B.gX1 = A.g_aux[2 * r + 1]; }
    B.fT = A.f_T[r];
    const float bg0 = 0 < A.bg_len ? A.bg[0] : 0.f, bg1 = 1 < A.bg_len ? A.bg[1] : 0.f, bg2 = 2 < A.bg_len ? A.bg[2] : 0.f;
    B.bgdot = bg0 * B.gR0 + bg1 * B.gR1 + bg2 * B.gR2;
    // final sums without the background term (suffix = final - prefix)
    B.fr0 = A.f_rgb[3 * r] - B.fT * bg0; B.fr1 = A.f_rgb[3 * r + 1] - B.fT * bg1; B.fr2 = A.f_rgb[3 * r + 2] - B.fT * bg2;
    B.fD = A.f_dpt[r]; B.fA = A.f_acc[r];
    B.fN0 = A.f_norm[3 * r]; B.fN1 = A.f_norm[3 * r + 1]; B.fN2 = A.f_norm[3 * r + 2];
    B.fX0 = A.f_aux[2 * r]; B.fX1 = A.f_aux[2 * r + 1];
    B.dl2 = B.dx * B.dx + B.dy * B.dy + B.dz * B.dz; B.il = 1.0f / sqrtf(B.dl2);
    B.ux = B.dx * B.il; B.uy = B.dy * B.il; B.uz = B.dz * B.il;
}

__device__ __forceinline__ void bwd_init_acc(BwdAcc &a)
{
    a.T = 1.0f; a.c0 = a.c1 = a.c2 = a.cD = a.cA = a.cN0 = a.cN1 = a.cN2 = a.cX0 = a.cX1 = 0.f;
    a.dO0 = a.dO1 = a.dO2 = a.dD0 = a.dD1 = a.dD2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) a.Sk[k] = 0.f;
}

// Gradient of one composited hit.  Returns false when the ray terminates at this hit (it is then NOT blended).
// Out: dc[3] (dL/dcolour of the surfel from this hit) and gv[15] (the geometry-record words).
__device__ __forceinline__ bool bwd_hit(const TraceArgs &A, const BwdRay &B, BwdAcc &a, const float *basis, const int nb,
                                        const int sid, float &dc0, float &dc1, float &dc2, float *gv)
{
    const float4 *sr = A.srec + (size_t)sid * 4;
    const float4 s0 = sr[0], s1 = sr[1], s2 = sr[2], s3 = sr[3];
    const SurfHit h = hit_surfel(s0, s1, s2, s3, B.ox, B.oy, B.oz, B.dx, B.dy, B.dz);
    const float alpha = h.alpha;
    const float test_T = a.T * (1.0f - alpha);
    if (test_T < T_EPS) return false;
    const float T = a.T;
    const float w = alpha * T;
    float col[3]; bool cl[3];
    float shv[48];
    if (A.M > 0) {
        load_sh(A, sid, nb, shv);
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < nb) { const float b = basis[k]; r0 += b * shv[k * 3]; r1 += b * shv[k * 3 + 1]; r2 += b * shv[k * 3 + 2]; }
        r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
        cl[0] = r0 < 0.f; cl[1] = r1 < 0.f; cl[2] = r2 < 0.f;
        col[0] = cl[0] ? 0.f : r0; col[1] = cl[1] ? 0.f : r1; col[2] = cl[2] ? 0.f : r2;
    } else surfel_color(A, sid, basis, col, cl);
    const float sgn = h.denom < 0.0f ? 1.0f : -1.0f;
    const float nf0 = sgn * s3.x, nf1 = sgn * s3.y, nf2 = sgn * s3.z;
    const float x0 = A.has_others ? A.others[2 * sid] : 0.f, x1 = A.has_others ? A.others[2 * sid + 1] : 0.f;
    a.c0 += w * col[0]; a.c1 += w * col[1]; a.c2 += w * col[2];
    a.cD += w * h.t; a.cA += w;
    a.cN0 += w * nf0; a.cN1 += w * nf1; a.cN2 += w * nf2;
    a.cX0 += w * x0; a.cX1 += w * x1;
    const float inv1m = 1.0f / (1.0f - alpha);
    float dLa = B.gR0 * (T * col[0] - (B.fr0 - a.c0) * inv1m) + B.gR1 * (T * col[1] - (B.fr1 - a.c1) * inv1m) + B.gR2 * (T * col[2] - (B.fr2 - a.c2) * inv1m);
    dLa += B.gD * (T * h.t - (B.fD - a.cD) * inv1m);
    dLa += B.gA * (T - (B.fA - a.cA) * inv1m);
    dLa += B.gN0 * (T * nf0 - (B.fN0 - a.cN0) * inv1m) + B.gN1 * (T * nf1 - (B.fN1 - a.cN1) * inv1m) + B.gN2 * (T * nf2 - (B.fN2 - a.cN2) * inv1m);
    dLa += B.gX0 * (T * x0 - (B.fX0 - a.cX0) * inv1m) + B.gX1 * (T * x1 - (B.fX1 - a.cX1) * inv1m);
    dLa += -(B.fT * inv1m) * B.bgdot;
    dc0 = cl[0] ? 0.f : w * B.gR0; dc1 = cl[1] ? 0.f : w * B.gR1; dc2 = cl[2] ? 0.f : w * B.gR2;
    if (A.M > 0) {
        // dL/d(dir) = sum_k grad(basis_k) * (sh_k . dc): accumulate the 16 scalars, apply grad(basis) once per ray
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < nb) a.Sk[k] += shv[k * 3] * dc0 + shv[k * 3 + 1] * dc1 + shv[k * 3 + 2] * dc2;
    }
    if (A.has_others && A.dothers) { atomic_add_f32(A.dothers + 2 * sid, w * B.gX0); atomic_add_f32(A.dothers + 2 * sid + 1, w * B.gX1); }
    const float dLG = s0.w * dLa;
    const float dLu = dLG * (-h.G * h.u), dLv = dLG * (-h.G * h.v);
    const float su = s1.w, sv = s2.w;
    const float qx = B.ox + h.t * B.dx - s0.x, qy = B.oy + h.t * B.dy - s0.y, qz = B.oz + h.t * B.dz - s0.z;
    // u = (a/su).q : dL/dq = dLu*(a/su) + dLv*(b/sv) ; dL/da = (dLu/su) q ; dL/dsu = -dLu*u/su
    const float dq0 = dLu * s1.x + dLv * s2.x, dq1 = dLu * s1.y + dLv * s2.y, dq2 = dLu * s1.z + dLv * s2.z;
    const float cu = dLu / su, cv = dLv / sv;
    const float dLt_tot = w * B.gD + dq0 * B.dx + dq1 * B.dy + dq2 * B.dz;
    const float kt = dLt_tot / h.denom;
    gv[0] = -dq0 + kt * s3.x; gv[1] = -dq1 + kt * s3.y; gv[2] = -dq2 + kt * s3.z;
    gv[3] = cu * qx; gv[4] = cu * qy; gv[5] = cu * qz;
    gv[6] = cv * qx; gv[7] = cv * qy; gv[8] = cv * qz;
    gv[9] = w * sgn * B.gN0 - kt * qx; gv[10] = w * sgn * B.gN1 - kt * qy; gv[11] = w * sgn * B.gN2 - kt * qz;
    gv[12] = -dLu * h.u / su * A.mod; gv[13] = -dLv * h.v / sv * A.mod;
    gv[14] = h.G * dLa;
    a.dO0 += dq0 - kt * s3.x; a.dO1 += dq1 - kt * s3.y; a.dO2 += dq2 - kt * s3.z;
    a.dD0 += h.t * (dq0 - kt * s3.x); a.dD1 += h.t * (dq1 - kt * s3.y); a.dD2 += h.t * (dq2 - kt * s3.z);
    a.T = test_T;
    return true;
}

// Cooperative flush: one hit at a time, the WHOLE wavefront writes that surfel's contiguous gradient words:
// lanes 0..47 the (16,3) SH block, lanes 48..62 the 15-word geometry record -> 1 instruction, ~3 cache lines per hit
// (instead of 63 per-lane atomics that each touch 64 different lines).  Must be called wave-uniformly.
struct FlushRole { float form[11]; int fc; bool sh_lane, geo_lane; };

__device__ __forceinline__ FlushRole flush_role(const TraceArgs &A, int lane)
{
    FlushRole R;
    const int fk = lane / 3;
    R.fc = lane - 3 * fk;
#pragma unroll
    for (int i = 0; i < 11; i++) R.form[i] = kShForm[fk < 16 ? fk : 0][i];
    const int nb = (A.D + 1) * (A.D + 1);
    R.sh_lane = A.M > 0 ? (lane < 48 && fk < nb) : (lane < 3);
    R.geo_lane = lane >= 48 && lane < 48 + 15;
    return R;
}

__device__ __forceinline__ void flush_hits(const TraceArgs &A, float (*fld)[65], const int lane, const FlushRole &R,
                                           const bool has, const int sid, const float dc0, const float dc1, const float dc2, const float *gv)
{
    const unsigned long long hm = __builtin_amdgcn_ballot_w64(has);
    if (hm == 0) return;
    fld[0][lane] = __int_as_float(sid); fld[1][lane] = dc0; fld[2][lane] = dc1; fld[3][lane] = dc2;
#pragma unroll
    for (int k = 0; k < 15; k++) fld[4 + k][lane] = gv[k];
    __syncthreads();
    unsigned long long m = hm;
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const int hs = __float_as_int(fld[0][l]);
        float val = 0.f; float *dst = nullptr;
        if (A.M > 0) {
            const float x = fld[19][l], y = fld[20][l], z = fld[21][l];
            const float lin = R.form[0] + R.form[1] * x + R.form[2] * y + R.form[3] * z;
            const float quad = R.form[4] + R.form[5] * (x * x) + R.form[6] * (y * y) + R.form[7] * (z * z) + R.form[8] * (x * y) + R.form[9] * (y * z) + R.form[10] * (x * z);
            const float dcc = R.fc == 0 ? fld[1][l] : (R.fc == 1 ? fld[2][l] : fld[3][l]);
            val = lin * quad * dcc;
            dst = A.dshs + (size_t)hs * A.M * 3 + lane;
        } else {
            val = fld[1 + (lane < 3 ? lane : 0)][l];
            dst = A.dcolors + (size_t)hs * 3 + lane;
        }
        if (R.geo_lane) { val = fld[4 + (lane - 48)][l]; dst = A.geo_rec + (size_t)hs * GEO + (lane - 48); }
        if (R.sh_lane || R.geo_lane) atomic_add_f32(dst, val);
    }
    __syncthreads();
}

__device__ __forceinline__ void bwd_store_ray(const TraceArgs &A, int r, const BwdRay &B, const BwdAcc &a)
{
    float dd0 = 0.f, dd1 = 0.f, dd2 = 0.f;
    if (A.M > 0) {
        float bgx[16], bgy[16], bgz[16];
        sh_basis_grad(A.D, B.ux, B.uy, B.uz, bgx, bgy, bgz);
#pragma unroll
        for (int k = 0; k < 16; k++) { dd0 += bgx[k] * a.Sk[k]; dd1 += bgy[k] * a.Sk[k]; dd2 += bgz[k] * a.Sk[k]; }
    }
    const float inv3 = B.il * B.il * B.il;
    const float e0 = a.dD0 + ((B.dl2 - B.dx * B.dx) * dd0 - B.dy * B.dx * dd1 - B.dz * B.dx * dd2) * inv3;
    const float e1 = a.dD1 + (-B.dx * B.dy * dd0 + (B.dl2 - B.dy * B.dy) * dd1 - B.dz * B.dy * dd2) * inv3;
    const float e2 = a.dD2 + (-B.dx * B.dz * dd0 - B.dy * B.dz * dd1 + (B.dl2 - B.dz * B.dz) * dd2) * inv3;
    A.dray_o[3 * r] = a.dO0; A.dray_o[3 * r + 1] = a.dO1; A.dray_o[3 * r + 2] = a.dO2;
    A.dray_d[3 * r] = e0; A.dray_d[3 * r + 1] = e1; A.dray_d[3 * r + 2] = e2;
}

// Rays are processed in a coherence-sorted order when A.order is set: 64 consecutive slots = one wavefront = rays with nearly the same
// direction (and nearby origins), so its lanes walk nearly the same BVH nodes and hit the same surfels -- the loads coalesce.
__device__ __forceinline__ int ray_of(const TraceArgs &A, int slot) { return slot < A.R ? (A.order ? (int)A.order[slot] : slot) : A.R; }

// ---------------------------------------------------------------------------------- list path ---
// MI355X-first variant of T2/T3 for bounce-free tracing (what EnvGS runs: max_trace_depth = 0).  HBM is plentiful
// (288 GB), so instead of re-traversing the BVH in rounds of K hits -- and again in the backward -- the ray's hits
// are collected ONCE, unordered, into a per-ray list in HBM (collect_hits: no K-buffer, few registers, high
// occupancy), sorted by (t, id) per ray in LDS by the whole wavefront (sort_hit_lists), and then walked front to back
// by the forward (composite_lists_fwd) and again by the backward (composite_lists_bwd), which never touches the BVH.
// Rays whose list overflows `cap` fall back to the K-buffer kernels above (only_overflow mode).

// XCD-affine batch fetch.  Rays are coherence-sorted, so a contiguous run of 64-ray batches covers one region of direction space;
// each of the 8 XCDs (private 4 MB L2) takes its own contiguous eighth of the batches, so the BVH nodes and surfel records that
// region touches stay in THAT L2 instead of streaming from the Infinity Cache for every XCD.  An XCD that runs dry steals.
// workgroup b of a grid whose size is a multiple of 8: workgroups that share an XCD (b % 8) get one contiguous run of block slots
__device__ __forceinline__ int xcd_block(int b, int nblocks) { return (b & 7) * (nblocks >> 3) + (b >> 3); }

__device__ __forceinline__ int xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return (int)(v & 7u);
}

__device__ __forceinline__ int fetch_batch(unsigned *ctr /*8 counters*/, int nbatch, int home, int lane)
{
    const int per = (nbatch + 7) >> 3;
    int b = -1;
    if (lane == 0) {
        for (int k = 0; k < 8 && b < 0; k++) {
            const int x = (home + k) & 7;
            const int lo = x * per, hi = min(lo + per, nbatch);
            if (lo >= hi) continue;
            const int i = (int)atomicAdd(ctr + x, 1u);
            if (lo + i < hi) b = lo + i;
        }
    }
    return __builtin_amdgcn_readfirstlane(b);
}

// Where a ray's per-hit state rows start, and a batch's region of entries / pairs: compact (row offsets from the scan of the hit counts in
// sorted order) or the (R, cap) / (batches, 64 cap) layouts.
__device__ __forceinline__ size_t state_row0(const TraceArgs &A, const int slot, const int r)
{
    return A.row_off ? (size_t)A.row_off[slot] : (size_t)r * A.cap;
}
// Plane 1 of the per-hit state, row i: 16 B rows behind plane 0's state_plane rows -- 24 B rows with `others` (TraceArgs::state)
__device__ __forceinline__ char *state_row1(const TraceArgs &A, const size_t i)
{
    return reinterpret_cast<char *>(A.state + A.state_plane) + i * (A.has_others ? (size_t)24 : (size_t)16);
}
__device__ __forceinline__ void batch_region(const TraceArgs &A, const int batch, size_t &start, size_t &size)
{
    if (A.batch_rows) { const uint2 br = A.batch_rows[batch]; start = br.x; size = br.y; }
    else { size = (size_t)64 * A.cap; start = (size_t)batch * size; }
}

// Conservative termination bound for the unordered collection.  The ray's accepted hits are binned by distance into 16
// linear bins over its chord through the scene box (16 registers of optical depth -ln(1-alpha)); as soon as the bins up to edge e hold more optical depth than
// the compositing can survive (T < 1e-4), every hit beyond e is provably after the terminating hit: it is dropped and BVH nodes
// that start beyond e are pruned.  Exact (never drops a composited hit) and it removes most of the 3x over-collection of a fog.
constexpr int NBIN = 16;
constexpr float KILL_OD = 9.2104f * 1.03f + 0.05f;       // -ln(1e-4) with margin for fp32 product vs sum-of-logs

// Wave-uniform stack of the packet kernels, in LDS.  Sized so that it does not overflow: the LBVH is at most 63 levels deep (62-bit unique Morton keys),
// the binary walk holds one postponed child per level and the 4-wide walk at most three per TWO levels.  Should a child ever not fit, the batch is
// flagged: its rays are handed to the K-buffer kernels (per-lane stacks) and counters[20] counts the event -- never a silently dropped subtree.
constexpr int PSTACK = 128;
constexpr int WIDE_EMPTY = ENVGS_WIDE_EMPTY;   // reference of an unused slot of a 4-wide node (never a surfel: ids are < 2^24 on the list path)
constexpr int SORT_MAX = 1024;  // longest list the sort / composite pass takes (16 keys per lane)
constexpr int RH_W = 8;         // register_hits: wavefronts per batch -- wave q takes list positions q, q + RH_W, ... of every ray

// ---- kernels (the launch bounds / occupancy attributes are repeated here: a declaration without them makes the compiler assume 1024-thread
//      workgroups, i.e. a 128-VGPR budget, for every other translation unit AND for the definition that follows it) ------------------------------
#ifndef ENVGS_BSB_WAVES
#define ENVGS_BSB_WAVES 2       // wavefronts per SIMD of batch_surfel_bwd (223 / 231 VGPRs): 3 needs 168 VGPRs = 41 / 52 spilled dwords -- measured, see DESIGN.md section 9
#endif
struct ForwardPrepare {
    ZeroBatch zero; int zero_blocks;                                                                      // 512 blocks per buffer to clear
    int P, rec_blocks; float mod; const float *means, *scales, *rots, *opac; float *srec;                 // surfel records (rec_blocks = 0: none)
    int perm_blocks, nb, f16; const void *shs; void *shp;                                                 // quad-permuted SH blocks (perm_blocks = 0: none)
};
__global__ void __launch_bounds__(256) forward_prepare(const ForwardPrepare F);
__global__ void __launch_bounds__(64) trace_fwd(const TraceArgs A, const int ray_h, const int ray_w);
__global__ void __launch_bounds__(64) trace_bwd(const TraceArgs A, const int ray_h, const int ray_w);
#ifdef ENVGS_DIAG
__global__ void __launch_bounds__(64) collect_hits(const TraceArgs A);
__global__ void __attribute__((amdgpu_waves_per_eu(8, 8))) __launch_bounds__(64)
collect_hits_packet(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ srec);
__global__ void __attribute__((amdgpu_waves_per_eu(6, 8))) __launch_bounds__(64)
collect_hits_packet4(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ nodes4, const float4 *__restrict__ srec);
__global__ void __launch_bounds__(64) composite_lists_bwd(const TraceArgs A);
#endif
template <bool DEFER, int WAVES> __global__ void __launch_bounds__(256, WAVES)
collect_hits_coop(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ nodes4, const float4 *__restrict__ srec);
#define ENVGS_COOP_DECL(D, W) extern template __global__ void __launch_bounds__(256, W) \
    collect_hits_coop<D, W>(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ nodes4, const float4 *__restrict__ srec);
ENVGS_COOP_DECL(false, 8)
#ifdef ENVGS_DIAG
ENVGS_COOP_DECL(true, 8) ENVGS_COOP_DECL(true, 6)
#endif
#undef ENVGS_COOP_DECL

template <int EMAX, bool LONG, bool QSH> __global__ void __launch_bounds__(256) sort_composite_fwd(const TraceArgs A);
extern template __global__ void __launch_bounds__(256) sort_composite_fwd<4, false, false>(const TraceArgs A);
extern template __global__ void __launch_bounds__(256) sort_composite_fwd<8, true, false>(const TraceArgs A);
extern template __global__ void __launch_bounds__(256) sort_composite_fwd<16, true, false>(const TraceArgs A);
extern template __global__ void __launch_bounds__(256) sort_composite_fwd<4, false, true>(const TraceArgs A);
extern template __global__ void __launch_bounds__(256) sort_composite_fwd<8, true, true>(const TraceArgs A);
extern template __global__ void __launch_bounds__(256) sort_composite_fwd<16, true, true>(const TraceArgs A);
template <bool CACHED> __global__ void __launch_bounds__(64 * RH_W) register_hits(const TraceArgs A);
__global__ void __launch_bounds__(256) row_count(const TraceArgs A, unsigned *__restrict__ blk);
__global__ void __launch_bounds__(256) row_scan_blocks(unsigned *__restrict__ blk, int n, unsigned *rows_used, unsigned *seg_base);
__global__ void __launch_bounds__(256) row_offsets(const TraceArgs A, const unsigned *__restrict__ blk, unsigned *__restrict__ row_off, uint2 *__restrict__ batch_rows,
                                                   const unsigned *__restrict__ seg_base, unsigned long long limit);      // blk: exclusive scan of the per-BATCH row counts
__global__ void __launch_bounds__(256) unpack_surfel_acc(int P, int wfrac, const unsigned long long *__restrict__ acc, unsigned *__restrict__ cnt,
                                                         float *__restrict__ wet, unsigned *ray_counter);
__global__ void __launch_bounds__(256) sparse_hits_bwd(const TraceArgs A, const int rgbo);
template <bool RGBO, bool OTH> __global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64) batch_surfel_bwd(const TraceArgs A);
extern template __global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64) batch_surfel_bwd<false, false>(const TraceArgs A);
extern template __global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64) batch_surfel_bwd<false, true>(const TraceArgs A);
extern template __global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64) batch_surfel_bwd<true, false>(const TraceArgs A);
__global__ void __launch_bounds__(256) reduce_surfel_records(const TraceArgs A);
__global__ void __launch_bounds__(256) finish_surfel_grads(int P, const float *__restrict__ rots, const float *__restrict__ geo_rec,
                                                           float *__restrict__ dmeans, float *__restrict__ dscales, float *__restrict__ dopac,
                                                           float *__restrict__ drots, float *__restrict__ dgrads3D);

}  // namespace envgs
#endif /* ENVGS_TRACE_COMMON_H */
