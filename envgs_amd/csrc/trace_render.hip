// trace_render.hip -- T2 (trace forward) and T3 (trace backward): differentiable front-to-back compositing of
// 2D Gaussians along arbitrary rays, over the LBVH of trace_bvh.hip.
//
// CDNA4 mapping: persistent wavefronts (one 64-lane workgroup each, grid = a few per CU) pull batches of 64 rays
// from a global counter; one lane = one ray.  Traversal keeps the per-lane node stack in LDS ([level][lane], so a
// push/pop is one conflict-free ds_write/ds_read_b32 per wavefront) and the K nearest accepted hits sorted in
// registers; a ray is composited in rounds of K hits, restarting the traversal from (t, id) of the last hit, until
// its transmittance drops below 1e-4 or the scene is exhausted.  A BVH node is one 64 B record holding both child
// boxes; a surfel is one 64 B record (centre, opacity, a/s_u, b/s_v, normal).  The backward re-traces in the
// identical order and uses the stored stage-0 sums for the suffix terms, so no hit list is ever written to HBM.
//
// Stands behind SurfelTracer.forward/backward (easyvolcap/utils/optix_utils.py:188-201); semantics are restated in
// oracle/surfel_trace_oracle.c ("parity unpinned": the OptiX sources are not in the reference tree).
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

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
__device__ __constant__ float tC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float tC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f};

// ---- per-surfel record ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
make_surfel_records(int P, float mod, const float *__restrict__ means, const float *__restrict__ scales,
                    const float *__restrict__ rots, const float *__restrict__ opac, float *__restrict__ srec)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float q0 = rots[4 * i], q1 = rots[4 * i + 1], q2 = rots[4 * i + 2], q3 = rots[4 * i + 3];
    const float inv = 1.0f / sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    const float r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
    const float su = scales[2 * i] * mod, sv = scales[2 * i + 1] * mod;
    float4 *o = reinterpret_cast<float4 *>(srec + (size_t)i * SREC);
    o[0] = make_float4(means[3 * i], means[3 * i + 1], means[3 * i + 2], opac[i]);
    o[1] = make_float4((1.f - 2.f * (y * y + z * z)) / su, (2.f * (x * y + r * z)) / su, (2.f * (x * z - r * y)) / su, su);
    o[2] = make_float4((2.f * (x * y - r * z)) / sv, (1.f - 2.f * (x * x + z * z)) / sv, (2.f * (y * z + r * x)) / sv, sv);
    o[3] = make_float4(2.f * (x * z + r * y), 2.f * (y * z - r * x), 1.f - 2.f * (x * x + y * y), 0.f);
}

struct SurfHit { float t, u, v, G, alpha, denom; bool ok; };

__device__ __forceinline__ SurfHit hit_surfel(const float4 s0, const float4 s1, const float4 s2, const float4 s3,
                                              const float ox, const float oy, const float oz, const float dx,
                                              const float dy, const float dz)
{
    SurfHit h;
    h.denom = s3.x * dx + s3.y * dy + s3.z * dz;
    const float num = s3.x * (s0.x - ox) + s3.y * (s0.y - oy) + s3.z * (s0.z - oz);
    h.t = num / h.denom;
    const float qx = ox + h.t * dx - s0.x, qy = oy + h.t * dy - s0.y, qz = oz + h.t * dz - s0.z;
    h.u = s1.x * qx + s1.y * qy + s1.z * qz;
    h.v = s2.x * qx + s2.y * qy + s2.z * qz;
    h.G = __expf(-0.5f * (h.u * h.u + h.v * h.v));
    const float a = s0.w * h.G;
    h.alpha = a < ALPHA_CAP ? a : ALPHA_CAP;
    h.ok = (h.denom != 0.0f) && (fabsf(h.u) <= UV_MAX) && (fabsf(h.v) <= UV_MAX) && (h.alpha >= ALPHA_MIN);
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
    float spec_thr;
    const float4 *nodes;
    const float4 *srec;
    const float *shs, *colors, *others, *bg;
    const float *ray_o, *ray_d;
    unsigned *counter;
    unsigned long long *stats;      // [hits, node visits, rounds] totals of the forward (diagnostics)
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
    int exp;            // diagnostic switches (ENVGS_TRACE_EXP env var; 0 in production): 8 = atomic-flush backward instead of records,
                        // 16 = binary packet traversal instead of the 4-wide one,
                        // 64 = no coherence sort of the rays, 512 = per-ray collection kernel even when the rays are sorted
    int *stack_spill;   // collect_hits: (grid, STACK, 64) ints of stack overflow space
    int only_overflow;  // K-buffer kernels: process only rays whose hit_cnt exceeds cap
    int batch0, batch1; // list-path forward kernels: the range of 64-ray batches this launch owns (segments run on two streams)
    int seg;            // segment index: selects the batch-fetch counters and the stack-spill slab
    int spill_stride;   // stack-spill slabs per segment
    unsigned long long *entries;  // (batches, 64*cap) distinct (batch, surfel) entries, see register_hits
    unsigned *pairs;              // (batches, 64*cap) (lane << 16 | k) of every composited hit, grouped by entry
    int *n_entries;               // (batches, 2) table entries, single entries
    float4 *state;      // (R, cap, 2 | 3) x 16 B per composited hit: transmittance before it and the prefix sums after it (for the backward)
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

// SH block of one surfel into registers (zeros beyond the active degree).
__device__ __forceinline__ void load_sh(const TraceArgs &A, const int sid, const int nb, float *v)
{
    if (A.M == 16) {
        const float4 *s4 = reinterpret_cast<const float4 *>(A.shs + (size_t)sid * 48);
        const int nq = (nb * 3 + 3) >> 2;
#pragma unroll
        for (int q = 0; q < 12; q++) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < nq) x = s4[q];
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
        const float *sh = A.shs + (size_t)sid * A.M * 3;
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
        if (A.M == 16) {
            // the usual layout (16 coefficients x RGB = 192 B, 16 B aligned): 12 x 16 B loads instead of 48 x 4 B gathers
            const float4 *s4 = reinterpret_cast<const float4 *>(A.shs + (size_t)sid * 48);
            const int nq = (nb * 3 + 3) >> 2;
            float v[48];
#pragma unroll
            for (int q = 0; q < 12; q++) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < nq) x = s4[q];
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            }
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nb) { const float b = basis[k]; r0 += b * v[k * 3]; r1 += b * v[k * 3 + 1]; r2 += b * v[k * 3 + 2]; }
        } else {
            const float *sh = A.shs + (size_t)sid * A.M * 3;
            for (int k = 0; k < nb; k++) { const float b = basis[k]; r0 += b * sh[k * 3]; r1 += b * sh[k * 3 + 1]; r2 += b * sh[k * 3 + 2]; }
        }
        r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
        cl[0] = r0 < 0.f; cl[1] = r1 < 0.f; cl[2] = r2 < 0.f;
        col[0] = cl[0] ? 0.f : r0; col[1] = cl[1] ? 0.f : r1; col[2] = cl[2] ? 0.f : r2;
    } else {
        col[0] = A.colors[3 * sid]; col[1] = A.colors[3 * sid + 1]; col[2] = A.colors[3 * sid + 2];
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

// ------------------------------------------------------------------------------------------ T2 ---
__global__ void __launch_bounds__(64)
trace_fwd(const TraceArgs A, const int ray_h, const int ray_w)
{
    __shared__ int stk[STACK][64];
    const int lane = threadIdx.x;
    // overflow pass after the list path: counter[1] holds the longest list of this call -- nothing overflowed, nothing to do
    if (A.only_overflow && (int)__hip_atomic_load(A.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= A.cap) return;
    while (true) {
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(A.counter, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= A.R) break;
        const int slot = base + lane;
        bool valid = slot < A.R;
        const int r = valid ? ray_index(slot, A.R, ray_h, ray_w) : 0;
        if (A.only_overflow) {
            valid = valid && A.hit_cnt[r] > A.cap;
            if (__builtin_amdgcn_ballot_w64(valid) == 0) continue;
        }
        float ox = A.ray_o[3 * r], oy = A.ray_o[3 * r + 1], oz = A.ray_o[3 * r + 2];
        float dx = A.ray_d[3 * r], dy = A.ray_d[3 * r + 1], dz = A.ray_d[3 * r + 2];
        float tmin = first_tmin(A.start_from_first);
        float out_rgb[3] = {0.f, 0.f, 0.f};
        unsigned st_hits = 0, st_visits = 0, st_rounds = 0;
        float thr = 1.0f;                               // product of specular weights of the previous stages
        bool chain = valid;
        StageSums s0;
        for (int stage = 0; stage < A.ND; stage++) {
            StageSums S;
            S.rgb[0] = S.rgb[1] = S.rgb[2] = 0.f; S.dpt = 0.f; S.acc = 0.f; S.nrm[0] = S.nrm[1] = S.nrm[2] = 0.f;
            S.dist = 0.f; S.aux[0] = S.aux[1] = 0.f; S.T = 1.0f; S.M1 = 0.f; S.M2 = 0.f;
            float basis[16];
            {
                const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                sh_basis(A.D, dx * il, dy * il, dz * il, basis);
            }
            bool done = !chain || A.P == 0;
            float tlo = tmin; int idlo = 0x7fffffff;
            for (int round = 0; round < MAX_ROUNDS; round++) {
                if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
                KBuf kb;
                st_rounds += done ? 0u : 1u;
                traverse(A, stk, lane, !done, ox, oy, oz, dx, dy, dz, tlo, idlo, kb, st_visits);
#pragma unroll 1
                for (int i = 0; i < KBUF; i++) {
                    int sid = 0;
#pragma unroll
                    for (int k = 0; k < KBUF; k++) sid = (k == i) ? kb.id[k] : sid;     // dynamic pick from the register buffer
                    if (!done && i < kb.n) {
                        const float4 *sr = A.srec + (size_t)sid * 4;
                        const float4 s3 = sr[3];
                        const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], s3, ox, oy, oz, dx, dy, dz);
                        const float test_T = S.T * (1.0f - h.alpha);
                        if (test_T < T_EPS) { done = true; }
                        else {
                            const float w = h.alpha * S.T;
                            float col[3]; bool cl[3];
                            surfel_color(A, sid, basis, col, cl);
                            const float tt = h.t > NEAR_N ? h.t : NEAR_N;
                            const float m = FAR_N / (FAR_N - NEAR_N) * (1.0f - NEAR_N / tt);
                            S.dist += (m * m * (1.0f - S.T) + S.M2 - 2.0f * m * S.M1) * w;
                            S.M1 += m * w; S.M2 += m * m * w;
                            S.rgb[0] += w * col[0]; S.rgb[1] += w * col[1]; S.rgb[2] += w * col[2];
                            S.dpt += w * h.t; S.acc += w;
                            const float sg = h.denom < 0.0f ? w : -w;
                            S.nrm[0] += sg * s3.x; S.nrm[1] += sg * s3.y; S.nrm[2] += sg * s3.z;
                            if (A.has_others) { S.aux[0] += w * A.others[2 * sid]; S.aux[1] += w * A.others[2 * sid + 1]; }
                            if (stage == 0) atomic_add_f32(A.wet + sid, w);
                            S.T = test_T;
                            st_hits++;
                        }
                    }
                }
                if (!done) {
                    if (kb.n < KBUF) done = true;
                    else { tlo = kb.t[KBUF - 1]; idlo = kb.id[KBUF - 1]; }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) S.rgb[c] += S.T * (c < A.bg_len ? A.bg[c] : 0.0f);
            if (valid && chain) {
                float *m = A.mid + ((size_t)r * A.ND + stage) * MID;
                m[0] = ox; m[1] = oy; m[2] = oz; m[3] = dx; m[4] = dy; m[5] = dz; m[6] = S.dpt; m[7] = S.acc;
                m[8] = S.nrm[0]; m[9] = S.nrm[1]; m[10] = S.nrm[2]; m[11] = S.aux[0]; m[12] = S.aux[1];
                m[13] = S.rgb[0]; m[14] = S.rgb[1]; m[15] = S.rgb[2];
            }
            if (stage == 0) s0 = S;
            if (chain) {
                // rgb = (1-s0) c0 + s0 ((1-s1) c1 + s1 c2 ...): this stage enters with weight thr * (1 - s_stage) unless it is the last
                const float nl = sqrtf(S.nrm[0] * S.nrm[0] + S.nrm[1] * S.nrm[1] + S.nrm[2] * S.nrm[2]);
                const bool bounce = (stage + 1 < A.ND) && (S.aux[0] > A.spec_thr) && (S.acc > 0.5f) && (nl > 0.0f);
                const float wgt = bounce ? thr * (1.0f - S.aux[0]) : thr;
                out_rgb[0] += wgt * S.rgb[0]; out_rgb[1] += wgt * S.rgb[1]; out_rgb[2] += wgt * S.rgb[2];
                if (bounce) {
                    thr *= S.aux[0];
                    const float inl = 1.0f / nl;
                    const float nx = S.nrm[0] * inl, ny = S.nrm[1] * inl, nz = S.nrm[2] * inl;
                    const float td = S.dpt / S.acc;
                    const float dn = dx * nx + dy * ny + dz * nz;
                    ox = ox + dx * td; oy = oy + dy * td; oz = oz + dz * td;
                    dx = dx - 2.0f * dn * nx; dy = dy - 2.0f * dn * ny; dz = dz - 2.0f * dn * nz;
                    tmin = 1e-3f;
                } else chain = false;
            }
        }
        if (valid) {
            A.rgb[3 * r] = out_rgb[0]; A.rgb[3 * r + 1] = out_rgb[1]; A.rgb[3 * r + 2] = out_rgb[2];
            A.dpt[r] = s0.dpt; A.acc[r] = s0.acc; A.dist[r] = s0.dist;
            A.norm[3 * r] = s0.nrm[0]; A.norm[3 * r + 1] = s0.nrm[1]; A.norm[3 * r + 2] = s0.nrm[2];
            A.aux[2 * r] = s0.aux[0]; A.aux[2 * r + 1] = s0.aux[1];
            A.final_T[r] = s0.T;
        }
        if (A.stats) {
            // per-wavefront totals -> 3 atomics per 64 rays
            const float fh = wave_sum((float)st_hits), fv = wave_sum((float)st_visits), fr = wave_sum((float)st_rounds);
            if (lane == 0) {
                atomicAdd(A.stats + 0, (unsigned long long)fh);
                atomicAdd(A.stats + 1, (unsigned long long)fv);
                atomicAdd(A.stats + 2, (unsigned long long)fr);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ T3 ---
// SH basis k as (l0 + l1 x + l2 y + l3 z) * (q0 + q1 xx + q2 yy + q3 zz + q4 xy + q5 yz + q6 xz): lets lane j evaluate
// "its" basis function (k = j / 3) of ANOTHER lane's ray direction during the cooperative gradient flush.
__device__ __constant__ float kShForm[16][11] = {
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
    B.gR0 = A.g_rgb[3 * r]; B.gR1 = A.g_rgb[3 * r + 1]; B.gR2 = A.g_rgb[3 * r + 2];
    B.gD = A.g_dpt[r]; B.gA = A.g_acc[r];
    B.gN0 = A.g_norm[3 * r]; B.gN1 = A.g_norm[3 * r + 1]; B.gN2 = A.g_norm[3 * r + 2];
    B.gX0 = A.g_aux[2 * r]; B.gX1 = A.g_aux[2 * r + 1];
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

// K-buffer backward (re-traces).  Used for bounce-free rays whose hit list overflowed, and when no list was kept.
__global__ void __launch_bounds__(64)
trace_bwd(const TraceArgs A, const int ray_h, const int ray_w)
{
    __shared__ int stk[STACK][64];
    __shared__ float fld[NFLD][65];                 // row stride 65: lanes 48..62 read 15 different rows of one column conflict-free
    const int lane = threadIdx.x;
    if (A.only_overflow && (int)__hip_atomic_load(A.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= A.cap) return;
    const FlushRole role = flush_role(A, lane);
    const int nb = (A.D + 1) * (A.D + 1);
    while (true) {
        int base = 0;
        if (lane == 0) base = (int)atomicAdd(A.counter, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= A.R) break;
        const int slot = base + lane;
        bool valid = slot < A.R;
        const int r = valid ? ray_index(slot, A.R, ray_h, ray_w) : 0;
        if (A.only_overflow) {
            valid = valid && A.hit_cnt[r] > A.cap;
            if (__builtin_amdgcn_ballot_w64(valid) == 0) continue;
        }
        BwdRay B;
        bwd_load_ray(A, r, B);
        BwdAcc acc;
        bwd_init_acc(acc);
        const float tmin = first_tmin(A.start_from_first);
        float basis[16];
        sh_basis(A.D, B.ux, B.uy, B.uz, basis);
        __syncthreads();                                  // previous batch's flush reads are done
        fld[19][lane] = B.ux; fld[20][lane] = B.uy; fld[21][lane] = B.uz;
        bool done = !valid || A.P == 0;
        float tlo = tmin; int idlo = 0x7fffffff;
        for (int round = 0; round < MAX_ROUNDS; round++) {
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
            KBuf kb;
            unsigned visits_unused = 0;
            traverse(A, stk, lane, !done, B.ox, B.oy, B.oz, B.dx, B.dy, B.dz, tlo, idlo, kb, visits_unused);
#pragma unroll 1
            for (int i = 0; i < KBUF; i++) {
                if (__builtin_amdgcn_ballot_w64(!done && i < kb.n) == 0) break;
                int sid = 0;
#pragma unroll
                for (int k = 0; k < KBUF; k++) sid = (k == i) ? kb.id[k] : sid;     // dynamic pick from the register buffer
                bool has = false;
                float dc0 = 0.f, dc1 = 0.f, dc2 = 0.f, gv[15];
#pragma unroll
                for (int k = 0; k < 15; k++) gv[k] = 0.f;
                if (!done && i < kb.n) {
                    has = bwd_hit(A, B, acc, basis, nb, sid, dc0, dc1, dc2, gv);
                    if (!has) done = true;
                }
                flush_hits(A, fld, lane, role, has, sid, dc0, dc1, dc2, gv);
            }
            if (!done) {
                if (kb.n < KBUF) done = true;
                else { tlo = kb.t[KBUF - 1]; idlo = kb.id[KBUF - 1]; }
            }
        }
        if (valid) bwd_store_ray(A, r, B, acc);
    }
}

// Rays are processed in a coherence-sorted order when A.order is set: 64 consecutive slots = one wavefront = rays with nearly the same
// direction (and nearby origins), so its lanes walk nearly the same BVH nodes and hit the same surfels -- the loads coalesce.
__device__ __forceinline__ int ray_of(const TraceArgs &A, int slot) { return slot < A.R ? (A.order ? (int)A.order[slot] : slot) : A.R; }

// Sort key of a ray: octahedral direction (2 x 8 bits, Morton-interleaved) in the high bits, origin cell (3 x 5 bits) below.
__global__ void __launch_bounds__(256)
make_ray_keys(int R, const float *__restrict__ ray_o, const float *__restrict__ ray_d, const float4 *__restrict__ nodes, int P,
              unsigned *__restrict__ keys, unsigned *__restrict__ vals)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float lo[3] = {-1.f, -1.f, -1.f}, ext[3] = {2.f, 2.f, 2.f};
    if (P > 0) {
        const float4 n0 = nodes[0], n1 = nodes[1], n2 = nodes[2];
        lo[0] = fminf(n0.x, n1.z); lo[1] = fminf(n0.y, n1.w); lo[2] = fminf(n0.z, n2.x);
        ext[0] = fmaxf(n0.w, n2.y) - lo[0]; ext[1] = fmaxf(n1.x, n2.z) - lo[1]; ext[2] = fmaxf(n1.y, n2.w) - lo[2];
    }
    const float dx = ray_d[3 * r], dy = ray_d[3 * r + 1], dz = ray_d[3 * r + 2];
    const float inv = 1.0f / (fabsf(dx) + fabsf(dy) + fabsf(dz) + 1e-30f);
    float u = dx * inv, v = dy * inv;
    if (dz < 0.f) { const float uu = (1.f - fabsf(v)) * (u >= 0.f ? 1.f : -1.f), vv = (1.f - fabsf(u)) * (v >= 0.f ? 1.f : -1.f); u = uu; v = vv; }
    const unsigned qu = (unsigned)fminf(fmaxf((u * 0.5f + 0.5f) * 256.f, 0.f), 255.f), qv = (unsigned)fminf(fmaxf((v * 0.5f + 0.5f) * 256.f, 0.f), 255.f);
    unsigned dkey = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) dkey |= ((qu >> b) & 1u) << (2 * b) | ((qv >> b) & 1u) << (2 * b + 1);
    unsigned okey = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = ext[c] > 0.f ? (ray_o[3 * r + c] - lo[c]) / ext[c] : 0.f;
        const unsigned q = (unsigned)fminf(fmaxf(t * 32.f, 0.f), 31.f);
#pragma unroll
        for (int b = 0; b < 5; b++) okey |= ((q >> b) & 1u) << (3 * b + c);
    }
    keys[r] = (dkey << 15) | okey;
    vals[r] = (unsigned)r;
}

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

// Conservative termination bound for the unordered collection.  The ray's accepted hits are binned by distance into 16
// linear bins over its chord through the scene box (16 registers of optical depth -ln(1-alpha)); as soon as the bins up to edge e hold more optical depth than
// the compositing can survive (T < 1e-4), every hit beyond e is provably after the terminating hit: it is dropped and BVH nodes
// that start beyond e are pruned.  Exact (never drops a composited hit) and it removes most of the 3x over-collection of a fog.
constexpr int NBIN = 16;
constexpr float KILL_OD = 9.2104f * 1.03f + 0.05f;       // -ln(1e-4) with margin for fp32 product vs sum-of-logs

__global__ void __launch_bounds__(64)
collect_hits(const TraceArgs A)
{
    // Shallow LDS stack (6 KB per wavefront -> ~26 wavefronts per CU instead of 10); the rare deeper pushes spill to a
    // per-wavefront slab in HBM.  Hits are sorted afterwards, so the visiting order only matters for how early the bound tightens.
    __shared__ int stk[LDS_STACK][64];
    int *spill = A.stack_spill + ((size_t)A.seg * A.spill_stride + blockIdx.x) * (STACK * 64);
    const int lane = threadIdx.x;
    unsigned visits = 0, rays_done = 0, found_tot = 0;
    // the scene box = union of the root's two child boxes (uniform loads)
    float rlx = 0.f, rly = 0.f, rlz = 0.f, rhx = 0.f, rhy = 0.f, rhz = 0.f;
    if (A.P > 0) {
        const float4 n0 = A.nodes[0], n1 = A.nodes[1], n2 = A.nodes[2];
        rlx = fminf(n0.x, n1.z); rly = fminf(n0.y, n1.w); rlz = fminf(n0.z, n2.x);
        rhx = fmaxf(n0.w, n2.y); rhy = fmaxf(n1.x, n2.z); rhz = fmaxf(n1.y, n2.w);
    }
    const int home = xcc_id();
    const int nbatch = A.batch1 - A.batch0;
    while (true) {
        const int fb = fetch_batch(A.counter + 32 + 8 * A.seg, nbatch, home, lane);
        if (fb < 0) break;
        const int batch = A.batch0 + fb;
        const int base = batch << 6;
        const int r = ray_of(A, base + lane);
        const bool valid = r < A.R;
        const int rr = valid ? r : 0;
        const float ox = A.ray_o[3 * rr], oy = A.ray_o[3 * rr + 1], oz = A.ray_o[3 * rr + 2];
        const float dx = A.ray_d[3 * rr], dy = A.ray_d[3 * rr + 1], dz = A.ray_d[3 * rr + 2];
        const float tmin = first_tmin(A.start_from_first);
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
        // bins: LINEAR in t over the ray's chord through the scene box (a fog terminates after a roughly constant optical depth, i.e. at a
        // roughly constant fraction of the chord: 16 linear bins resolve that point to 1/15 of the chord, half-octave bins to +41 %)
        float tA, bin_w, inv_bin_w;
        {
            const float a0 = (rlx - ox) * ix, a1 = (rhx - ox) * ix, b0 = (rly - oy) * iy, b1 = (rhy - oy) * iy, c0 = (rlz - oz) * iz, c1 = (rhz - oz) * iz;
            const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1)), tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
            tA = fmaxf(tn, tmin);
            float span = tf - tA;
            if (!(span > 0.0f) || !(span < 1.0e30f)) span = 1.0f;               // the ray misses the scene (or a degenerate box): nothing to bin
            bin_w = span * (1.00001f / (float)(NBIN - 1));
            inv_bin_w = 1.0f / bin_w;
        }
        float od[NBIN];
#pragma unroll
        for (int b = 0; b < NBIN; b++) od[b] = 0.f;
        float tkill = 3.0e38f;
        uint2 *list = A.hits + (size_t)rr * A.cap;
        int n = 0;
        int sp = 0;
        int cur = (valid && A.P > 0) ? 0 : -1;
        while (true) {
            if (cur < 0) {
                if (sp == 0) break;
                --sp;
                cur = sp < LDS_STACK ? stk[sp][lane] : spill[(sp - LDS_STACK) * 64 + lane];
            }
            const float4 *nd = A.nodes + (size_t)cur * 4;
            visits++;
            const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
            const int lc = __float_as_int(n3.x), rc = __float_as_int(n3.y);
            float a0 = (n0.x - ox) * ix, a1 = (n0.w - ox) * ix, b0 = (n0.y - oy) * iy, b1 = (n1.x - oy) * iy, c0 = (n0.z - oz) * iz, c1 = (n1.y - oz) * iz;
            const float tnL = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
            const float tfL = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
            a0 = (n1.z - ox) * ix; a1 = (n2.y - ox) * ix; b0 = (n1.w - oy) * iy; b1 = (n2.z - oy) * iy; c0 = (n2.x - oz) * iz; c1 = (n2.w - oz) * iz;
            const float tnR = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
            const float tfR = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
            bool hitL = (tnL <= tfL) && (tfL >= tmin) && (tnL <= tkill);
            bool hitR = (tnR <= tfR) && (tfR >= tmin) && (tnR <= tkill);
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const bool hit = side == 0 ? hitL : hitR;
                const int ch = side == 0 ? lc : rc;
                if (hit && ch < 0) {
                    const int sid = ~ch;
                    const float4 *sr = A.srec + (size_t)sid * 4;
                    const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], sr[3], ox, oy, oz, dx, dy, dz);
                    if (h.ok && h.t > tmin && h.t <= tkill) {
                        if (n < A.cap) list[n] = make_uint2(__float_as_uint(h.t), (unsigned)sid);
                        n++;
                        // bin (biased upwards: a hit may only ever be filed FARTHER than it is, which keeps the bound conservative)
                        const float x = (h.t - tA) * inv_bin_w;
                        int b = x <= 0.0f ? 0 : (int)ceilf(x + 1e-3f);
                        b = b > NBIN - 1 ? NBIN - 1 : b;
                        const float dep = -__logf(1.0f - h.alpha);
#pragma unroll
                        for (int q = 0; q < NBIN; q++) od[q] += (q == b) ? dep : 0.f;
                        float cum = 0.f; int kb = NBIN - 1;
#pragma unroll
                        for (int q = 0; q < NBIN - 1; q++) { cum += od[q]; kb = (cum >= KILL_OD && kb == NBIN - 1) ? q : kb; }
                        tkill = kb < NBIN - 1 ? tA + (float)kb * bin_w : 3.0e38f;
                    }
                }
            }
            hitL = hitL && lc >= 0;
            hitR = hitR && rc >= 0;
            if (hitL && hitR) {
                const bool leftFirst = tnL <= tnR;
                const int farc = leftFirst ? rc : lc;
                if (sp < LDS_STACK) stk[sp][lane] = farc; else if (sp < LDS_STACK + STACK) spill[(sp - LDS_STACK) * 64 + lane] = farc;
                sp++;
                cur = leftFirst ? lc : rc;
            }
            else if (hitL) cur = lc;
            else if (hitR) cur = rc;
            else cur = -1;
        }
        if (valid) { A.hit_cnt[r] = n; rays_done++; found_tot += (unsigned)n; }
        // wave max of n -> global max (adaptive cap of the next call)
        int mx = n;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        if (lane == 0) atomicMax((int *)(A.counter + 1), mx);
    }
    if (A.stats) {
        const float fv = wave_sum((float)visits), ff = wave_sum((float)found_tot);
        if (lane == 0) { atomicAdd(A.stats + 1, (unsigned long long)fv); atomicAdd(A.stats + 3, (unsigned long long)ff); }
    }
}

// Packet form of collect_hits for coherence-sorted rays.  The 64 rays of a batch mostly walk the SAME nodes (measured on the bench
// scene: the union of the surfels a batch finds is 2.4x what one of its rays finds), so the wavefront walks the tree ONCE with a single,
// wave-uniform stack: node and surfel records come in through the scalar unit (s_load: one 64 B fetch per wavefront instead of up to
// 64 gathers), every lane tests its own ray against them with its own termination bound, and control flow never diverges.  A lane
// that is pruned simply stops passing box tests.  Visit order (near child first by majority vote) only affects how early the bounds
// tighten: the lists are sorted afterwards.
constexpr int PSTACK = 64;
// (8 waves per SIMD: with two segments in flight ~10 k wavefronts want a slot; the (n - o) * (1/d) slab form is kept on purpose -- the
//  one-fma form n*(1/d) - o/d needs an error margin proportional to |o/d|, and in a packet ONE ray with a tiny direction component then
//  drags the whole wavefront through nodes nobody hits: measured +0.6 ms)
__global__ void __attribute__((amdgpu_waves_per_eu(8, 8))) __launch_bounds__(64)
collect_hits_packet(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ srec)
{
    __shared__ int stk[PSTACK];
    int *spill = A.stack_spill + ((size_t)A.seg * A.spill_stride + blockIdx.x) * (STACK * 64);
    const int lane = threadIdx.x;
    unsigned visits = 0, found_tot = 0;
    float rlx, rly, rlz, rhx, rhy, rhz;                   // the scene box = union of the root's two child boxes
    {
        const float4 n0 = nodes[0], n1 = nodes[1], n2 = nodes[2];
        rlx = fminf(n0.x, n1.z); rly = fminf(n0.y, n1.w); rlz = fminf(n0.z, n2.x);
        rhx = fmaxf(n0.w, n2.y); rhy = fmaxf(n1.x, n2.z); rhz = fmaxf(n1.y, n2.w);
    }
    const int home = xcc_id();
    const int nbatch = A.batch1 - A.batch0;
    while (true) {
        const int fb = fetch_batch(A.counter + 32 + 8 * A.seg, nbatch, home, lane);
        if (fb < 0) break;
        const int batch = A.batch0 + fb;
        const int base = batch << 6;
        const int r = ray_of(A, base + lane);
        const bool valid = r < A.R;
        const int rr = valid ? r : 0;
        const float ox = A.ray_o[3 * rr], oy = A.ray_o[3 * rr + 1], oz = A.ray_o[3 * rr + 2];
        const float dx = A.ray_d[3 * rr], dy = A.ray_d[3 * rr + 1], dz = A.ray_d[3 * rr + 2];
        const float tmin = first_tmin(A.start_from_first);
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
        // bins: LINEAR in t over the ray's chord through the scene box (a fog terminates after a roughly constant optical depth, i.e. at a
        // roughly constant fraction of the chord: 16 linear bins resolve that point to 1/15 of the chord, half-octave bins to +41 %)
        float tA, bin_w, inv_bin_w;
        {
            const float a0 = (rlx - ox) * ix, a1 = (rhx - ox) * ix, b0 = (rly - oy) * iy, b1 = (rhy - oy) * iy, c0 = (rlz - oz) * iz, c1 = (rhz - oz) * iz;
            const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1)), tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
            tA = fmaxf(tn, tmin);
            float span = tf - tA;
            if (!(span > 0.0f) || !(span < 1.0e30f)) span = 1.0f;               // the ray misses the scene (or a degenerate box): nothing to bin
            bin_w = span * (1.00001f / (float)(NBIN - 1));
            inv_bin_w = 1.0f / bin_w;
        }
        float od[NBIN];
#pragma unroll
        for (int b = 0; b < NBIN; b++) od[b] = 0.f;
        float tkill = 3.0e38f, odtot = 0.f;
        int pend = 0;
        uint2 *list = A.hits + (size_t)rr * A.cap;
        int n = 0;
        int sp = 0;
        int cur = 0;
        visits += (unsigned)__popcll(__ballot(valid));
        while (true) {
            if (cur < 0) {
                if (sp == 0) break;
                --sp;
                int top;
                if (sp < PSTACK) top = stk[sp]; else top = __builtin_nontemporal_load(spill + (sp - PSTACK));
                cur = __builtin_amdgcn_readfirstlane(top);
            }
            const float4 *nd = nodes + (size_t)cur * 4;
            const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3];
            const int lc = __builtin_amdgcn_readfirstlane(__float_as_int(n3.x)), rc = __builtin_amdgcn_readfirstlane(__float_as_int(n3.y));
            float a0 = (n0.x - ox) * ix, a1 = (n0.w - ox) * ix, b0 = (n0.y - oy) * iy, b1 = (n1.x - oy) * iy, c0 = (n0.z - oz) * iz, c1 = (n1.y - oz) * iz;
            const float tnL = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
            const float tfL = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
            a0 = (n1.z - ox) * ix; a1 = (n2.y - ox) * ix; b0 = (n1.w - oy) * iy; b1 = (n2.z - oy) * iy; c0 = (n2.x - oz) * iz; c1 = (n2.w - oz) * iz;
            const float tnR = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
            const float tfR = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
            const bool hitL = valid && (tnL <= tfL) && (tfL >= tmin) && (tnL <= tkill);
            const bool hitR = valid && (tnR <= tfR) && (tfR >= tmin) && (tnR <= tkill);
            const unsigned long long mL = __ballot(hitL), mR = __ballot(hitR);
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const bool hit = side == 0 ? hitL : hitR;
                const int ch = side == 0 ? lc : rc;
                const unsigned long long m = side == 0 ? mL : mR;
                if (ch < 0 && m != 0ull) {
                    const int sid = ~ch;
                    const float4 *sr = srec + (size_t)sid * 4;
                    const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], sr[3], ox, oy, oz, dx, dy, dz);
                    if (hit && h.ok && h.t > tmin && h.t <= tkill) {
                        if (n < A.cap) list[n] = make_uint2(__float_as_uint(h.t), (unsigned)sid);
                        n++;
                        const float x = (h.t - tA) * inv_bin_w;
                        int b = x <= 0.0f ? 0 : (int)ceilf(x + 1e-3f);
                        b = b > NBIN - 1 ? NBIN - 1 : b;
                        const float dep = -__logf(1.0f - h.alpha);
#pragma unroll
                        for (int q = 0; q < NBIN; q++) od[q] += (q == b) ? dep : 0.f;
                        odtot += dep;
                    }
                    pend++;
                }
            }
            if (pend >= 3) {               // refresh the bound every third leaf test (a stale bound only collects a little more)
                pend = 0;
                if (__ballot(odtot >= KILL_OD) != 0ull) {
                    float cum = 0.f; int kb = NBIN - 1;
#pragma unroll
                    for (int q = 0; q < NBIN - 1; q++) { cum += od[q]; kb = (cum >= KILL_OD && kb == NBIN - 1) ? q : kb; }
                    tkill = kb < NBIN - 1 ? tA + (float)kb * bin_w : 3.0e38f;
                }
            }
            const bool goL = lc >= 0 && mL != 0ull, goR = rc >= 0 && mR != 0ull;
            if (goL) visits += (unsigned)__popcll(mL);
            if (goR) visits += (unsigned)__popcll(mR);
            if (goL && goR) {
                const unsigned long long both = mL & mR, lf = __ballot(hitL && hitR && tnL <= tnR);
                const bool leftFirst = both ? (2 * __popcll(lf) >= __popcll(both)) : (__popcll(mL) >= __popcll(mR));
                const int farc = leftFirst ? rc : lc;
                if (sp < PSTACK) stk[sp] = farc; else if (sp < PSTACK + STACK * 64) spill[sp - PSTACK] = farc;
                sp++;
                cur = leftFirst ? lc : rc;
            }
            else if (goL) cur = lc;
            else if (goR) cur = rc;
            else cur = -1;
        }
        if (valid) { A.hit_cnt[r] = n; found_tot += (unsigned)n; }
        int mx = n;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        if (lane == 0) atomicMax((int *)(A.counter + 1), mx);
    }
    if (A.stats) {
        const float ff = wave_sum((float)found_tot);
        if (lane == 0) { atomicAdd(A.stats + 1, (unsigned long long)visits); atomicAdd(A.stats + 3, (unsigned long long)ff); }
    }
}

// The same traversal over the 4-wide nodes (trace_bvh.hip: node4[i] = the grandchildren of binary node i): half the steps, and each step is
// one scalar-load round trip plus the stack / mask bookkeeping of the scalar unit, which is what the binary walk spends most of its time on.
// Children that any ray hits are entered nearest first, ordered by the entry distance of each child's first hitting lane (the rays of a
// batch are coherent; the order only affects how early the termination bounds tighten).  `visits` counts 64 B units (two per wide node).
__global__ void __attribute__((amdgpu_waves_per_eu(6, 8))) __launch_bounds__(64)
collect_hits_packet4(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ nodes4, const float4 *__restrict__ srec)
{
    __shared__ int stk[PSTACK];
    int *spill = A.stack_spill + ((size_t)A.seg * A.spill_stride + blockIdx.x) * (STACK * 64);
    const int lane = threadIdx.x;
    unsigned visits = 0, found_tot = 0;
    float rlx, rly, rlz, rhx, rhy, rhz;                   // the scene box = union of the root's two child boxes
    {
        const float4 n0 = nodes[0], n1 = nodes[1], n2 = nodes[2];
        rlx = fminf(n0.x, n1.z); rly = fminf(n0.y, n1.w); rlz = fminf(n0.z, n2.x);
        rhx = fmaxf(n0.w, n2.y); rhy = fmaxf(n1.x, n2.z); rhz = fmaxf(n1.y, n2.w);
    }
    const int home = xcc_id();
    const int nbatch = A.batch1 - A.batch0;
    while (true) {
        const int fb = fetch_batch(A.counter + 32 + 8 * A.seg, nbatch, home, lane);
        if (fb < 0) break;
        const int batch = A.batch0 + fb;
        const int base = batch << 6;
        const int r = ray_of(A, base + lane);
        const bool valid = r < A.R;
        const int rr = valid ? r : 0;
        const float ox = A.ray_o[3 * rr], oy = A.ray_o[3 * rr + 1], oz = A.ray_o[3 * rr + 2];
        const float dx = A.ray_d[3 * rr], dy = A.ray_d[3 * rr + 1], dz = A.ray_d[3 * rr + 2];
        const float tmin = first_tmin(A.start_from_first);
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
        float tA, bin_w, inv_bin_w;
        {
            const float a0 = (rlx - ox) * ix, a1 = (rhx - ox) * ix, b0 = (rly - oy) * iy, b1 = (rhy - oy) * iy, c0 = (rlz - oz) * iz, c1 = (rhz - oz) * iz;
            const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1)), tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
            tA = fmaxf(tn, tmin);
            float span = tf - tA;
            if (!(span > 0.0f) || !(span < 1.0e30f)) span = 1.0f;
            bin_w = span * (1.00001f / (float)(NBIN - 1));
            inv_bin_w = 1.0f / bin_w;
        }
        float od[NBIN];
#pragma unroll
        for (int b = 0; b < NBIN; b++) od[b] = 0.f;
        float tkill = 3.0e38f, odtot = 0.f;
        int pend = 0;
        uint2 *list = A.hits + (size_t)rr * A.cap;
        int n = 0;
        int sp = 0;
        int cur = 0;
        while (true) {
            if (cur < 0) {
                if (sp == 0) break;
                --sp;
                int top;
                if (sp < PSTACK) top = stk[sp]; else top = __builtin_nontemporal_load(spill + (sp - PSTACK));
                cur = __builtin_amdgcn_readfirstlane(top);
            }
            const float4 *nd = nodes4 + (size_t)cur * 8;
            const float4 qlx = nd[0], qly = nd[1], qlz = nd[2], qhx = nd[3], qhy = nd[4], qhz = nd[5], qrf = nd[6];
            const float lxs[4] = {qlx.x, qlx.y, qlx.z, qlx.w}, lys[4] = {qly.x, qly.y, qly.z, qly.w}, lzs[4] = {qlz.x, qlz.y, qlz.z, qlz.w};
            const float hxs[4] = {qhx.x, qhx.y, qhx.z, qhx.w}, hys[4] = {qhy.x, qhy.y, qhy.z, qhy.w}, hzs[4] = {qhz.x, qhz.y, qhz.z, qhz.w};
            const float rfs[4] = {qrf.x, qrf.y, qrf.z, qrf.w};
            int key[4], ref[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int ch = __builtin_amdgcn_readfirstlane(__float_as_int(rfs[c]));
                const float a0 = (lxs[c] - ox) * ix, a1 = (hxs[c] - ox) * ix, b0 = (lys[c] - oy) * iy, b1 = (hys[c] - oy) * iy,
                            c0 = (lzs[c] - oz) * iz, c1 = (hzs[c] - oz) * iz;
                const float tn = fmaxf(fmaxf(fminf(a0, a1), fminf(b0, b1)), fminf(c0, c1));
                const float tf = fminf(fminf(fmaxf(a0, a1), fmaxf(b0, b1)), fmaxf(c0, c1));
                const bool hit = valid && (tn <= tf) && (tf >= tmin) && (tn <= tkill);
                const unsigned long long m = __ballot(hit);
                key[c] = 0x7fffffff; ref[c] = -1;
                if (m != 0ull) {
                    if (ch < 0) {
                        const int sid = ~ch;
                        const float4 *sr = srec + (size_t)sid * 4;
                        const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], sr[3], ox, oy, oz, dx, dy, dz);
                        if (hit && h.ok && h.t > tmin && h.t <= tkill) {
                            if (n < A.cap) list[n] = make_uint2(__float_as_uint(h.t), (unsigned)sid);
                            n++;
                            const float x = (h.t - tA) * inv_bin_w;
                            int b = x <= 0.0f ? 0 : (int)ceilf(x + 1e-3f);
                            b = b > NBIN - 1 ? NBIN - 1 : b;
                            const float dep = -__logf(1.0f - h.alpha);
#pragma unroll
                            for (int q = 0; q < NBIN; q++) od[q] += (q == b) ? dep : 0.f;
                            odtot += dep;
                        }
                        pend++;
                    } else {
                        // entry distance of the first hitting lane, clamped at 0 so that the float bits order like integers
                        const int fl = (int)__builtin_ctzll(m);
                        key[c] = __builtin_amdgcn_readlane(__float_as_int(fmaxf(tn, 0.0f)), fl);
                        ref[c] = ch;
                        visits += 2u * (unsigned)__popcll(m);
                    }
                }
            }
            if (pend >= 3) {               // refresh the bound every third leaf test (a stale bound only collects a little more)
                pend = 0;
                if (__ballot(odtot >= KILL_OD) != 0ull) {
                    float cum = 0.f; int kb = NBIN - 1;
#pragma unroll
                    for (int q = 0; q < NBIN - 1; q++) { cum += od[q]; kb = (cum >= KILL_OD && kb == NBIN - 1) ? q : kb; }
                    tkill = kb < NBIN - 1 ? tA + (float)kb * bin_w : 3.0e38f;
                }
            }
            // sort the (at most four) internal children by key: 5 scalar compare-exchanges; unused slots carry INT_MAX and end up last
#define ENVGS_CSWAP(a, b) { const bool sw = key[a] > key[b]; const int ka = sw ? key[b] : key[a], kb2 = sw ? key[a] : key[b], \
                                       ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b]; key[a] = ka; key[b] = kb2; ref[a] = ra; ref[b] = rb; }
            ENVGS_CSWAP(0, 1) ENVGS_CSWAP(2, 3) ENVGS_CSWAP(0, 2) ENVGS_CSWAP(1, 3) ENVGS_CSWAP(1, 2)
#undef ENVGS_CSWAP
            // nearest next; the others go on the stack far to near
#pragma unroll
            for (int c = 3; c >= 1; c--)
                if (ref[c] >= 0) {
                    if (sp < PSTACK) stk[sp] = ref[c]; else if (sp < PSTACK + STACK * 64) spill[sp - PSTACK] = ref[c];
                    sp++;
                }
            cur = ref[0];
        }
        if (valid) { A.hit_cnt[r] = n; found_tot += (unsigned)n; }
        int mx = n;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        if (lane == 0) atomicMax((int *)(A.counter + 1), mx);
    }
    if (A.stats) {
        const float ff = wave_sum((float)found_tot);
        if (lane == 0) { atomicAdd(A.stats + 1, (unsigned long long)visits); atomicAdd(A.stats + 3, (unsigned long long)ff); }
    }
}

constexpr int SORT_MAX = 1024;
// Cross-lane fetch of a 32-bit value from lane ^ S (S < 64): DPP quad permutes for 1 and 2, ds_swizzle (crossbar only, no LDS memory)
// for 4, 8, 16, v_permlane32_swap for 32.
template <int S>
__device__ __forceinline__ unsigned xlane(unsigned v)
{
    if constexpr (S == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
    else if constexpr (S == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
    else if constexpr (S == 32) {
        const envgs_u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r.x : r.y;
    } else return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (S << 10));
}

// One compare-exchange layer of the bitonic network over E*64 keys held as E registers per lane (element e*64 + lane).
template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_layer(unsigned long long (&k)[E], const int lane)
{
    if constexpr (STRIDE >= 64) {
        constexpr int SE = STRIDE / 64;
#pragma unroll
        for (int e = 0; e < E; e++)
            if ((e & SE) == 0) {
                const bool up = ((e * 64) & SIZE) == 0;                   // SIZE >= 128 here: decided by the register index alone
                const unsigned long long a = k[e], b = k[e | SE];
                const bool sw = (a > b) == up;
                k[e] = sw ? b : a; k[e | SE] = sw ? a : b;
            }
    } else {
        const bool lower = (lane & STRIDE) == 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const bool up = SIZE < 64 ? ((lane & SIZE) == 0) : (((e * 64) & SIZE) == 0);
            const unsigned long long mine = k[e];
            const unsigned long long p = ((unsigned long long)xlane<STRIDE>((unsigned)(mine >> 32)) << 32) | xlane<STRIDE>((unsigned)mine);
            const bool keepmin = lower == up;
            k[e] = ((p < mine) == keepmin) ? p : mine;
        }
    }
}
template <int E, int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge(unsigned long long (&k)[E], const int lane)
{
    bitonic_layer<E, SIZE, STRIDE>(k, lane);
    if constexpr (STRIDE > 1) bitonic_merge<E, SIZE, STRIDE / 2>(k, lane);
}
template <int E, int SIZE>
__device__ __forceinline__ void bitonic_sort(unsigned long long (&k)[E], const int lane)
{
    if constexpr (SIZE > 2) bitonic_sort<E, SIZE / 2>(k, lane);
    bitonic_merge<E, SIZE, SIZE / 2>(k, lane);
}

// Sort AND composite, one wavefront per ray, one LANE per hit.  A lane-per-ray walk is a chain of dependent
// gathers -- list entry -> surfel record + SH block -> blend -> next entry -- whose length is the ray's hit count; here the 64 hits of
// a chunk fetch their records independently (all gathers in flight at once) and the front-to-back recurrences (transmittance product,
// the two distortion moments, the ten blended sums) become wavefront scans.  The (t, id) keys are sorted IN REGISTERS -- E keys per
// lane, a bitonic network whose cross-lane layers use DPP / ds_swizzle / permlane32 and whose long strides are register-to-register --
// so the sorted chunk c is simply register c: no LDS, no barriers, no bank conflicts.  The sorted list is written back only up to the
// terminating hit.
template <int E>
__device__ __forceinline__ void sort_composite_ray(const TraceArgs &A, const int r, const int n, const int lane, unsigned &st_hits)
{
    uint2 *list = A.hits + (size_t)r * A.cap;
    unsigned long long kreg[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int i = e * 64 + lane;
        unsigned long long kk = ~0ull;
        if (i < n) { const uint2 q = list[i]; kk = ((unsigned long long)q.x << 32) | q.y; }
        kreg[e] = kk;
    }
    bitonic_sort<E, E * 64>(kreg, lane);
    const float ox = A.ray_o[3 * r], oy = A.ray_o[3 * r + 1], oz = A.ray_o[3 * r + 2];
    const float dx = A.ray_d[3 * r], dy = A.ray_d[3 * r + 1], dz = A.ray_d[3 * r + 2];
    float basis[16];
    {
        const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        sh_basis(A.D, dx * il, dy * il, dz * il, basis);
    }
    // carried across chunks (wave-uniform): transmittance, the two distortion moments, and the ten blended sums
    // [rgb 3, depth, acc, normal 3, aux 2] -- kept as running PREFIX sums because the backward needs them per hit
    float T = 1.0f, M1 = 0.f, M2 = 0.f, C[10];
#pragma unroll
    for (int j = 0; j < 10; j++) C[j] = 0.f;
    float dist = 0.f;                                   // per-lane partial sum
    int used = 0;
    const int sstr = A.has_others ? 3 : 2;              // per-hit state row: 32 B, or 48 B when the two `others` sums are needed too
    float4 *state = A.state ? A.state + (size_t)r * A.cap * sstr : nullptr;
#pragma unroll
    for (int ce = 0; ce < E; ce++) {
        const int cb = ce * 64;
        if (cb >= n) break;
        const int i = cb + lane;
        const bool has = i < n;
        int sid = 0;
        float alpha = 0.f, t = 0.f, sg = 0.f;
        float4 s3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has) {
            sid = (int)(unsigned)kreg[ce];
            const float4 *sr = A.srec + (size_t)sid * 4;
            s3 = sr[3];
            const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], s3, ox, oy, oz, dx, dy, dz);
            alpha = h.alpha; t = h.t; sg = h.denom < 0.0f ? 1.f : -1.f;
        }
        const float P = wave_scan_mul(1.0f - alpha);                    // prod_{j<=i} (1 - alpha_j) within the chunk
        const float Pex = dpp_fill<0x138>(P, 1.f);                      // wave_shr:1
        const float test_T = T * P, Tb = T * Pex;                       // transmittance after / before this hit
        const unsigned long long stop = __ballot(has && test_T < T_EPS);
        const int f = stop ? (int)__builtin_ctzll(stop) : 64;           // first terminating lane: it and everything behind is dropped
        const bool use = has && lane < f;
        const float w = use ? alpha * Tb : 0.f;
        float col[3] = {0.f, 0.f, 0.f}; bool cl[3];
        if (use) surfel_color(A, sid, basis, col, cl);
        const float tt = t > NEAR_N ? t : NEAR_N;
        const float m = FAR_N / (FAR_N - NEAR_N) * (1.0f - NEAR_N / tt);
        const float mw = m * w, mmw = m * m * w;
        const float S1 = wave_scan_add(mw), S2 = wave_scan_add(mmw);
        const float M1b = M1 + (S1 - mw), M2b = M2 + (S2 - mmw);        // moments before this hit
        dist += (m * m * (1.0f - Tb) + M2b - 2.0f * m * M1b) * w;
        float x0 = 0.f, x1 = 0.f;
        if (A.has_others && use) { x0 = A.others[2 * sid]; x1 = A.others[2 * sid + 1]; }
        float S[10] = {w * col[0], w * col[1], w * col[2], w * t, w, sg * w * s3.x, sg * w * s3.y, sg * w * s3.z, w * x0, w * x1};
#pragma unroll
        for (int j = 0; j < 10; j++) S[j] = C[j] + wave_scan_add(S[j]);               // inclusive: this hit already added
        if (use) {
            list[i] = make_uint2(__float_as_uint(w), (unsigned)sid);
            if (state) {
                // (the acc sum S[4] is not stored: sum_{j<=k} w_j = 1 - T_before * (1 - alpha), which the backward rebuilds)
                float4 *o = state + (size_t)i * sstr;
                // streamed once, read once by the backward much later: non-temporal, so it does not evict the surfel records / SH blocks
                typedef float nt4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((nt4){Tb, S[0], S[1], S[2]}, reinterpret_cast<nt4 *>(o));
                __builtin_nontemporal_store((nt4){S[3], S[5], S[6], S[7]}, reinterpret_cast<nt4 *>(o + 1));
                if (A.has_others) __builtin_nontemporal_store((nt4){S[8], S[9], 0.f, 0.f}, reinterpret_cast<nt4 *>(o + 2));
            }
        }
        const int nu = f < 64 ? f : min(64, n - cb);                    // hits of this chunk that were blended
        used += nu;
        M1 += wave_bcast(S1, 63); M2 += wave_bcast(S2, 63);
#pragma unroll
        for (int j = 0; j < 10; j++) C[j] = wave_bcast(S[j], 63);
        if (nu > 0) T = T * wave_bcast(P, nu - 1);
        if (f < 64) break;
    }
    st_hits += (unsigned)used;
    dist = wave_sum(dist);
    if (lane == 0) {
        A.n_used[r] = used;
        const float c0 = C[0] + T * (0 < A.bg_len ? A.bg[0] : 0.f), c1 = C[1] + T * (1 < A.bg_len ? A.bg[1] : 0.f), c2 = C[2] + T * (2 < A.bg_len ? A.bg[2] : 0.f);
        A.rgb[3 * r] = c0; A.rgb[3 * r + 1] = c1; A.rgb[3 * r + 2] = c2;
        A.dpt[r] = C[3]; A.acc[r] = C[4]; A.dist[r] = dist;
        A.norm[3 * r] = C[5]; A.norm[3 * r + 1] = C[6]; A.norm[3 * r + 2] = C[7];
        A.aux[2 * r] = C[8]; A.aux[2 * r + 1] = C[9];
        A.final_T[r] = T;
        float *mm = A.mid + (size_t)r * MID;
        mm[0] = ox; mm[1] = oy; mm[2] = oz; mm[3] = dx; mm[4] = dy; mm[5] = dz; mm[6] = C[3]; mm[7] = C[4];
        mm[8] = C[5]; mm[9] = C[6]; mm[10] = C[7]; mm[11] = C[8]; mm[12] = C[9]; mm[13] = c0; mm[14] = c1; mm[15] = c2;
    }
}

// The widest sort a kernel must be able to run sets its register count (E <= 4: 123 VGPRs = 4 waves/SIMD, 8: 150 = 3, 16: 211 = 2), and
// lists longer than 256 hits are rare, so the work is split by list length: the main pass (LONG = false) takes every ray with at most 256
// hits at 4 waves/SIMD; when the capacity allows longer lists a second launch (LONG = true, EMAX = 8 or 16) picks up the few rays
// beyond 256 -- it scans the hit counts 64 rays per wavefront step and only sorts what the ballot finds.
template <int EMAX, bool LONG>
__global__ void __launch_bounds__(256)
sort_composite_fwd(const TraceArgs A)
{
    // 4 wavefronts per workgroup take 4 CONSECUTIVE rays of the coherence-sorted order: they blend mostly the same surfels at the same
    // time, so the records / SH blocks one of them pulls into this CU's L1 serve the others
    const int lane = threadIdx.x & 63;
    unsigned st_hits = 0;
    const int slot_end = min(A.R, A.batch1 * 64);
    if constexpr (!LONG) {
        for (int slot = A.batch0 * 64 + blockIdx.x * 4 + (threadIdx.x >> 6); slot < slot_end; slot += gridDim.x * 4) {
            const int r = ray_of(A, slot);
            const int n = A.hit_cnt[r];
            if (n > A.cap || n > 256) continue;                 // overflow: the K-buffer kernel owns this ray; long: the LONG pass does
            if (n <= 64) sort_composite_ray<1>(A, r, n, lane, st_hits);
            else if (n <= 128) sort_composite_ray<2>(A, r, n, lane, st_hits);
            else sort_composite_ray<4>(A, r, n, lane, st_hits);
        }
    } else {
        // the longest list so far (this segment's collection has finished, so its own maximum is in): nothing to do in the usual case
        if ((int)__hip_atomic_load(A.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 256) return;
        for (int base = A.batch0 * 64 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; base < slot_end; base += gridDim.x * 256) {
            const int slot = base + lane;
            int r = 0, n = 0;
            if (slot < slot_end) { r = ray_of(A, slot); n = A.hit_cnt[r]; }
            unsigned long long todo = __ballot(n > 256 && n <= A.cap);
            while (todo) {
                const int l = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                const int rr = __shfl(r, l), nn = __shfl(n, l);
                if (EMAX == 8 || nn <= 512) sort_composite_ray<8>(A, rr, nn, lane, st_hits);
                else if constexpr (EMAX >= 16) sort_composite_ray<16>(A, rr, nn, lane, st_hits);
            }
        }
    }
    if (A.stats && lane == 0 && st_hits) atomicAdd(A.stats + 0, (unsigned long long)st_hits);
}

// Register every composited hit with its surfel, per BATCH of 64 coherence-sorted rays.  The rays of a batch mostly composite the SAME
// surfels (measured: 27 hits per distinct surfel per batch), and device-scope atomics run at ~10 G/s on this chip whatever their width
// or scope, so a wavefront first merges its batch in an LDS hash table (ds_cmpst / ds_add: hit count, weight sum in 40-bit fixed point)
// and then spends ONE global 64-bit atomic per DISTINCT surfel: weight += sum (rounded up, so any contribution keeps the surfel
// "visible") and entry count += 1, whose old value is the slot of this (batch, surfel) ENTRY among the surfel's entries -- where the
// backward will put the entry's gradient record.  Outputs for the backward, per batch b (region = 64*cap slots):
//   entries[b][e]  e < D: the distinct surfels of the table, packed  sid | (hits-1) << 24 | slot << 32 ; singles that found no room in
//                  the table are filed from the TOP of the region downwards (n_entries[2b] = D, n_entries[2b+1] = singles)
//   pairs[b][...]  (lane << 16 | k) of every hit, grouped by entry in entry order (singles again from the top)
constexpr int RH_TAB = 1024;
constexpr int RH_STAGE = 8192;                           // pairs staged in LDS per batch (32 KB); the rest, if any, is stored directly
constexpr int RH_W = 8;                                  // wavefronts per batch: wave q takes list positions q, q + RH_W, ... of every ray
__global__ void __launch_bounds__(64 * RH_W)
register_hits(const TraceArgs A)
{
    __shared__ int key[RH_TAB];
    __shared__ unsigned long long acc[RH_TAB];       // low 8 bits: hits of this surfel in the batch (<= 64), above: fixed-point weight sum -- ONE
                                                     // returning ds_add_rtn_u64 per hit gives its rank; after the flush: offset of its first pair
    __shared__ unsigned nfail, ndense;
    __shared__ unsigned short hod[RH_TAB];           // table slot of the d-th distinct surfel, in order of first appearance
    __shared__ unsigned pstage[RH_STAGE];            // the batch's pairs, assembled here and written out as one contiguous run (a scattered 4 B
                                                     // store costs a whole 32 B sector of write traffic)
    __shared__ unsigned ptotal;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const float wscale = __builtin_ldexpf(1.0f, A.wfrac);
    const size_t region = (size_t)64 * A.cap;
    for (int base = (A.batch0 + (int)blockIdx.x) * 64; base < min(A.R, A.batch1 * 64); base += gridDim.x * 64) {
        const int batch = base >> 6;
        const int copy = batch & (NCOPY - 1);
        unsigned long long *ent = A.entries ? A.entries + (size_t)batch * region : nullptr;
        unsigned *prs = A.pairs ? A.pairs + (size_t)batch * region : nullptr;
        __syncthreads();
        for (int i = threadIdx.x; i < RH_TAB; i += 64 * RH_W) { key[i] = -1; acc[i] = 0ull; }
        if (threadIdx.x == 0) { nfail = 0u; ndense = 0u; }
        __syncthreads();
        const int r = ray_of(A, base + lane);
        int n = 0;
        uint2 *list = A.hits;
        if (r < A.R && A.hit_cnt[r] <= A.cap) { n = A.n_used[r]; list = A.hits + (size_t)r * A.cap; }
        constexpr int U = 4;
        // (each lane walks its own list row: the loads of one step are 64 different cache lines, so the next step's entries are requested
        //  before this step's chain of LDS atomics starts)
        uint2 nxt[U];
#pragma unroll
        for (int j = 0; j < U; j++) { const int k = part + j * RH_W; nxt[j] = (k < n) ? list[k] : make_uint2(0u, 0u); }
        for (int kb = part; kb < n; kb += U * RH_W) {
            uint2 e[U];
#pragma unroll
            for (int j = 0; j < U; j++) e[j] = nxt[j];
#pragma unroll
            for (int j = 0; j < U; j++) { const int k = kb + (U + j) * RH_W; nxt[j] = (k < n) ? list[k] : make_uint2(0u, 0u); }
#pragma unroll
            for (int j = 0; j < U; j++) {
                const int k = kb + j * RH_W;
                if (k >= n) break;
                const unsigned long long wq = (unsigned long long)ceilf(__uint_as_float(e[j].x) * wscale);
                unsigned h = (e[j].y * 2654435761u) >> 22;
                bool ok = false;
                for (int t = 0; t < 24; t++) {
                    const int old = atomicCAS(&key[h], -1, (int)e[j].y);
                    if (old == -1) hod[atomicAdd(&ndense, 1u)] = (unsigned short)h;      // first to see this surfel: entries keep this order
                    if (old == -1 || old == (int)e[j].y) { ok = true; break; }
                    h = (h + 1) & (RH_TAB - 1);
                }
                unsigned x = 0xFFFFFFFFu;
                if (ok) {
                    const unsigned rank = (unsigned)(atomicAdd(&acc[h], (wq << 8) | 1ull) & 0xFFull);
                    x = (h << 8) | rank;                                  // rank < 64: a ray meets a planar surfel once
                } else {                                                  // table full around h: an entry of its own
                    const unsigned long long old = atomicAdd(A.surf_acc + (size_t)e[j].y * NCOPY + copy, (wq << 24) | 1ull);
                    const unsigned f = atomicAdd(&nfail, 1u);
                    if (ent) ent[region - 1 - f] = (unsigned long long)e[j].y | ((old & 0xFFFFFFull) << 32);
                    if (prs) prs[region - 1 - f] = ((unsigned)lane << 16) | (unsigned)k;
                }
                list[k].x = x;
            }
        }
        __syncthreads();
        // Flush in order of FIRST APPEARANCE along the rays (~ front to back): the backward then meets each ray's hits in roughly ascending
        // list position, so the per-hit state it gathers is consumed cache line by cache line instead of at random.
        const unsigned D = ndense;
        if (part == 0) {
            unsigned carry_off = 0u;
            for (unsigned c = 0; c < D; c += 64) {
                const unsigned d = c + lane;
                const bool occ = d < D;
                const int h = occ ? (int)hod[d] : 0;
                const int sid = occ ? key[h] : 0;
                const unsigned long long av = occ ? acc[h] : 0ull;
                const unsigned cn = (unsigned)(av & 0xFFull);
                const float incl = wave_scan_add((float)cn);                 // exact: at most 64*cap < 2^24 hits per batch
                const unsigned offh = carry_off + (unsigned)incl - cn;
                if (occ) {
                    const unsigned long long old = atomicAdd(A.surf_acc + (size_t)sid * NCOPY + copy, ((av >> 8) << 24) | 1ull);
                    if (ent) ent[d] = (unsigned long long)(unsigned)sid | ((unsigned long long)(cn - 1u) << 24) | ((old & 0xFFFFFFull) << 32);
                    acc[h] = (unsigned long long)offh;
                }
                carry_off += (unsigned)wave_bcast(incl, 63);
            }
            if (A.n_entries && lane == 0) { A.n_entries[2 * batch] = (int)D; A.n_entries[2 * batch + 1] = (int)nfail; }
            if (lane == 0) ptotal = carry_off;
        }
        __syncthreads();
        if (prs) {
            constexpr int U2 = 4;                               // independent loads first: one memory round trip per 4 hits, not per hit
            for (int kb = part; kb < n; kb += U2 * RH_W) {
                unsigned x[U2];
#pragma unroll
                for (int j = 0; j < U2; j++) { const int k = kb + j * RH_W; x[j] = (k < n) ? list[k].x : 0xFFFFFFFFu; }
#pragma unroll
                for (int j = 0; j < U2; j++)
                    if (x[j] != 0xFFFFFFFFu) {
                        const unsigned idx = (unsigned)acc[x[j] >> 8] + (x[j] & 255u), v = ((unsigned)lane << 16) | (unsigned)(kb + j * RH_W);
                        if (idx < (unsigned)RH_STAGE) pstage[idx] = v; else prs[idx] = v;
                    }
            }
            __syncthreads();
            const unsigned T = min(ptotal, (unsigned)RH_STAGE);
            for (unsigned i = threadIdx.x; i < T; i += 64 * RH_W) prs[i] = pstage[i];
        }
    }
}

__global__ void __launch_bounds__(64)
composite_lists_bwd(const TraceArgs A)
{
    __shared__ float fld[NFLD][65];
    const int lane = threadIdx.x;
    const FlushRole role = flush_role(A, lane);
    const int nb = (A.D + 1) * (A.D + 1);
    for (int base = blockIdx.x * 64; base < A.R; base += gridDim.x * 64) {
        const int r = ray_of(A, base + lane);
        const bool valid = r < A.R && A.hit_cnt[r < A.R ? r : 0] <= A.cap;
        const int rr = r < A.R ? r : 0;
        BwdRay B;
        bwd_load_ray(A, rr, B);
        BwdAcc acc;
        bwd_init_acc(acc);
        float basis[16];
        sh_basis(A.D, B.ux, B.uy, B.uz, basis);
        __syncthreads();
        fld[19][lane] = B.ux; fld[20][lane] = B.uy; fld[21][lane] = B.uz;
        const int n = valid ? A.n_used[rr] : 0;
        const uint2 *list = A.hits + (size_t)rr * A.cap;
        int nmax = n;
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
        for (int k = 0; k < nmax; k++) {
            bool has = false;
            int sid = 0;
            float dc0 = 0.f, dc1 = 0.f, dc2 = 0.f, gv[15];
#pragma unroll
            for (int q = 0; q < 15; q++) gv[q] = 0.f;
            if (k < n) {
                sid = (int)list[k].y;
                has = bwd_hit(A, B, acc, basis, nb, sid, dc0, dc1, dc2, gv);
            }
            flush_hits(A, fld, lane, role, has, sid, dc0, dc1, dc2, gv);
        }
        if (valid) bwd_store_ray(A, r, B, acc);
    }
}

// Split the packed per-surfel accumulators of composite_lists_fwd into hit counts (for the scan) and weights (added to `wet`,
// which the K-buffer path may already have contributed to in float).
__global__ void __launch_bounds__(256)
unpack_surfel_acc(int P, int wfrac, const unsigned long long *__restrict__ acc, unsigned *__restrict__ cnt, float *__restrict__ wet)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    unsigned long long wsum = 0;
#pragma unroll
    for (int c = 0; c < NCOPY; c++) {
        const unsigned long long a = acc[(size_t)i * NCOPY + c];
        cnt[(size_t)i * NCOPY + c] = (unsigned)(a & 0xFFFFFFull);
        wsum += a >> 24;
    }
    const float w = (float)((double)wsum / (double)(1ull << wfrac));
    if (w != 0.0f) wet[i] += w;
}

// Backward of the list path, SURFEL-MAJOR per batch (the tracer's counterpart of the rasterizer's tile backward): one wavefront owns a
// batch of 64 coherence-sorted rays, LANE = RAY.  It walks the batch's entries (distinct surfels); the surfel's record and SH block are
// staged through LDS 16 entries ahead (coalesced, off the critical path) and read back as broadcasts, each ray that composited the surfel
// fetches the per-hit state the forward stored (transmittance before the hit, the ten prefix sums after it), evaluates its gradient
// independently of every other hit, and the 63 gradient words (48 SH + 15 geometry) are transpose-reduced over the wavefront into ONE
// 256 B record per (batch, surfel) written by the 64 lanes as one coalesced line pair.  ~27x fewer records than one per hit, no
// per-hit gathers of surfel data, no dependent chain along the ray, no atomics.
constexpr int BS_GROUP = 16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RECW = 64;      // floats per (batch, surfel) gradient record: 48 SH (or 3 colour) + 15 geometry + pad
__global__ void __attribute__((amdgpu_waves_per_eu(2, 2))) __launch_bounds__(64)
batch_surfel_bwd(const TraceArgs A)
{
    __shared__ float4 sdat[2][BS_GROUP][16];               // per entry: surfel record (4 x 16 B) + SH block (12 x 16 B)
    __shared__ unsigned long long sdesc[2][BS_GROUP];      // sid | (hits-1) << 24 ; record index << 32
    __shared__ unsigned short kmat[2][BS_GROUP][64];       // per entry and ray: list position of the hit + 1, 0 = the ray did not blend it
    __shared__ unsigned spb[2][BS_GROUP];                  // per entry: index of its first pair
    __shared__ unsigned scn[2][BS_GROUP];                  // per entry: hits
    __shared__ float btile[16][64];                        // B operand of the reduction MFMAs: 16 words per ray, swizzled (see below)
    const int lane = threadIdx.x;
    const int nb = (A.D + 1) * (A.D + 1);
    const size_t region = (size_t)64 * A.cap;
    const int nbatch = (A.R + 63) >> 6;
    for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const int base = batch << 6;
        const int copy = batch & (NCOPY - 1);
        const int r = ray_of(A, base + lane);
        const bool valid = r < A.R && A.hit_cnt[r < A.R ? r : 0] <= A.cap;
        const int rr = r < A.R ? r : 0;
        // Per-ray constants.  The suffix terms of dL/dalpha only ever appear as  sum_j g_j (final_j - prefix_j)  (+ the background term), so
        // the twelve final sums fold into ONE scalar F = sum_j g_j final_j + T_final (bg . g_rgb): 17 live registers instead of 33.
        float basis[16], Box, Boy, Boz, Bdx, Bdy, Bdz, gR0, gR1, gR2, gD, gA, gN0, gN1, gN2, gX0, gX1, Fsum;
        {
            BwdRay B;
            bwd_load_ray(A, rr, B);
#pragma unroll
            for (int k = 0; k < 16; k++) basis[k] = 0.f;
            sh_basis(A.D, B.ux, B.uy, B.uz, basis);
            if (A.M == 0) basis[0] = kC0;
            Box = B.ox; Boy = B.oy; Boz = B.oz; Bdx = B.dx; Bdy = B.dy; Bdz = B.dz;
            gR0 = B.gR0; gR1 = B.gR1; gR2 = B.gR2; gD = B.gD; gA = B.gA; gN0 = B.gN0; gN1 = B.gN1; gN2 = B.gN2; gX0 = B.gX0; gX1 = B.gX1;
            Fsum = B.gR0 * B.fr0 + B.gR1 * B.fr1 + B.gR2 * B.fr2 + B.gD * B.fD + B.gA * B.fA + B.gN0 * B.fN0 + B.gN1 * B.fN1 + B.gN2 * B.fN2 +
                   B.gX0 * B.fX0 + B.gX1 * B.fX1 + B.fT * B.bgdot;
        }
        // A operand of the reduction MFMAs, constant for the batch: lane l holds basis_{l & 15} of ray 4s + (l >> 4)
        float Areg[16];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) btile[k][(lane + 2 * k) & 63] = (valid && k < nb) ? basis[k] : (k == 0 ? kC0 : 0.f);
        __syncthreads();
#pragma unroll
        for (int sI = 0; sI < 16; sI++) Areg[sI] = btile[lane & 15][(4 * sI + (lane >> 4) + 2 * (lane & 15)) & 63];
        __syncthreads();
        float Sk[16], dO0 = 0.f, dO1 = 0.f, dO2 = 0.f, dD0 = 0.f, dD1 = 0.f, dD2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) Sk[k] = 0.f;
        const int sstr = A.has_others ? 3 : 2;
        const float4 *state = A.state + (size_t)rr * A.cap * sstr;
        const unsigned long long *ent = A.entries + (size_t)batch * region;
        const unsigned *prs = A.pairs + (size_t)batch * region;
        const int D = A.n_entries[2 * batch], NE = D + A.n_entries[2 * batch + 1];
        unsigned poff = 0u;                                  // pairs of the table entries staged so far
        // Stage one group of entries, a whole group ahead of its use: 4 lanes per entry fetch the surfel record and SH block, then the
        // group's (lane, k) pairs (one contiguous run) are scattered into kmat -- so the main loop touches no global memory except
        // each ray's per-hit state.
        auto stage = [&](int g, int buf) {
            const int el = lane >> 2, part = lane & 3;
            const int e = g * BS_GROUP + el;
            unsigned long long d = 0ull;
            if (e < NE) {
                d = e < D ? ent[e] : ent[region - 1 - (size_t)(e - D)];
                const int sid = (int)(d & 0xFFFFFFull);
                sdat[buf][el][part] = A.srec[(size_t)sid * 4 + part];
                if (A.M == 16) {
                    const float4 *s4 = reinterpret_cast<const float4 *>(A.shs + (size_t)sid * 48);
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        float4 x = s4[part * 3 + q];
                        const int i0 = (part * 3 + q) * 4;             // words beyond the active degree are staged as zeros: the entry
                        if (i0 + 0 >= nb * 3) x.x = 0.f;               // loop then needs no degree checks
                        if (i0 + 1 >= nb * 3) x.y = 0.f;
                        if (i0 + 2 >= nb * 3) x.z = 0.f;
                        if (i0 + 3 >= nb * 3) x.w = 0.f;
                        sdat[buf][el][4 + part * 3 + q] = x;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int idx = (part * 3 + q) * 4 + c;
                            v[c] = A.M > 0 ? (idx < nb * 3 ? A.shs[(size_t)sid * A.M * 3 + idx] : 0.f) : (idx < 3 ? A.colors[(size_t)sid * 3 + idx] : 0.f);
                        }
                        sdat[buf][el][4 + part * 3 + q] = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
                if (part == 0) {
                    const size_t ci = (size_t)sid * NCOPY + copy;
                    const unsigned rec = A.surf_off[ci] - A.surf_cnt[ci] + (unsigned)(d >> 32);
                    sdesc[buf][el] = (d & 0x3FFFFFFFull) | ((unsigned long long)rec << 32);
                }
            }
            if (part == 0) scn[buf][el] = e < NE ? (unsigned)((d >> 24) & 63ull) + 1u : 0u;
            {   // clear kmat[buf]: 2 KB = 32 B per lane
                uint4 *km = reinterpret_cast<uint4 *>(&kmat[buf][0][0]);
                km[lane * 2] = make_uint4(0u, 0u, 0u, 0u); km[lane * 2 + 1] = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
            // prefix of the hit counts (every lane reads the 16 counts as broadcasts) and each entry's first pair
            unsigned pref[BS_GROUP + 1];
            pref[0] = 0u;
#pragma unroll
            for (int q = 0; q < BS_GROUP; q++) pref[q + 1] = pref[q] + scn[buf][q];
            unsigned tab_before = 0u;                        // hits of the group's TABLE entries before entry q (singles live elsewhere)
#pragma unroll
            for (int q = 0; q < BS_GROUP; q++) {
                const int eq = g * BS_GROUP + q;
                if (lane == q) spb[buf][q] = eq < D ? poff + tab_before : (unsigned)(region - 1 - (size_t)(eq - D));
                if (eq < D) tab_before += scn[buf][q];
            }
            __syncthreads();
            const unsigned total = pref[BS_GROUP];
            for (unsigned q = lane; q < total; q += 64) {
                int eli = 0;
#pragma unroll
                for (int t = 1; t < BS_GROUP; t++) eli += q >= pref[t] ? 1 : 0;
                const unsigned pr = prs[spb[buf][eli] + (q - pref[eli])];
                kmat[buf][eli][pr >> 16] = (unsigned short)((pr & 0xFFFFu) + 1u);
            }
            poff += tab_before;
        };
        __syncthreads();
        stage(0, 0);
        for (int g = 0; g * BS_GROUP < NE; g++) {
            const int buf = g & 1;
            __syncthreads();                                   // group g staged; group g-1 fully consumed
            if ((g + 1) * BS_GROUP < NE) stage(g + 1, buf ^ 1);
            const int ne = min(BS_GROUP, NE - g * BS_GROUP);
            // software pipeline over the entries: the per-hit state of entry el+1 is in flight while entry el is evaluated
            int k1 = valid ? (int)kmat[buf][0][lane] : 0;
            float4 st0, st1, st2 = make_float4(0.f, 0.f, 0.f, 0.f);
            { const float4 *sp = k1 > 0 ? state + (size_t)(k1 - 1) * sstr : A.state; st0 = sp[0]; st1 = sp[1]; if (A.has_others) st2 = sp[2]; }      // unconditional (idle lanes share one address): no branch, no wait
            for (int el = 0; el < ne; el++) {
                const unsigned long long d = sdesc[buf][el];
                const int sid = (int)(d & 0xFFFFFFull);
                const unsigned long long rec = d >> 32;
                const bool act = k1 > 0;
                // this ray's 16 B-matrix words -- dL/dcolour (3) and the first 13 geometry words -- go straight to the LDS tile (zeros from
                // rays that did not blend this surfel); the last two geometry words are summed with DPP.  Tile layout: word n of ray j at
                // n*64 + ((j + 2n) & 63): conflict-free both for these writes (fixed n, 64 rays) and for the MFMA operand reads (16 words
                // x 4 rays).  One wavefront per workgroup: its LDS operations execute in program order, so no barrier is needed -- and a
                // barrier's vmcnt(0) would drain the state prefetch that is in flight.
                float g13 = 0.f, g14 = 0.f;
#define BT(n) btile[(n)][(lane + 2 * (n)) & 63]
                if (act) {
                    const float4 s0 = sdat[buf][el][0], s1 = sdat[buf][el][1], s2 = sdat[buf][el][2], s3 = sdat[buf][el][3];
                    const SurfHit h = hit_surfel(s0, s1, s2, s3, Box, Boy, Boz, Bdx, Bdy, Bdz);
                    float col[3]; bool cl[3] = {false, false, false};
                    if (A.M > 0) {
                        float rc[3] = {0.f, 0.f, 0.f};
#pragma unroll
                        for (int q4 = 0; q4 < 12; q4++) {
                            const float4 x = sdat[buf][el][4 + q4];
                            const float xe[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                            for (int e = 0; e < 4; e++) { const int idx = 4 * q4 + e; rc[idx % 3] += basis[idx / 3] * xe[e]; }
                        }
#pragma unroll
                        for (int c = 0; c < 3; c++) { const float v = rc[c] + 0.5f; cl[c] = v < 0.f; col[c] = cl[c] ? 0.f : v; }
                    } else { const float4 x = sdat[buf][el][4]; col[0] = x.x; col[1] = x.y; col[2] = x.z; }
                    const float x0 = A.has_others ? A.others[2 * sid] : 0.f, x1 = A.has_others ? A.others[2 * sid + 1] : 0.f;
                    const float alpha = h.alpha, Tb = st0.x;
                    const float w = alpha * Tb;
                    const float sgn = h.denom < 0.0f ? 1.0f : -1.0f;
                    const float nf0 = sgn * s3.x, nf1 = sgn * s3.y, nf2 = sgn * s3.z;
                    const float inv1m = __builtin_amdgcn_rcpf(1.0f - alpha);          // v_rcp_f32 (1 ulp): gradient-only terms need no IEEE division
                    const float gv_ = gR0 * col[0] + gR1 * col[1] + gR2 * col[2] + gD * h.t + gA + gN0 * nf0 + gN1 * nf1 + gN2 * nf2 + gX0 * x0 + gX1 * x1;
                    const float gS = gR0 * st0.y + gR1 * st0.z + gR2 * st0.w + gD * st1.x + gA * (1.0f - Tb * (1.0f - alpha)) + gN0 * st1.y + gN1 * st1.z +
                                     gN2 * st1.w + gX0 * st2.x + gX1 * st2.y;
                    const float dLa = Tb * gv_ - (Fsum - gS) * inv1m;
                    const float dc[3] = {cl[0] ? 0.f : w * gR0, cl[1] ? 0.f : w * gR1, cl[2] ? 0.f : w * gR2};
                    if (A.M > 0) {
#pragma unroll
                        for (int q4 = 0; q4 < 12; q4++) {
                            const float4 x = sdat[buf][el][4 + q4];
                            const float xe[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                            for (int e = 0; e < 4; e++) { const int idx = 4 * q4 + e; Sk[idx / 3] += xe[e] * dc[idx % 3]; }
                        }
                    }
                    if (A.has_others && A.dothers) { atomic_add_f32(A.dothers + 2 * sid, w * gX0); atomic_add_f32(A.dothers + 2 * sid + 1, w * gX1); }
                    const float dLG = s0.w * dLa;
                    const float dLu = dLG * (-h.G * h.u), dLv = dLG * (-h.G * h.v);
                    const float isu = __builtin_amdgcn_rcpf(s1.w), isv = __builtin_amdgcn_rcpf(s2.w);
                    const float qx = Box + h.t * Bdx - s0.x, qy = Boy + h.t * Bdy - s0.y, qz = Boz + h.t * Bdz - s0.z;
                    const float dq0 = dLu * s1.x + dLv * s2.x, dq1 = dLu * s1.y + dLv * s2.y, dq2 = dLu * s1.z + dLv * s2.z;
                    const float cu = dLu * isu, cv = dLv * isv;
                    const float dLt_tot = w * gD + dq0 * Bdx + dq1 * Bdy + dq2 * Bdz;
                    const float kt = dLt_tot * __builtin_amdgcn_rcpf(h.denom);
                    BT(0) = dc[0]; BT(1) = dc[1]; BT(2) = dc[2];
                    const float e0 = dq0 - kt * s3.x, e1 = dq1 - kt * s3.y, e2 = dq2 - kt * s3.z;
                    BT(3) = -e0; BT(4) = -e1; BT(5) = -e2;
                    BT(6) = cu * qx; BT(7) = cu * qy; BT(8) = cu * qz;
                    BT(9) = cv * qx; BT(10) = cv * qy; BT(11) = cv * qz;
                    const float ws = w * sgn;
                    BT(12) = ws * gN0 - kt * qx; BT(13) = ws * gN1 - kt * qy; BT(14) = ws * gN2 - kt * qz;
                    BT(15) = -cu * h.u * A.mod;
                    g13 = -cv * h.v * A.mod;
                    g14 = h.G * dLa;
                    dO0 += e0; dO1 += e1; dO2 += e2;
                    dD0 += h.t * e0; dD1 += h.t * e1; dD2 += h.t * e2;
                } else {
#pragma unroll
                    for (int n = 0; n < 16; n++) BT(n) = 0.f;
                }
#undef BT
                if (el + 1 < ne) {                           // next entry's state: in flight during the reduction below
                    k1 = valid ? (int)kmat[buf][el + 1][lane] : 0;
                    const float4 *sp = k1 > 0 ? state + (size_t)(k1 - 1) * sstr : A.state;
                    st0 = sp[0]; st1 = sp[1]; if (A.has_others) st2 = sp[2];
                }
                // Sum over the 64 rays on the matrix cores: D[16 x 16] = basis^T[16 x 64 rays] . B[64 rays x 16], K = 64 in 16 exact-f32
                // MFMAs (four independent chains: the dependent latency is 40 cycles).  Columns 0-2 are the (16,3) SH gradient block;
                // basis_0 is the constant C0 for every ray, so row 0 of the other 13 columns is C0 x (the plain sum of a geometry word).
                f32x4 acc4 = {0.f, 0.f, 0.f, 0.f}, accB = acc4, accC = acc4, accD = acc4;
                {
                    const int n = lane & 15, j = lane >> 4;
                    const float *brow = &btile[n][0];
                    const int rot = 2 * n + j;
#pragma unroll
                    for (int sI = 0; sI < 16; sI += 4) {
                        acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI], brow[(4 * sI + rot) & 63], acc4, 0, 0, 0);
                        accB = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI + 1], brow[(4 * sI + 4 + rot) & 63], accB, 0, 0, 0);
                        accC = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI + 2], brow[(4 * sI + 8 + rot) & 63], accC, 0, 0, 0);
                        accD = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI + 3], brow[(4 * sI + 12 + rot) & 63], accD, 0, 0, 0);
                    }
                }
                acc4 = (acc4 + accB) + (accC + accD);
                const float s13 = wave_sum(g13), s14 = wave_sum(g14);
                if (rec < A.num_records) {
                    float *ro = A.records + rec * RECW;
                    const int n = lane & 15, mrow = (lane >> 4) * 4;
                    if (A.M > 0) {
                        if (n < 3) { ro[(mrow + 0) * 3 + n] = acc4[0]; ro[(mrow + 1) * 3 + n] = acc4[1]; ro[(mrow + 2) * 3 + n] = acc4[2]; ro[(mrow + 3) * 3 + n] = acc4[3]; }
                    } else if (lane < 3) ro[lane] = acc4[0] * (1.0f / kC0);
                    if (lane >= 3 && lane < 16) ro[48 + lane - 3] = acc4[0] * (1.0f / kC0);
                    if (lane == 0) { ro[61] = s13; ro[62] = s14; }
                }
            }
        }
        if (valid) {
            BwdRay B;
            bwd_load_ray(A, r, B);
            BwdAcc acc;
            bwd_init_acc(acc);
            acc.dO0 = dO0; acc.dO1 = dO1; acc.dO2 = dO2; acc.dD0 = dD0; acc.dD1 = dD1; acc.dD2 = dD2;
#pragma unroll
            for (int k = 0; k < 16; k++) acc.Sk[k] = Sk[k];
            bwd_store_ray(A, r, B, acc);
        }
    }
}

// Stage 2: sum each surfel's (batch, surfel) records into the (zeroed) gradient buffers -- plain stores, every word has one owner; the
// K-buffer pass for overflowed rays runs afterwards and adds to the same buffers atomically.  16 lanes per surfel, 16 B per lane = one
// 256 B record per load instruction; the typical surfel has ~15 records, but a few are seen by thousands of batches: those are deferred
// and summed by the whole workgroup (16 records per instruction) so that no lane group walks a megabyte on its own.
constexpr int RED_LONG = 96;
__device__ __forceinline__ void reduce_store(const TraceArgs &A, const int sid, const int q, const int nb, const float *v)
{
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int wd = 4 * q + e;
        if (wd < 48) {
            if (A.M > 0) { if (wd / 3 < nb) A.dshs[(size_t)sid * A.M * 3 + wd] = v[e]; }
            else if (wd < 3) A.dcolors[(size_t)sid * 3 + wd] = v[e];
        } else if (wd < 63) A.geo_rec[(size_t)sid * GEO + (wd - 48)] = v[e];
    }
}
__global__ void __launch_bounds__(256)
reduce_surfel_records(const TraceArgs A)
{
    __shared__ int longs[64];
    __shared__ int nlong;
    __shared__ float4 part[16][16];
    const int sub = threadIdx.x >> 4, q = threadIdx.x & 15;               // 16 surfels per workgroup
    const int nb = (A.D + 1) * (A.D + 1);
    const float4 *rp = reinterpret_cast<const float4 *>(A.records) + q;
    if (threadIdx.x == 0) nlong = 0;
    __syncthreads();
    for (int sid0 = blockIdx.x * 16; sid0 < A.P; sid0 += gridDim.x * 16) {
        const int sid = sid0 + sub;
        if (sid >= A.P) continue;
        const unsigned end = A.surf_off[(size_t)sid * NCOPY + NCOPY - 1];
        const unsigned begin = A.surf_off[(size_t)sid * NCOPY] - A.surf_cnt[(size_t)sid * NCOPY];     // the NCOPY sub-segments are adjacent
        if (end <= begin) continue;
        if (end - begin > (unsigned)RED_LONG) {
            int k = 0;
            if (q == 0) k = atomicAdd(&nlong, 1);
            k = __shfl(k, 0, 16);
            if (k < 64) { if (q == 0) longs[k] = sid; continue; }          // (list full: fall through and do it the slow way)
        }
        const unsigned long long hi = end < A.num_records ? end : A.num_records;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        unsigned long long i = begin;
        for (; i + 4 <= hi; i += 4) {
            const float4 x0 = rp[i * 16], x1 = rp[(i + 1) * 16], x2 = rp[(i + 2) * 16], x3 = rp[(i + 3) * 16];
            a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w; a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
            a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w; a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
        }
        for (; i < hi; i++) { const float4 x0 = rp[i * 16]; a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w; }
        const float v[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w)};
        reduce_store(A, sid, q, nb, v);
    }
    __syncthreads();
    const int nl = nlong < 64 ? nlong : 64;
    for (int k = 0; k < nl; k++) {
        const int sid = longs[k];
        const unsigned end = A.surf_off[(size_t)sid * NCOPY + NCOPY - 1];
        const unsigned begin = A.surf_off[(size_t)sid * NCOPY] - A.surf_cnt[(size_t)sid * NCOPY];
        const unsigned long long hi = end < A.num_records ? end : A.num_records;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        unsigned long long i = (unsigned long long)begin + sub;
        for (; i + 16 < hi; i += 32) {
            const float4 x0 = rp[i * 16], x1 = rp[(i + 16) * 16];
            a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w; a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
        }
        if (i < hi) { const float4 x0 = rp[i * 16]; a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w; }
        __syncthreads();
        part[sub][q] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        __syncthreads();
        if (sub == 0) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 16; g++) { const float4 x = part[g][q]; v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w; }
            reduce_store(A, sid, q, nb, v);
        }
    }
}

// per-surfel geometry record [dmu 3, da 3, db 3, dn 3, dsu, dsv, dopacity] -> parameter gradients; the rotation columns
// (a,b,n) chain to the unit quaternion; dmeans is also copied into the densification sink.
__global__ void __launch_bounds__(256)
finish_surfel_grads(int P, const float *__restrict__ rots, const float *__restrict__ geo_rec, float *__restrict__ dmeans,
                    float *__restrict__ dscales, float *__restrict__ dopac, float *__restrict__ drots, float *__restrict__ dgrads3D)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float q0 = rots[4 * i], q1 = rots[4 * i + 1], q2 = rots[4 * i + 2], q3 = rots[4 * i + 3];
    const float inv = 1.0f / sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
    const float r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
    const float *g = geo_rec + (size_t)i * GEO;
    const float *rr = g + 3;
    // V[row][col]: col 0 = dL/da, col 1 = dL/db, col 2 = dL/dn
#define VR(a, b) rr[(b) * 3 + (a)]
    drots[4 * i + 0] = 2.f * (x * (VR(2, 1) - VR(1, 2)) + y * (VR(0, 2) - VR(2, 0)) + z * (VR(1, 0) - VR(0, 1)));
    drots[4 * i + 1] = 2.f * (-2.f * x * (VR(1, 1) + VR(2, 2)) + y * (VR(1, 0) + VR(0, 1)) + z * (VR(2, 0) + VR(0, 2)) + r * (VR(2, 1) - VR(1, 2)));
    drots[4 * i + 2] = 2.f * (x * (VR(1, 0) + VR(0, 1)) - 2.f * y * (VR(0, 0) + VR(2, 2)) + z * (VR(2, 1) + VR(1, 2)) + r * (VR(0, 2) - VR(2, 0)));
    drots[4 * i + 3] = 2.f * (x * (VR(2, 0) + VR(0, 2)) + y * (VR(2, 1) + VR(1, 2)) - 2.f * z * (VR(0, 0) + VR(1, 1)) + r * (VR(1, 0) - VR(0, 1)));
#undef VR
    dmeans[3 * i] = g[0]; dmeans[3 * i + 1] = g[1]; dmeans[3 * i + 2] = g[2];
    dscales[2 * i] = g[12]; dscales[2 * i + 1] = g[13];
    dopac[i] = g[14];
    if (dgrads3D) { dgrads3D[3 * i] = g[0]; dgrads3D[3 * i + 1] = g[1]; dgrads3D[3 * i + 2] = g[2]; }
}

static int persistent_grid(int R, int per_cu = 8)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int want = (R + 63) / 64;
    const int cap = cus * per_cu;
    return want < cap ? (want > 0 ? want : 1) : cap;
}

// grid for a grid-stride kernel that handles `per_block` rays per block iteration
static int stride_grid(int R, int per_block)
{
    const int want = (R + per_block - 1) / per_block;
    const int cap = 256 * 32;
    const int g = want < cap ? (want > 0 ? want : 1) : cap;
    return (g + 7) & ~7;                 // multiple of 8: xcd_block() needs it
}

// The list path packs surfel ids, ray slots and per-surfel entry counts into 24-bit fields: beyond 2^24 surfels or rays both directions take
// the K-buffer kernels (correct at any size, slower).
static bool lists_usable(const envgs_trace_cfg *cfg, const envgs_trace_lists *L)
{
    return L && L->cap > 0 && cfg->max_trace_depth == 0 && cfg->P > 0 && cfg->P < (1 << 24) && cfg->num_rays < (1 << 24) && L->hit_lists &&
           L->hit_cnt && L->n_used && L->stack_spill && L->surf_cnt && L->surf_off && L->surf_acc && L->scan_temp;
}

}  // namespace envgs

using namespace envgs;

static void ray_layout(const envgs_trace_cfg *cfg, int *rh, int *rw)
{
    const bool ok = cfg->ray_h > 0 && cfg->ray_w > 0 && (long long)cfg->ray_h * cfg->ray_w == cfg->num_rays;
    *rh = ok ? cfg->ray_h : 0;
    *rw = ok ? cfg->ray_w : 0;
}

extern "C" {

size_t envgs_trace_ray_sort_temp_bytes(int32_t num_rays)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned *)nullptr, (unsigned *)nullptr,
                                    (size_t)(num_rays > 0 ? num_rays : 1), 0u, 31u);
    return bytes;
}

// one slab per persistent wavefront of the collection kernels, for each of the (at most two) batch segments that run concurrently
size_t envgs_trace_stack_spill_ints(int32_t num_rays) { return (size_t)2 * persistent_grid(num_rays, 24) * STACK * 64; }

int envgs_trace_forward(const envgs_trace_cfg *cfg, const float *nodes, const float *ray_o, const float *ray_d,
                        const float *means3D, const float *scales, const float *rotations, const float *opacities,
                        const float *shs, const float *colors_precomp, const float *others_precomp, const float *bg,
                        float *srec, uint32_t *counters, float *rgb, float *dpt, float *acc, float *norm, float *dist,
                        float *aux, float *mid, float *wet, float *final_T, const envgs_trace_lists *L, void *stream_)
{
    if (!cfg || cfg->P < 0 || cfg->num_rays < 0 || cfg->sh_degree < 0 || cfg->sh_degree > 3 || cfg->max_trace_depth < 0 || cfg->max_trace_depth > 7)
        return ENVGS_ERR_BAD_ARG;
    if (cfg->num_rays == 0) return 0;
    if (!ray_o || !ray_d || !bg || !counters || !rgb || !dpt || !acc || !norm || !dist || !aux || !mid || !final_T) return ENVGS_ERR_BAD_ARG;
    if (cfg->P > 0 && (!nodes || !means3D || !scales || !rotations || !opacities || !srec || !wet)) return ENVGS_ERR_BAD_ARG;
    if (cfg->P > 0 && (cfg->sh_coeffs > 0 ? (!shs || cfg->sh_coeffs < (cfg->sh_degree + 1) * (cfg->sh_degree + 1)) : !colors_precomp)) return ENVGS_ERR_BAD_ARG;
    if (cfg->has_others && cfg->P > 0 && !others_precomp) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    envgs_raster_cfg dbg; dbg.debug = cfg->debug;
    const envgs_raster_cfg *dcfg = &dbg;
    hipError_t e = hipMemsetAsync(counters, 0, 96 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    if (cfg->P > 0) {
        e = hipMemsetAsync(wet, 0, sizeof(float) * (size_t)cfg->P, stream);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(make_surfel_records, dim3((cfg->P + 255) / 256), dim3(256), 0, stream, cfg->P, cfg->scale_modifier,
                           means3D, scales, rotations, opacities, srec);
        ENVGS_CHECK_LAUNCH(dcfg, stream);
    }
    e = hipMemsetAsync(mid, 0, sizeof(float) * (size_t)cfg->num_rays * MID * (cfg->max_trace_depth + 1), stream);
    if (e != hipSuccess) return (int)e;
    TraceArgs A;
    A = TraceArgs{};
    A.P = cfg->P; A.R = cfg->num_rays; A.D = cfg->sh_degree; A.M = cfg->sh_coeffs; A.ND = cfg->max_trace_depth + 1;
    A.start_from_first = cfg->start_from_first; A.has_others = cfg->has_others; A.bg_len = cfg->bg_len; A.spec_thr = cfg->specular_threshold;
    A.nodes = (const float4 *)nodes; A.srec = (const float4 *)srec; A.shs = shs; A.colors = colors_precomp; A.others = others_precomp;
    A.bg = bg; A.ray_o = ray_o; A.ray_d = ray_d; A.counter = counters; A.stats = (unsigned long long *)(counters + 2);
    A.rgb = rgb; A.dpt = dpt; A.acc = acc; A.norm = norm; A.dist = dist; A.aux = aux; A.mid = mid; A.wet = wet; A.final_T = final_T;
    A.mod = cfg->scale_modifier;
    { const char *ev = getenv("ENVGS_TRACE_EXP"); A.exp = ev ? atoi(ev) : 0; }
    int rh, rw; ray_layout(cfg, &rh, &rw);
    const bool lists = lists_usable(cfg, L);
    if (L && L->cap > SORT_MAX) return ENVGS_ERR_BAD_ARG;
    ProfScope prof_(K_TRACE_FWD, stream);
    if (lists) {
        if (L->scan_temp_bytes < scan_temp_bytes(cfg->P * NCOPY)) return ENVGS_ERR_TEMP_TOO_SMALL;
        A.hits = (uint2 *)L->hit_lists; A.hit_cnt = L->hit_cnt; A.n_used = L->n_used; A.cap = L->cap; A.stack_spill = L->stack_spill;
        A.surf_cnt = L->surf_cnt; A.surf_off = L->surf_off; A.surf_acc = (unsigned long long *)L->surf_acc;
        if (L->ray_keys && L->ray_order && L->ray_sort_temp && !(A.exp & 64)) {
            // coherence sort of the rays (keys / values double-buffered in ray_keys / ray_order: 2R words each)
            const int R = cfg->num_rays;
            hipLaunchKernelGGL(make_ray_keys, dim3((R + 255) / 256), dim3(256), 0, stream, R, ray_o, ray_d, A.nodes, cfg->P, L->ray_keys, L->ray_order);
            ENVGS_CHECK_LAUNCH(dcfg, stream);
            size_t tb = L->ray_sort_temp_bytes;
            e = rocprim::radix_sort_pairs(L->ray_sort_temp, tb, L->ray_keys, L->ray_keys + R, L->ray_order, L->ray_order + R, (size_t)R, 0u, 31u, stream);
            if (e != hipSuccess) return (int)e;
            A.order = L->ray_order + R;
        }
        {   // 40-bit fixed-point weight: enough integer bits that even a surfel seen with w = 1 by every ray cannot overflow
            int ib = 1;
            while ((1ll << ib) <= (long long)cfg->num_rays) ib++;
            A.wfrac = 40 - ib > 30 ? 30 : 40 - ib;
        }
        e = hipMemsetAsync(L->surf_acc, 0, sizeof(unsigned long long) * (size_t)cfg->P * NCOPY, stream);
        if (e != hipSuccess) return (int)e;
        A.state = (float4 *)L->hit_state; A.entries = (unsigned long long *)L->entries; A.pairs = L->pairs; A.n_entries = L->n_entries;
        // The ray batches are split into two segments that run collect -> sort+composite -> register on two streams: the collection
        // kernel is a persistent grid whose wavefronts drain over the time of one whole batch, and the second segment's wavefronts
        // (and the first segment's next kernel) move into the CUs it leaves idle.
        const int nbatch_all = (cfg->num_rays + 63) / 64;
        int nseg = 2;                                         // measured: 1 -> 18.3 ms / step, 2 -> 17.5, 4 -> 19.4 (each collection launch lasts at least one batch)
        { const char *sv = getenv("ENVGS_SEGMENTS"); if (sv) nseg = atoi(sv); }
        if (nseg > 2) nseg = 2;                               // (the stack-spill slab and the fetch counters are sized for two)
        while (nseg > 1 && nbatch_all / nseg < 256) nseg >>= 1;
        if (nseg < 1) nseg = 1;
        hipStream_t aux = nullptr;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        if (nseg > 1) {
            // one auxiliary stream + fork / join events per device, created on first use (one process drives one GPU in this design,
            // but nothing here assumes it)
            static hipStream_t s_aux[16] = {};
            static hipEvent_t s_fork[16] = {}, s_join[16] = {};
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) nseg = 1;
            else if (!s_aux[dev]) {
                if (hipStreamCreateWithFlags(&s_aux[dev], hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&s_fork[dev], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&s_join[dev], hipEventDisableTiming) != hipSuccess) { s_aux[dev] = nullptr; nseg = 1; }
            }
            if (nseg > 1) {
                aux = s_aux[dev]; ev_fork = s_fork[dev]; ev_join = s_join[dev];
                if (hipEventRecord(ev_fork, stream) != hipSuccess || hipStreamWaitEvent(aux, ev_fork, 0) != hipSuccess) return ENVGS_ERR_BAD_ARG;
            }
        }
        for (int sg = 0; sg < nseg; sg++) {                   // even segments on the caller's stream, odd ones on the auxiliary stream
            hipStream_t st = (sg & 1) ? aux : stream;
            TraceArgs S = A;
            S.seg = sg;
            S.spill_stride = persistent_grid(cfg->num_rays, 24);     // >= this segment's grid; matches envgs_trace_stack_spill_ints
            S.batch0 = (int)((long long)nbatch_all * sg / nseg);
            S.batch1 = (int)((long long)nbatch_all * (sg + 1) / nseg);
            const int rays_seg = (S.batch1 - S.batch0) * 64;
            {
                ProfScope p1(K_TRACE_COLLECT, st);
                if (S.order && !(S.exp & 512) && !(S.exp & 16))
                    hipLaunchKernelGGL(collect_hits_packet4, dim3(persistent_grid(rays_seg, 24)), dim3(64), 0, st, S, S.nodes,
                                       S.nodes + (size_t)(cfg->P > 1 ? cfg->P - 1 : 1) * 4, S.srec);
                else if (S.order && !(S.exp & 512))
                    hipLaunchKernelGGL(collect_hits_packet, dim3(persistent_grid(rays_seg, 24)), dim3(64), 0, st, S, S.nodes, S.srec);
                else
                    hipLaunchKernelGGL(collect_hits, dim3(persistent_grid(rays_seg, 24)), dim3(64), 0, st, S);
            }
            ENVGS_CHECK_LAUNCH(dcfg, st);
            {
                ProfScope p2(K_TRACE_SORT, st);
                const dim3 g(stride_grid(rays_seg, 4)), b(256);
                hipLaunchKernelGGL((sort_composite_fwd<4, false>), g, b, 0, st, S);
                if (S.cap > 256) {
                    const dim3 gl(min(stride_grid(rays_seg, 256), 512));
                    if (S.cap <= 512) hipLaunchKernelGGL((sort_composite_fwd<8, true>), gl, b, 0, st, S);
                    else hipLaunchKernelGGL((sort_composite_fwd<16, true>), gl, b, 0, st, S);
                }
            }
            ENVGS_CHECK_LAUNCH(dcfg, st);
            { ProfScope p8(K_TRACE_REGISTER, st); hipLaunchKernelGGL(register_hits, dim3(stride_grid(rays_seg, 64)), dim3(64 * RH_W), 0, st, S); }
            ENVGS_CHECK_LAUNCH(dcfg, st);
        }
        if (nseg > 1) {
            if (hipEventRecord(ev_join, aux) != hipSuccess || hipStreamWaitEvent(stream, ev_join, 0) != hipSuccess) return ENVGS_ERR_BAD_ARG;
        }
        hipLaunchKernelGGL(unpack_surfel_acc, dim3((cfg->P + 255) / 256), dim3(256), 0, stream, cfg->P, A.wfrac, A.surf_acc, L->surf_cnt, wet);
        ENVGS_CHECK_LAUNCH(dcfg, stream);
        {   // records of the backward are addressed through the inclusive scan of the per-surfel hit counts
            const int rc = launch_scan(L->surf_cnt, L->surf_off, cfg->P * NCOPY, L->scan_temp, L->scan_temp_bytes, stream);
            if (rc) return rc;
        }
        e = hipMemsetAsync(counters, 0, sizeof(uint32_t), stream);          // ray-fetch counter for the overflow pass
        if (e != hipSuccess) return (int)e;
        A.only_overflow = 1;
    }
    { ProfScope p4(K_TRACE_KBUF_FWD, stream); hipLaunchKernelGGL(trace_fwd, dim3(persistent_grid(cfg->num_rays)), dim3(64), 0, stream, A, rh, rw); }
    ENVGS_CHECK_LAUNCH(dcfg, stream);
    return 0;
}

int envgs_trace_backward(const envgs_trace_cfg *cfg, const float *nodes, const float *ray_o, const float *ray_d,
                         const float *means3D, const float *scales, const float *rotations, const float *opacities,
                         const float *shs, const float *colors_precomp, const float *others_precomp, const float *bg,
                         const float *srec, uint32_t *counters, const float *rgb, const float *dpt, const float *acc,
                         const float *norm, const float *aux, const float *final_T, const float *dL_drgb, const float *dL_ddpt,
                         const float *dL_dacc, const float *dL_dnorm, const float *dL_daux, float *geo_rec, float *dmeans3D,
                         float *dgrads3D, float *dscales, float *drots, float *dopacities, float *dshs, float *dcolors,
                         float *dothers, float *dray_o, float *dray_d, const envgs_trace_lists *L, void *stream_)
{
    if (!cfg || cfg->P < 0 || cfg->num_rays < 0) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    envgs_raster_cfg dbg; dbg.debug = cfg->debug;
    const envgs_raster_cfg *dcfg = &dbg;
    const size_t P = (size_t)cfg->P, R = (size_t)cfg->num_rays;
    hipError_t e;
#define ZERO(ptr, n) do { if ((ptr) && (n) > 0) { e = hipMemsetAsync((ptr), 0, sizeof(float) * (n), stream); if (e != hipSuccess) return (int)e; } } while (0)
    ZERO(geo_rec, P * ENVGS_GEOREC_STRIDE); ZERO(dmeans3D, P * 3); ZERO(dgrads3D, P * 3); ZERO(dscales, P * 2); ZERO(drots, P * 4);
    ZERO(dopacities, P); ZERO(dothers, P * 2); ZERO(dray_o, R * 3); ZERO(dray_d, R * 3);
    if (cfg->sh_coeffs > 0) ZERO(dshs, P * cfg->sh_coeffs * 3); else ZERO(dcolors, P * 3);
#undef ZERO
    if (cfg->num_rays == 0 || cfg->P == 0) return 0;
    if (!nodes || !ray_o || !ray_d || !srec || !counters || !rgb || !dpt || !acc || !norm || !aux || !final_T || !dL_drgb || !dL_ddpt ||
        !dL_dacc || !dL_dnorm || !dL_daux || !geo_rec || !dmeans3D || !dscales || !drots || !dopacities || !dray_o || !dray_d || !rotations || !bg)
        return ENVGS_ERR_BAD_ARG;
    if (cfg->sh_coeffs > 0 ? (!shs || !dshs) : (!colors_precomp || !dcolors)) return ENVGS_ERR_BAD_ARG;
    e = hipMemsetAsync(counters, 0, sizeof(uint32_t), stream);      // only the ray-fetch counter: [1] (largest list) and the stats stay readable
    if (e != hipSuccess) return (int)e;
    TraceArgs A;
    A = TraceArgs{};
    A.P = cfg->P; A.R = cfg->num_rays; A.D = cfg->sh_degree; A.M = cfg->sh_coeffs; A.ND = 1;
    A.start_from_first = cfg->start_from_first; A.has_others = cfg->has_others; A.bg_len = cfg->bg_len; A.spec_thr = cfg->specular_threshold;
    A.nodes = (const float4 *)nodes; A.srec = (const float4 *)srec; A.shs = shs; A.colors = colors_precomp; A.others = others_precomp;
    A.bg = bg; A.ray_o = ray_o; A.ray_d = ray_d; A.counter = counters;
    A.f_rgb = rgb; A.f_dpt = dpt; A.f_acc = acc; A.f_norm = norm; A.f_aux = aux; A.f_T = final_T;
    A.g_rgb = dL_drgb; A.g_dpt = dL_ddpt; A.g_acc = dL_dacc; A.g_norm = dL_dnorm; A.g_aux = dL_daux;
    A.geo_rec = geo_rec; A.dshs = dshs; A.dcolors = dcolors;
    { const char *ev = getenv("ENVGS_TRACE_EXP"); A.exp = ev ? atoi(ev) : 0; }
    A.dothers = dothers; A.dray_o = dray_o; A.dray_d = dray_d; A.mod = cfg->scale_modifier;
    int rh, rw; ray_layout(cfg, &rh, &rw);
    {
        ProfScope prof_(K_TRACE_BWD, stream);
        if (lists_usable(cfg, L)) {                    // the same test as the forward: the lists exist exactly when it filled them
            A.hits = (uint2 *)L->hit_lists; A.hit_cnt = L->hit_cnt; A.n_used = L->n_used; A.cap = L->cap;
            if (L->ray_keys && L->ray_order && L->ray_sort_temp && !(A.exp & 64)) A.order = L->ray_order + cfg->num_rays;
            if (L->records && L->num_records > 0 && L->surf_cnt && L->surf_off && L->hit_state && L->entries && L->pairs && L->n_entries && !(A.exp & 8)) {
                // atomic-free: one record per (batch, surfel) entry, grouped by surfel; then each surfel's records are summed
                A.surf_cnt = L->surf_cnt; A.surf_off = L->surf_off; A.records = L->records; A.num_records = L->num_records;
                A.state = (float4 *)L->hit_state; A.entries = (unsigned long long *)L->entries; A.pairs = L->pairs; A.n_entries = L->n_entries;
                { ProfScope p5(K_TRACE_LIST_BWD, stream); hipLaunchKernelGGL(batch_surfel_bwd, dim3(stride_grid((cfg->num_rays + 63) / 64, 1)), dim3(64), 0, stream, A); }
                { ProfScope p7(K_TRACE_REDUCE, stream); hipLaunchKernelGGL(reduce_surfel_records, dim3(stride_grid(cfg->P, 16)), dim3(256), 0, stream, A); }
            } else {
                ProfScope p5(K_TRACE_LIST_BWD, stream);
                hipLaunchKernelGGL(composite_lists_bwd, dim3(stride_grid(cfg->num_rays, 64)), dim3(64), 0, stream, A);
            }
            A.only_overflow = 1;
        }
        { ProfScope p6(K_TRACE_KBUF_BWD, stream); hipLaunchKernelGGL(trace_bwd, dim3(persistent_grid(cfg->num_rays)), dim3(64), 0, stream, A, rh, rw); }
    }
    ENVGS_CHECK_LAUNCH(dcfg, stream);
    hipLaunchKernelGGL(finish_surfel_grads, dim3((cfg->P + 255) / 256), dim3(256), 0, stream, cfg->P, rotations, geo_rec, dmeans3D, dscales,
                       dopacities, drots, dgrads3D);
    ENVGS_CHECK_LAUNCH(dcfg, stream);
    return 0;
}

}  // extern "C"
