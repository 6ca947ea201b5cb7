// trace_lists.hip -- list path, steps 2 and 3: register-resident bitonic sort + lane-per-hit compositing of every ray's list, then the per-batch
// registration of the composited hits with their surfels (entries, pairs, per-surfel weights).
#include "trace_common.h"

#ifndef ENVGS_SCF_KO
#define ENVGS_SCF_KO 0          // measurement builds only (scratch/ab_bsb.sh): 1 = no plane-1 state store, 2 = no `others` gather / scans -- results wrong by construction
#endif
namespace envgs {

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

// 192 keys in three registers per lane: the 256-key network with its fourth block known to be +inf, which removes every operation on that
// block (a quarter of the cross-lane layers).  Blocks 0-2 are sorted as in the full network up to size 64; at size 128 the pair (2, inf)
// merges DOWNWARD -- the keys move to block 3, which only changes their direction: block 2 is sorted descending in place; at size 256
// block 0 meets the inf block (nothing to do), block 1 meets the keys of block 3, then (0, 1) and (keys, inf) at stride 64 -- the keys
// return to block 2 -- and the six cross-lane layers finish the three blocks.
__device__ __forceinline__ void bitonic_sort_192(unsigned long long (&k)[3], const int lane)
{
    auto ce = [](unsigned long long &a, unsigned long long &b) { const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; };
    bitonic_sort<3, 64>(k, lane);
    ce(k[0], k[1]);
    bitonic_merge<3, 128, 32>(k, lane);
    ce(k[1], k[2]);
    ce(k[0], k[1]);
    bitonic_merge<3, 256, 32>(k, lane);
}

// Quad-permuted copy of the (P,16,3) SH blocks for the cooperative colour evaluation below.  The 48 words of a block are split among the four
// lanes of a quad -- lane q owns coefficients 4q .. 4q+3, i.e. source words [12q, 12q+12) -- and stored so that load i (of three) of the four
// lanes is ONE contiguous, aligned 64 B run (32 B with fp16 storage): word 4i + j of lane q sits at (4i + q) * 4 + j.  Coefficients beyond the
// active degree are stored as zeros.  31 MB per step at the bench size: noise next to what it saves (see sort_composite_ray).
static __device__ void permute_sh_word(const size_t i, int P, int nb, int f16, const void *__restrict__ shs, void *__restrict__ shp)
{
    if (i >= (size_t)P * 48) return;
    const int sid = (int)(i / 48), pos = (int)(i % 48);
    const int piece = pos >> 2, j = pos & 3, ii = piece >> 2, q = piece & 3;
    const int src = 12 * q + 4 * ii + j;                       // = 3 k + c
    const bool live = src < nb * 3;
    if (f16) reinterpret_cast<__half *>(shp)[i] = live ? reinterpret_cast<const __half *>(shs)[(size_t)sid * 48 + src] : __float2half(0.f);
    else reinterpret_cast<float *>(shp)[i] = live ? reinterpret_cast<const float *>(shs)[(size_t)sid * 48 + src] : 0.f;
}

// ---- per-surfel record ----------------------------------------------------------------------------
static __device__ void surfel_record(const int i, const float mod, const float *__restrict__ means, const float *__restrict__ scales,
                                     const float *__restrict__ rots, const float *__restrict__ opac, float *__restrict__ srec)
{
#pragma clang fp contract(off)      // the frame feeds the hit distance t, a sort key that is bit-exact against the oracle (see hit_surfel)
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

// What a forward needs before its first traversal, in ONE launch (round 4; three until then): the zero fill of the call's counters and
// accumulators, the per-surfel records, and (list path, SH colours) the quad-permuted copy of the SH blocks.  Block ranges: [0, zero_blocks)
// fill -- 512 blocks per buffer, grid-stride --, then one thread per surfel, then one thread per permuted SH word.
__global__ void __launch_bounds__(256)
forward_prepare(const ForwardPrepare F)
{
    int b = (int)blockIdx.x;
    if (b < F.zero_blocks) {
        const int which = b / 512, bx = b % 512;
        float *p = F.zero.ptr[which];
        const unsigned long long n = F.zero.n[which];
        for (unsigned long long i = (unsigned long long)bx * 256 + threadIdx.x; i < n; i += 512ull * 256ull) p[i] = 0.f;
        return;
    }
    b -= F.zero_blocks;
    if (b < F.rec_blocks) {
        const int i = b * 256 + (int)threadIdx.x;
        if (i < F.P) surfel_record(i, F.mod, F.means, F.scales, F.rots, F.opac, F.srec);
        return;
    }
    b -= F.rec_blocks;
    permute_sh_word((size_t)b * 256 + threadIdx.x, F.P, F.nb, F.f16, F.shs, F.shp);
}

template <int CTRL> __device__ __forceinline__ float quad_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ __forceinline__ int quad_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

// SH colour of the 64 hits of a chunk, four lanes per surfel.  A lane-per-hit gather of the 192 B blocks is twelve 16 B loads whose 64 lanes
// touch 64 different cache lines EACH -- the L1's tag pipeline (one line per cycle), not bandwidth or arithmetic, is what the kernel waits
// for.  All 64 hits belong to ONE ray, so the SH basis is the same in every lane: lane q of a quad takes coefficients 4q .. 4q+3 of all four
// surfels of its quad (same number of loads and FMAs per lane, but each load instruction now covers 16 lines instead of 64), and the partial
// sums are transpose-reduced inside the quad with DPP.  bk[m] = basis[4q + m].
template <bool F16>
__device__ __forceinline__ void quad_sh_color(const void *shp, const int sid, const bool use, const float (&bk)[4], const int lane, float *col, bool *cl)
{
    const int q = lane & 3;
    float acc[4][3];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int sid_s = s == 0 ? quad_i<0x00>(sid) : s == 1 ? quad_i<0x55>(sid) : s == 2 ? quad_i<0xAA>(sid) : quad_i<0xFF>(sid);
        const int us = use ? 1 : 0;
        const int use_s = s == 0 ? quad_i<0x00>(us) : s == 1 ? quad_i<0x55>(us) : s == 2 ? quad_i<0xAA>(us) : quad_i<0xFF>(us);
        acc[s][0] = acc[s][1] = acc[s][2] = 0.f;
        if (use_s) {
            float x[12];
            if constexpr (F16) {
                const uint2 *p2 = reinterpret_cast<const uint2 *>(reinterpret_cast<const __half *>(shp) + (size_t)sid_s * 48) + q;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const uint2 h = p2[4 * i];
                    x[4 * i] = __half2float(__ushort_as_half((unsigned short)(h.x & 0xFFFFu))); x[4 * i + 1] = __half2float(__ushort_as_half((unsigned short)(h.x >> 16)));
                    x[4 * i + 2] = __half2float(__ushort_as_half((unsigned short)(h.y & 0xFFFFu))); x[4 * i + 3] = __half2float(__ushort_as_half((unsigned short)(h.y >> 16)));
                }
            } else {
                const float4 *p4 = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(shp) + (size_t)sid_s * 48) + q;
#pragma unroll
                for (int i = 0; i < 3; i++) { const float4 v = p4[4 * i]; x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w; }
            }
#pragma unroll
            for (int f = 0; f < 12; f++) acc[s][f % 3] += bk[f / 3] * x[f];
        }
    }
    // transpose-reduce inside the quad: lane q ends up with the sums of surfel q
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    float t[2][3], r[3];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float keep = b0 ? acc[2 * h + 1][c] : acc[2 * h][c], send = b0 ? acc[2 * h][c] : acc[2 * h + 1][c];
            t[h][c] = keep + quad_f<0xB1>(send);
        }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float keep = b1 ? t[1][c] : t[0][c], send = b1 ? t[0][c] : t[1][c];
        r[c] = keep + quad_f<0x4E>(send) + 0.5f;
        cl[c] = r[c] < 0.f;
        col[c] = cl[c] ? 0.f : r[c];
    }
}

// Sort AND composite, one wavefront per ray, one LANE per hit.  A lane-per-ray walk is a chain of dependent
// gathers -- list entry -> surfel record + SH block -> blend -> next entry -- whose length is the ray's hit count; here the 64 hits of
// a chunk fetch their records independently (all gathers in flight at once) and the front-to-back recurrences (transmittance product,
// the two distortion moments, the ten blended sums) become wavefront scans.  The (t, id) keys are sorted IN REGISTERS -- E keys per
// lane, a bitonic network whose cross-lane layers use DPP / ds_swizzle / permlane32 and whose long strides are register-to-register --
// so the sorted chunk c is simply register c: no LDS, no barriers, no bank conflicts.  The sorted list is written back only up to the
// terminating hit.
template <int E, bool QSH>
__device__ __forceinline__ void sort_composite_ray(const TraceArgs &A, const int slot, const int r, const int n, const int lane, unsigned &st_hits)
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
    if constexpr (E == 3) bitonic_sort_192(kreg, lane);
    else bitonic_sort<E, E * 64>(kreg, lane);
    const float ox = A.ray_o[3 * r], oy = A.ray_o[3 * r + 1], oz = A.ray_o[3 * r + 2];
    const float dx = A.ray_d[3 * r], dy = A.ray_d[3 * r + 1], dz = A.ray_d[3 * r + 2];
    float basis[16];
    {
        const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
#pragma unroll
        for (int k = 0; k < 16; k++) basis[k] = 0.f;
        sh_basis(A.D, dx * il, dy * il, dz * il, basis);
    }
    float bk[4];                                        // this lane's four basis values of the quad-cooperative colour evaluation
    {
        const int q = lane & 3;
#pragma unroll
        for (int m = 0; m < 4; m++) bk[m] = q == 0 ? basis[m] : q == 1 ? basis[4 + m] : q == 2 ? basis[8 + m] : basis[12 + m];
    }
    // carried across chunks (wave-uniform): transmittance, the two distortion moments, and the ten blended sums
    // [rgb 3, depth, acc, normal 3, aux 2] -- kept as running PREFIX sums because the backward needs them per hit
    float T = 1.0f, M1 = 0.f, M2 = 0.f, C[10];
#pragma unroll
    for (int j = 0; j < 10; j++) C[j] = 0.f;
    float dist = 0.f;                                   // per-lane partial sum
    int used = 0;
    float4 *state = A.state ? A.state + state_row0(A, slot, r) : nullptr;      // per-hit state: plane 0 = 16 B rows, plane 1 = 16 B rows (24 B with `others`)
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
#ifdef ENVGS_SORT_RECOMPUTE_T
            const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], s3, ox, oy, oz, dx, dy, dz);
#else
            const SurfHit h = hit_surfel_at(sr[0], sr[1], sr[2], s3, __uint_as_float((unsigned)(kreg[ce] >> 32)), ox, oy, oz, dx, dy, dz);      // t = the key's own high word
#endif
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
        // (QSH is a KERNEL template parameter: with both colour paths in one kernel the generic one keeps the 16 basis values alive through
        //  the whole ray -- 124 VGPRs = 4 waves per SIMD instead of 86 = 5, and this kernel spends 60 % of its time waiting for gathers)
        if constexpr (QSH) { if (A.f16) quad_sh_color<true>(A.shp, sid, use, bk, lane, col, cl); else quad_sh_color<false>(A.shp, sid, use, bk, lane, col, cl); }
        else if (use) surfel_color(A, sid, basis, col, cl);
        const float tt = t > NEAR_N ? t : NEAR_N;
        const float m = FAR_N / (FAR_N - NEAR_N) * (1.0f - NEAR_N / tt);
        const float mw = m * w, mmw = m * m * w;
        const float S1 = wave_scan_add(mw), S2 = wave_scan_add(mmw);
        const float M1b = M1 + (S1 - mw), M2b = M2 + (S2 - mmw);        // moments before this hit
        dist += (m * m * (1.0f - Tb) + M2b - 2.0f * m * M1b) * w;
        float x0 = 0.f, x1 = 0.f;
        if (A.has_others && use && !(ENVGS_SCF_KO & 2)) { const float2 xo = reinterpret_cast<const float2 *>(A.others)[sid]; x0 = xo.x; x1 = xo.y; }      // one 8 B gather
        float S[10] = {w * col[0], w * col[1], w * col[2], w * t, w, sg * w * s3.x, sg * w * s3.y, sg * w * s3.z, w * x0, w * x1};
#pragma unroll
        for (int j = 0; j < 8; j++) S[j] = C[j] + wave_scan_add(S[j]);                // inclusive: this hit already added
        if (A.has_others && !(ENVGS_SCF_KO & 2)) { S[8] = C[8] + wave_scan_add(S[8]); S[9] = C[9] + wave_scan_add(S[9]); }      // (the two aux sums: only when there is something to sum)
        if (use) {
            list[i] = make_uint2(__float_as_uint(w), (unsigned)sid);
            if (state) {
                // (the acc sum S[4] is not stored: sum_{j<=k} w_j = 1 - T_before * (1 - alpha), which the backward rebuilds)
                float4 *o = state + (size_t)i;
                // streamed once, read once by the backward much later: non-temporal, so it does not evict the surfel records / SH blocks
                typedef float nt4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((nt4){Tb, S[0], S[1], S[2]}, reinterpret_cast<nt4 *>(o));
                if (!A.colour_state && !(ENVGS_SCF_KO & 1)) {
                    if (A.has_others) {     // plane 1 with `others`: 24 B rows (depth, normal, the two aux sums) as two 12 B halves (4 B alignment is all a dwordx3 needs)
                        typedef float nt3 __attribute__((ext_vector_type(3), aligned(4)));      // (sizeof is 16: the halves are addressed in floats)
                        float *q = reinterpret_cast<float *>(state_row1(A, (size_t)(o - A.state)));
                        __builtin_nontemporal_store((nt3){S[3], S[5], S[6]}, reinterpret_cast<nt3 *>(q));
                        __builtin_nontemporal_store((nt3){S[7], S[8], S[9]}, reinterpret_cast<nt3 *>(q + 3));
                    } else __builtin_nontemporal_store((nt4){S[3], S[5], S[6], S[7]}, reinterpret_cast<nt4 *>(o + A.state_plane));
                }
            }
        }
        const int nu = f < 64 ? f : min(64, n - cb);                    // hits of this chunk that were blended
        used += nu;
        M1 += wave_bcast(S1, 63); M2 += wave_bcast(S2, 63);
#pragma unroll
        for (int j = 0; j < 8; j++) C[j] = wave_bcast(S[j], 63);
        if (A.has_others) { C[8] = wave_bcast(S[8], 63); C[9] = wave_bcast(S[9], 63); }
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
template <int EMAX, bool LONG, bool QSH>
__global__ void __launch_bounds__(256)         // (QSH main pass: 86 VGPRs = 5 waves per SIMD.  Forcing 6 -- 80 VGPRs, 5 dwords spilled -- measured 2.27 -> 2.47 ms)
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
            if (n > A.cap) continue;                            // overflow: the K-buffer kernel owns this ray
            if (n > 256) {                                      // long: the LONG pass sorts it -- queued here, so that pass is one ray per wavefront step
                if (A.long_list && lane == 0) A.long_list[A.batch0 * 64 + atomicAdd(A.counter + 24 + A.seg, 1u)] = (unsigned)slot;
                continue;
            }
            if (n <= 64) sort_composite_ray<1, QSH>(A, slot, r, n, lane, st_hits);
            else if (n <= 128) sort_composite_ray<2, QSH>(A, slot, r, n, lane, st_hits);
            else if (n <= 192) sort_composite_ray<3, QSH>(A, slot, r, n, lane, st_hits);       // (a third of the rays of the bench scene)
            else sort_composite_ray<4, QSH>(A, slot, r, n, lane, st_hits);
        }
    } else {
        // the longest list so far (this segment's collection has finished, so its own maximum is in): nothing to do in the usual case
        if ((int)__hip_atomic_load(A.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 256) return;
        if (A.long_list) {
            // the main pass queued the long rays of this segment: one ray per wavefront step, spread over the whole grid (neighbouring rays have
            // similar hit counts, so a wavefront that scanned its own 64-ray window used to sort a dozen long rays one after the other)
            const int nl = (int)__hip_atomic_load(A.counter + 24 + A.seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < nl; i += gridDim.x * 4) {
                const int slot = (int)A.long_list[A.batch0 * 64 + i];
                const int rr = ray_of(A, slot), nn = A.hit_cnt[rr];
                if (EMAX == 8 || nn <= 512) sort_composite_ray<8, QSH>(A, slot, rr, nn, lane, st_hits);
                else if constexpr (EMAX >= 16) sort_composite_ray<16, QSH>(A, slot, rr, nn, lane, st_hits);
            }
            if (A.stats && lane == 0 && st_hits) atomicAdd(A.stats + 0, (unsigned long long)st_hits);
            return;
        }
        for (int base = A.batch0 * 64 + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; base < slot_end; base += gridDim.x * 256) {
            const int slot = base + lane;
            int r = 0, n = 0;
            if (slot < slot_end) { r = ray_of(A, slot); n = A.hit_cnt[r]; }
            unsigned long long todo = __ballot(n > 256 && n <= A.cap);
            while (todo) {
                const int l = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                const int rr = __shfl(r, l), nn = __shfl(n, l);
                if (EMAX == 8 || nn <= 512) sort_composite_ray<8, QSH>(A, base + l, rr, nn, lane, st_hits);
                else if constexpr (EMAX >= 16) sort_composite_ray<16, QSH>(A, base + l, rr, nn, lane, st_hits);
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
template <bool CACHED>                                   // CACHED: lists of at most 256 hits with pairs wanted (see `cached` below); the launch decides
#ifndef ENVGS_RH_WPE
#define ENVGS_RH_WPE 6
#endif
#if ENVGS_RH_WPE > 0
__global__ void __launch_bounds__(64 * RH_W) __attribute__((amdgpu_waves_per_eu(ENVGS_RH_WPE, ENVGS_RH_WPE)))
#else
__global__ void __launch_bounds__(64 * RH_W)
#endif
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
    for (int base = (A.batch0 + (int)blockIdx.x) * 64; base < min(A.R, A.batch1 * 64); base += gridDim.x * 64) {
        const int batch = base >> 6;
        const int copy = batch & (NCOPY - 1);
        size_t rstart, region;
        batch_region(A, batch, rstart, region);
        unsigned long long *ent = A.entries ? A.entries + rstart : nullptr;
        unsigned *prs = A.pairs ? A.pairs + rstart : nullptr;
        __syncthreads();
        for (int i = threadIdx.x; i < RH_TAB; i += 64 * RH_W) { key[i] = -1; acc[i] = 0ull; }
        if (threadIdx.x == 0) { nfail = 0u; ndense = 0u; }
        __syncthreads();
        const int r = ray_of(A, base + lane);
        int n = 0;
        uint2 *list = A.hits;
        if (r < A.R && A.hit_cnt[r] <= A.cap) { n = A.n_used[r]; list = A.hits + (size_t)r * A.cap; }
#ifndef ENVGS_RH_U
#define ENVGS_RH_U 4
#endif
#ifndef ENVGS_RH_KO
#define ENVGS_RH_KO 0
#endif
        constexpr int U = ENVGS_RH_U;
        // one hit into the table: weight and count of its surfel (RANK: the count before it = the hit's place among its surfel's pairs)
        auto insert = [&]<bool RANK>(const uint2 e, const int k) __attribute__((always_inline)) -> unsigned {
            const unsigned long long wq = (unsigned long long)ceilf(__uint_as_float(e.x) * wscale);
            unsigned h = (e.y * 2654435761u) >> 22;
            bool ok = false;
            for (int t = 0; t < 24; t++) {
                const int old = atomicCAS(&key[h], -1, (int)e.y);
                if (old == -1) hod[atomicAdd(&ndense, 1u)] = (unsigned short)h;      // first to see this surfel: entries keep this order
                if (old == -1 || old == (int)e.y) { ok = true; break; }
                h = (h + 1) & (RH_TAB - 1);
            }
            if (ok) {
                if constexpr (RANK) return h | ((unsigned)(atomicAdd(&acc[h], (wq << 8) | 1ull) & 0xFFull) << 10);
                atomicAdd(&acc[h], (wq << 8) | 1ull);                     // (no return value needed: the pairs get their ranks in the last phase)
                return 0u;
            }
            // table full around h: an entry of its own
            const unsigned long long old = atomicAdd(A.surf_acc + (size_t)e.y * NCOPY + copy, (wq << 24) | 1ull);
            const unsigned f = atomicAdd(&nfail, 1u);
            if (ent) ent[region - 1 - f] = (unsigned long long)e.y | ((old & 0xFFFFFFull) << 32);
            if (prs) prs[region - 1 - f] = ((unsigned)lane << 16) | (unsigned)k;
            return 0xFFFFFFFFu;
        };
        // Lists of at most 256 hits (32 per lane) keep each hit's table slot and rank in registers -- 16 bits per hit -- so the last phase
        // neither reads the list again nor probes nor takes an atomic (round 6: that phase was 0.19 of the kernel's 0.61 ms).
#ifndef ENVGS_RH_UC
#define ENVGS_RH_UC 2
#endif
        constexpr int UC = ENVGS_RH_UC;                                   // hits per lane and step (2: 78 registers = 6 waves per SIMD and 0.47 ms; 4: 94 = 5 waves, 0.57 ms)
        constexpr int CIT = 256 / (UC * RH_W);
        constexpr bool cached = CACHED;
        unsigned hr[CIT * UC / 2] = {};
        unsigned fmask = 0u;                                              // hits that found no room in the table
        // (each lane walks its own list row: the loads of one step are 64 different cache lines, so the next step's entries are requested
        //  before this step's chain of LDS atomics starts)
        if constexpr (cached) {
            // (a rolled loop: the newest codes enter at the bottom of hr[] and the older ones move up -- 15 register moves per step; the last
            //  phase takes them back newest first)
            uint2 nxt[UC];
#pragma unroll
            for (int j = 0; j < UC; j++) { const int k = part + j * RH_W; nxt[j] = (k < n) ? list[k] : make_uint2(0u, 0u); }
            for (int kb = part; kb < n; kb += UC * RH_W) {
                uint2 e[UC];
#pragma unroll
                for (int j = 0; j < UC; j++) e[j] = nxt[j];
#pragma unroll
                for (int j = 0; j < UC; j++) { const int k = kb + (UC + j) * RH_W; nxt[j] = (k < n) ? list[k] : make_uint2(0u, 0u); }
                unsigned c[UC / 2] = {}, fm = 0u;
#pragma unroll
                for (int j = 0; j < UC; j++) {
                    const int k = kb + j * RH_W;
                    if (k < n) {
                        unsigned code = insert.template operator()<true>(e[j], k);
                        if (code == 0xFFFFFFFFu) { fm |= 1u << j; code = 0u; }
                        c[j >> 1] |= code << ((j & 1) * 16);
                    }
                }
#pragma unroll
                for (int i = CIT * UC / 2 - 1; i >= UC / 2; i--) hr[i] = hr[i - UC / 2];
#pragma unroll
                for (int i = 0; i < UC / 2; i++) hr[i] = c[i];
                fmask = (fmask << UC) | fm;
            }
        } else {
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
                    insert.template operator()<false>(e[j], k);
                }
            }
        }
        __syncthreads();
        // Flush in order of FIRST APPEARANCE along the rays (~ front to back): the backward then meets each ray's hits in roughly ascending
        // list position, so the per-hit state it gathers is consumed cache line by cache line instead of at random.
        const unsigned D = ndense;
        // SPARSE entries (round 6; envgs_trace.h: sparse_hits): a surfel that at most sparse_max of the batch's 64 rays blended does not become an
        // entry -- batch_surfel_bwd pays one 64-lane pass per entry however few of its lanes are live, and a fifth of the entries of the benchmark
        // view carry 1.4 % of its hits.  Its hits are filed in a global list instead (one lane of sparse_hits_bwd and one gradient record PER HIT:
        // the surfel's record count grows by its hit count).  List space is claimed per 64-surfel chunk with one atomic; a chunk that finds no
        // room gives its claim back and files ordinary entries (successful claims stay compact: a later claim can only succeed once the
        // counter is back below the capacity, i.e. behind every successful one).
        const bool sparse_on = A.sparse != nullptr && A.sparse_max > 0 && ent != nullptr && prs != nullptr;
        if (part == 0) {
            unsigned carry_off = 0u, carry_d = 0u;
            for (unsigned c = 0; c < D; c += 64) {
                const unsigned d = c + lane;
                const bool occ = d < D;
                const int h = occ ? (int)hod[d] : 0;
                const int sid = occ ? key[h] : 0;
                const unsigned long long av = occ ? acc[h] : 0ull;
                const unsigned cn = (unsigned)(av & 0xFFull);
                bool sp = sparse_on && occ && cn <= (unsigned)A.sparse_max;
                unsigned sp_pos = 0u;
                if (sparse_on) {
                    const unsigned spn = sp ? cn : 0u;
                    const float sincl = wave_scan_add((float)spn);
                    const unsigned stot = (unsigned)wave_bcast(sincl, 63);
                    if (stot > 0u) {
                        unsigned sb = 0u;
                        if (lane == 0) {
                            sb = atomicAdd(A.counter + 64, stot);
                            if (sb > A.sparse_cap || stot > A.sparse_cap - sb) { atomicSub(A.counter + 64, stot); sb = 0xFFFFFFFFu; }
                        }
                        sb = (unsigned)__builtin_amdgcn_readfirstlane((int)sb);
                        if (sb == 0xFFFFFFFFu) sp = false;
                        else sp_pos = sb + (unsigned)sincl - spn;
                    }
                }
                const bool dense = occ && !sp;
                const unsigned dcn = dense ? cn : 0u;
                const float incl = wave_scan_add((float)dcn);                // exact: at most 64*cap < 2^24 hits per batch
                const unsigned offh = carry_off + (unsigned)incl - dcn;
                const float dincl = wave_scan_add(dense ? 1.0f : 0.0f);
                const unsigned dd = carry_d + (unsigned)dincl - 1u;         // this entry's place among the batch's (dense) entries: first-appearance order kept
                if (occ) {
                    const unsigned long long old = atomicAdd(A.surf_acc + (size_t)sid * NCOPY + copy, ((av >> 8) << 24) | (unsigned long long)(sp ? cn : 1u));
                    if (dense) {
                        if (ent) ent[dd] = (unsigned long long)(unsigned)sid | ((unsigned long long)(cn - 1u) << 24) | ((old & 0xFFFFFFull) << 32);
                        acc[h] = (unsigned long long)offh;
                    } else {
                        // bit 63: sparse; bits 0..31 list position of the surfel's first hit, 32..55 its first record slot, 56..62 the rank handed out next
                        acc[h] = (1ull << 63) | (unsigned long long)sp_pos | ((old & 0xFFFFFFull) << 32);
                    }
                }
                carry_off += (unsigned)wave_bcast(incl, 63);
                carry_d += (unsigned)wave_bcast(dincl, 63);
            }
            if (A.n_entries && lane == 0) { A.n_entries[2 * batch] = (int)carry_d; A.n_entries[2 * batch + 1] = (int)nfail; }
            if (ent && lane == 0) atomicAdd(A.counter + 65, carry_d + nfail);      // entries of the call (with counters[2..3], the composited hits: how coherent its batches are)
            if (lane == 0) ptotal = carry_off;
        }
        __syncthreads();
        if (prs && !(ENVGS_RH_KO & 1)) {
            // Every hit finds its surfel's table slot again (the same probe sequence; hits that found no room fail again and were filed in
            // the first phase) and takes the next free position of that surfel's run of pairs: acc[h] holds the run's offset in its low word
            // and hands out ranks from its high word.  (Writing slot and rank back into the list in the first phase instead cost a scattered
            // 4 B store -- a whole 32 B sector of write traffic -- and a second gather per hit: 2 GB per step.)
            if constexpr (cached) {
                for (int it = n > part ? (n - part - 1) / (UC * RH_W) : -1; it >= 0; it--) {
                    const int kb = part + it * UC * RH_W;
                    unsigned c[UC / 2];
#pragma unroll
                    for (int i = 0; i < UC / 2; i++) c[i] = hr[i];
#pragma unroll
                    for (int i = 0; i < CIT * UC / 2 - UC / 2; i++) hr[i] = hr[i + UC / 2];
                    const unsigned fm = fmask;
                    fmask >>= UC;
#pragma unroll
                    for (int j = 0; j < UC; j++) {
                        const int k = kb + j * RH_W;
                        if (k < n && !((fm >> j) & 1u)) {
                            const unsigned code = (c[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                            const unsigned h = code & (RH_TAB - 1), rank = code >> 10;
                            const unsigned long long o = acc[h];
                            if (sparse_on && (o >> 63))
                                A.sparse[(unsigned)o + rank] = make_uint4((unsigned)(base + lane), (unsigned)k, (unsigned)key[h], ((unsigned)(o >> 32) & 0xFFFFFFu) + rank);
                            else {
                                const unsigned idx = (unsigned)o + rank, v = ((unsigned)lane << 16) | (unsigned)k;
                                if (idx < (unsigned)RH_STAGE) pstage[idx] = v; else prs[idx] = v;
                            }
                        }
                    }
                }
            } else {
                constexpr int U2 = ENVGS_RH_U;                      // independent loads first: one memory round trip per 4 hits, not per hit
                for (int kb = part; kb < n; kb += U2 * RH_W) {
                    unsigned sidv[U2];
    #pragma unroll
                    for (int j = 0; j < U2; j++) { const int k = kb + j * RH_W; sidv[j] = (k < n) ? list[k].y : 0xFFFFFFFFu; }
    #pragma unroll
                    for (int j = 0; j < U2; j++)
                        if (sidv[j] != 0xFFFFFFFFu) {
                            unsigned h = (sidv[j] * 2654435761u) >> 22;
                            bool ok = false;
                            for (int t = 0; t < 24; t++) {
                                const int kk = key[h];
                                if (kk == (int)sidv[j]) { ok = true; break; }
                                if (kk == -1) break;
                                h = (h + 1) & (RH_TAB - 1);
                            }
                            if (ok) {
                                if (sparse_on && (acc[h] >> 63)) {               // (the flag never changes once the flush has set it)
                                    const unsigned long long o = atomicAdd(&acc[h], 1ull << 56);
                                    const unsigned rank = (unsigned)(o >> 56) & 0x7Fu;
                                    A.sparse[(unsigned)o + rank] = make_uint4((unsigned)(base + lane), (unsigned)(kb + j * RH_W), sidv[j], ((unsigned)(o >> 32) & 0xFFFFFFu) + rank);
                                } else {
                                    const unsigned long long o = atomicAdd(&acc[h], 1ull << 32);
                                    const unsigned idx = (unsigned)o + (unsigned)(o >> 32), v = ((unsigned)lane << 16) | (unsigned)(kb + j * RH_W);
                                    if (idx < (unsigned)RH_STAGE) pstage[idx] = v; else prs[idx] = v;
                                }
                            }
                        }
                }
            }
            __syncthreads();
            const unsigned T = min(ptotal, (unsigned)RH_STAGE);
            for (unsigned i = threadIdx.x; i < T; i += 64 * RH_W) prs[i] = pstage[i];
            // the hits that found no room in the table (an incoherent batch: more than ~1000 distinct surfels) are one-hit entries filed from the top of
            // the region: with sparse entries on they move to the sparse list as well -- one claim per batch -- and the batch kernel sees none
            if (sparse_on && nfail > 0u) {
                __shared__ unsigned fbase;
                if (threadIdx.x == 0) {
                    unsigned sb = atomicAdd(A.counter + 64, nfail);
                    if (sb > A.sparse_cap || nfail > A.sparse_cap - sb) { atomicSub(A.counter + 64, nfail); sb = 0xFFFFFFFFu; }
                    fbase = sb;
                }
                __syncthreads();                                   // (also: this workgroup's own ent / prs stores of the first phase are visible to it)
                if (fbase != 0xFFFFFFFFu) {
                    for (unsigned f = threadIdx.x; f < nfail; f += 64 * RH_W) {
                        const unsigned long long ev = ent[region - 1 - f];
                        const unsigned pv = prs[region - 1 - f];
                        A.sparse[fbase + f] = make_uint4((unsigned)base + (pv >> 16), pv & 0xFFFFu, (unsigned)(ev & 0xFFFFFFull), (unsigned)(ev >> 32));
                    }
                    if (threadIdx.x == 0) { if (A.n_entries) A.n_entries[2 * batch + 1] = 0; atomicSub(A.counter + 65, nfail); }
                }
            }
        }
    }
}

template __global__ void register_hits<false>(const TraceArgs A);
template __global__ void register_hits<true>(const TraceArgs A);

// Row offsets of the compact per-hit buffers (envgs_trace.h: compact_rows).  Runs per forward segment between the collection and the sort:
// rows of a ray = its hits found (none for a ray whose list overflowed), scanned over the segment's slots in coherence-sorted order.  The
// cooperative collection writes each batch's row count as it finishes the batch (TraceArgs::batch_cnt), so two small launches remain: one
// workgroup scanning the segment's batch counts, then one wavefront per batch placing its rays.  (row_count -- per-block sums from the hit
// counts -- serves the diagnostic collection kernels, which do not write batch counts.)
__device__ __forceinline__ unsigned rows_of_slot(const TraceArgs &A, const int slot, const int slot_end)
{
    if (slot >= slot_end) return 0u;
    const int r = ray_of(A, slot);
    const int n = A.hit_cnt[r];
    return n > A.cap ? 0u : (unsigned)n;
}

__global__ void __launch_bounds__(256)
row_count(const TraceArgs A, unsigned *__restrict__ blk)
{
    const int slot_end = min(A.R, A.batch1 * 64);
    const int slot = A.batch0 * 64 + (int)blockIdx.x * 256 + (int)threadIdx.x;
    if ((slot >> 6) >= A.batch1) return;
    const float s = wave_sum((float)rows_of_slot(A, slot, slot_end));            // exact: <= 64 * 1024
    if ((threadIdx.x & 63) == 0) blk[slot >> 6] = (unsigned)s;
}

__global__ void __launch_bounds__(256)
row_scan_blocks(unsigned *__restrict__ blk, int n, unsigned *rows_used, unsigned *seg_base)
{
    // (rows_used / seg_base: this segment CLAIMS its rows of the compact per-hit buffers from one counter shared by the call's segments -- round 3
    //  gave every segment the share of the rows that matched its share of the RAYS, which starves a segment whose rays happen to find most of
    //  the hits (the parked rays of a bounce stage sort into batches of their own: one segment may hold nearly all the live ones))
    // exclusive scan in place, ONE workgroup, 1024 counts per pass, four consecutive counts per thread: integer wave scans of the threads' sums (the
    // totals exceed 2^24: no float scan), the four wavefront totals through LDS, a running base carried from pass to pass.  Round 6: 256 threads
    // instead of 1024 -- the kernel sits on every forward segment's critical chain (collection -> scan -> row offsets -> sort pass) beside the
    // OTHER segment's kernels, and a 16-wavefront workgroup has to find sixteen free wavefront slots on ONE CU at once: 10 us of work took 103 us
    // on average (626 at worst) from first to last wavefront (profiles/r06_envgs_kernel_stats.csv); four wavefronts fit anywhere.
    __shared__ unsigned wtot[4];
    __shared__ unsigned s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0u;
    __syncthreads();
    constexpr int PRE = 8;                     // passes whose loads are issued up front (8192 batches = 524 k rays per segment; beyond that the loop loads as it goes)
    unsigned pre[PRE][4];
#pragma unroll
    for (int c = 0; c < PRE; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = c * 1024 + 4 * (int)threadIdx.x + q; pre[c][q] = i < n ? blk[i] : 0u; }
    int c = 0;
    for (int c0 = 0; c0 < n; c0 += 1024, c++) {
        const int i0 = c0 + 4 * (int)threadIdx.x;
        unsigned v[4];
        if (c < PRE) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v[q] = pre[0][q];
#pragma unroll
                for (int p_ = 1; p_ < PRE; p_++) v[q] = c == p_ ? pre[p_][q] : v[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = i0 + q < n ? blk[i0 + q] : 0u;
        }
        const unsigned tsum = v[0] + v[1] + v[2] + v[3];
        unsigned x = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = (unsigned)__shfl_up((int)x, o); if (lane >= o) x += y; }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        unsigned before = s_base;
        for (int w = 0; w < wave; w++) before += wtot[w];
        unsigned run = before + x - tsum;
#pragma unroll
        for (int q = 0; q < 4; q++) { if (i0 + q < n) blk[i0 + q] = run; run += v[q]; }
        __syncthreads();
        if (threadIdx.x == 255) s_base = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0 && rows_used && seg_base) *seg_base = atomicAdd(rows_used, s_base);
}

__global__ void __launch_bounds__(256)
row_offsets(const TraceArgs A, const unsigned *__restrict__ blk, unsigned *__restrict__ row_off, uint2 *__restrict__ batch_rows,
            const unsigned *__restrict__ seg_base, unsigned long long limit)
{
    const unsigned long long base = (unsigned long long)*seg_base;        // claimed by row_scan_blocks
    // one wavefront per batch; blk[batch] = rows of the segment's batches before this one (exclusive scan of the counts the collection wrote)
    const int slot_end = min(A.R, A.batch1 * 64);
    const int slot = A.batch0 * 64 + (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int lane = threadIdx.x & 63, batch = slot >> 6;
    if (batch >= A.batch1) return;
    const unsigned cnt = rows_of_slot(A, slot, slot_end);
    const float incl = wave_scan_add((float)cnt);
    const unsigned long long row0 = base + (unsigned long long)blk[batch];
    const unsigned long long row = row0 + (unsigned long long)((unsigned)incl - cnt);
    const bool fits = row + cnt <= limit;
    if (slot < slot_end) {
        row_off[slot] = (unsigned)(fits ? row : 0ull);
        if (!fits && cnt > 0u) {                                     // the segment's share of the rows is used up: this ray (and every later one
            A.hit_cnt[ray_of(A, slot)] = A.cap + 1;                 // of the segment) takes the K-buffer kernels, like a ray whose list overflowed
            atomicAdd(A.counter + 21, 1u);
        }
    }
    const unsigned tot = (unsigned)wave_bcast(incl, 63);
    if (lane == 0) {                                                // first row and rows, clipped to the segment's share (rays that do not fit are a suffix)
        const unsigned long long end = row0 + tot < limit ? row0 + tot : limit;
        batch_rows[batch] = make_uint2((unsigned)(row0 < limit ? row0 : limit), (unsigned)(end > row0 ? end - row0 : 0ull));
    }
}

// Split the packed per-surfel accumulators of composite_lists_fwd into hit counts (for the scan) and weights (added to `wet`,
// which the K-buffer path may already have contributed to in float).
__global__ void __launch_bounds__(256)
unpack_surfel_acc(int P, int wfrac, const unsigned long long *__restrict__ acc, unsigned *__restrict__ cnt, float *__restrict__ wet, unsigned *ray_counter)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && ray_counter) *ray_counter = 0u;          // the ray-fetch counter of the K-buffer overflow pass that follows (was a memset launch of its own)
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


template __global__ void sort_composite_fwd<4, false, false>(const TraceArgs A);
template __global__ void sort_composite_fwd<8, true, false>(const TraceArgs A);
template __global__ void sort_composite_fwd<16, true, false>(const TraceArgs A);
template __global__ void sort_composite_fwd<4, false, true>(const TraceArgs A);
template __global__ void sort_composite_fwd<8, true, true>(const TraceArgs A);
template __global__ void sort_composite_fwd<16, true, true>(const TraceArgs A);

}  // namespace envgs
