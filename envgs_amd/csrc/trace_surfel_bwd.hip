// trace_surfel_bwd.hip -- list path backward: the surfel-major batch kernel (lane = ray, MFMA reduction), the per-surfel record reduction, the
// parameter-gradient finish, and the per-ray atomic-flush fallback (composite_lists_bwd).
#include "trace_common.h"

namespace envgs {

#ifdef ENVGS_DIAG   // per-ray atomic-flush backward of the list path: superseded by the record backward, kept for A/B measurements and tests
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
#endif  // ENVGS_DIAG

// Backward of the list path, SURFEL-MAJOR per batch (the tracer's counterpart of the rasterizer's tile backward): one wavefront owns a
// batch of 64 coherence-sorted rays, LANE = RAY.  It walks the batch's entries (distinct surfels); the surfel's record and SH block are
// staged through LDS 16 entries ahead (coalesced, off the critical path) and read back as broadcasts, each ray that composited the surfel
// fetches the per-hit state the forward stored (transmittance before the hit, the ten prefix sums after it), evaluates its gradient
// independently of every other hit, and the 63 gradient words (48 SH + 15 geometry) are transpose-reduced over the wavefront into ONE
// 256 B record per (batch, surfel) written by the 64 lanes as one coalesced line pair.  ~27x fewer records than one per hit, no
// per-hit gathers of surfel data, no dependent chain along the ray, no atomics.
// every LDS read issued so far has returned, and the compiler may not sink a later use's read below this point (s_waitcnt lgkmcnt(0) + a
// scheduling barrier): used to make a set of MFMAs start with ALL its operands in registers instead of one round trip per operand
#define ENVGS_LDS_FENCE() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef ENVGS_BS_GROUP
#define ENVGS_BS_GROUP 15
#endif
constexpr int BS_GROUP = ENVGS_BS_GROUP;   // entries staged per group.  15 = three whole runs of five (round 6): with 16 every group ended in a run of ONE entry that paid the three
                                           // MFMA sets of a run on its own (12 MFMAs per entry instead of 9.6)
static_assert(BS_GROUP <= 16, "stage() assigns four lanes per entry");
#ifndef ENVGS_BSB_KO
#define ENVGS_BSB_KO 0          // measurement builds only (scratch/ab_bsb.sh): 1 = no dothers, 2 = no aux-plane fetch, 4 = no plane-1 fetch, 8 = no plane-0 fetch -- results wrong by construction
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BT_PITCH = 96;  // words per row of the reduction tile: 64 rays + the skew of 2 words per row (<= 30), never wrapped
constexpr int RECW = 64;      // floats per (batch, surfel) gradient record: 48 SH (or 3 colour) + 15 geometry + pad
// RGBO: the colour is the only output with an upstream gradient (g_dpt / g_acc / g_norm / g_aux all NULL) -- the EnvGS training step: the env
// pass's depth / accumulation / normal maps are not supervised.  Seven per-ray gradient constants and their terms drop out, and of the per-hit
// state only plane 0 (16 B: transmittance before the hit + the three colour prefix sums) is fetched instead of 32 / 48 B.
// OTH (generic form only): the call has `others_precomp` -- a kernel parameter rather than a run-time test of A.has_others because the two
// plane-1 row forms would otherwise meet in PHI copies right behind the branch, and the wait for them would undo the state prefetch (measured)
template <bool RGBO, bool OTH>
__global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64)
batch_surfel_bwd(const TraceArgs A)
{
    static_assert(!(RGBO && OTH), "the colour-only form never touches `others`");
    __shared__ float4 sdat[2][BS_GROUP][16];               // per entry: surfel record (4 x 16 B) + SH block (12 x 16 B)
    __shared__ unsigned long long sdesc[2][BS_GROUP];      // sid | (hits-1) << 24 ; record index << 32
    __shared__ unsigned short kmat[2][BS_GROUP][64];       // per entry and ray: list position of the hit + 1, 0 = the ray did not blend it
    __shared__ unsigned spb[2][BS_GROUP];                  // per entry: index of its first pair
    __shared__ unsigned scn[2][BS_GROUP];                  // per entry: hits
    __shared__ float btile[16][BT_PITCH];                  // B operand of the reduction MFMAs: 16 words per ray, skewed (see below)
    __shared__ float2 sox[2][BS_GROUP];                    // per entry: the surfel's two `others` values (generic form only)
    const int lane = threadIdx.x;
    const int nb = (A.D + 1) * (A.D + 1);
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
            if (!(B.il > 0.0f && B.il < 3.0e38f)) {
                // a ray without a direction (zero, NaN or infinite d) composited nothing, but its basis is a row of the MFMA operand that sums the
                // batch's colour gradients: 0 * NaN there would poison dL/dshs of every surfel the OTHER 63 rays blended
#pragma unroll
                for (int k = 0; k < 16; k++) basis[k] = 0.f;
            }
            Box = B.ox; Boy = B.oy; Boz = B.oz; Bdx = B.dx; Bdy = B.dy; Bdz = B.dz;
            gR0 = B.gR0; gR1 = B.gR1; gR2 = B.gR2;
            if constexpr (RGBO) { gD = gA = gN0 = gN1 = gN2 = gX0 = gX1 = 0.f; }
            else { gD = B.gD; gA = B.gA; gN0 = B.gN0; gN1 = B.gN1; gN2 = B.gN2; gX0 = B.gX0; gX1 = B.gX1; }
            Fsum = B.gR0 * B.fr0 + B.gR1 * B.fr1 + B.gR2 * B.fr2 + B.gD * B.fD + B.gA * B.fA + B.gN0 * B.fN0 + B.gN1 * B.fN1 + B.gN2 * B.fN2 +
                   B.gX0 * B.fX0 + B.gX1 * B.fX1 + B.fT * B.bgdot;
        }
        // A operand of the reduction MFMAs, constant for the batch: lane l holds basis_{l & 15} of ray 4s + (l >> 4)
        float Areg[16];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) btile[k][lane + 2 * k] = (valid && k < nb) ? basis[k] : (k == 0 ? kC0 : 0.f);
        __syncthreads();
#pragma unroll
        for (int sI = 0; sI < 16; sI++) Areg[sI] = btile[lane & 15][4 * sI + (lane >> 4) + 2 * (lane & 15)];
        // ... and of the COLOUR MFMAs (the other orientation: rows = rays): Acol[4 b + s] of lane l = basis_{4 s + (l >> 4)} of ray 16 b + (l & 15).
        // The rays' own basis[] registers are dead from here on (16 VGPRs either way).
        float Acol[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int k = 4 * (q & 3) + (lane >> 4);
            Acol[q] = btile[k][16 * (q >> 2) + (lane & 15) + 2 * k];
        }
        __syncthreads();
        // dL/d(SH basis) of the rays, accumulated on the matrix cores in the MFMA output layout: SkM[b][i] of lane l = (ray 16 b + 4 (l >> 4) + i,
        // basis l & 15); handed to the rays through LDS at the end of the batch
        f32x4 SkM[4];
#pragma unroll
        for (int k = 0; k < 4; k++) SkM[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        float dO0 = 0.f, dO1 = 0.f, dO2 = 0.f, dD0 = 0.f, dD1 = 0.f, dD2 = 0.f;
        const float4 *state = A.state + state_row0(A, min(base + lane, A.R - 1), rr);
        const char *plane1 = reinterpret_cast<const char *>(A.state + A.state_plane);
        size_t rstart, region;
        batch_region(A, batch, rstart, region);
        const unsigned long long *ent = A.entries + rstart;
        const unsigned *prs = A.pairs + rstart;
        const int D = A.n_entries[2 * batch], NE = D + A.n_entries[2 * batch + 1];
        unsigned poff = 0u;                                  // pairs of the table entries staged so far
        // Stage one group of entries, a whole group ahead of its use: 4 lanes per entry fetch the surfel record and SH block, then the
        // group's (lane, k) pairs (one contiguous run) are scattered into kmat -- so the main loop touches no global memory except
        // each ray's per-hit state.
        auto stage = [&](int g, int buf) {
            const int el = lane >> 2, part = lane & 3;
            const int e = g * BS_GROUP + el;
            const bool elv = el < BS_GROUP;
            unsigned long long d = 0ull;
            if (elv && e < NE) {
                d = e < D ? ent[e] : ent[region - 1 - (size_t)(e - D)];
                const int sid = (int)(d & 0xFFFFFFull);
                sdat[buf][el][part] = A.srec[(size_t)sid * 4 + part];
                if (A.M == 16 && A.f16) {
                    // fp16 storage: this lane's 12 coefficients are 24 B = three 8 B loads, each converted into one staged float4
                    const uint2 *s2 = reinterpret_cast<const uint2 *>(reinterpret_cast<const __half *>(A.shs) + (size_t)sid * 48);
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const uint2 h = s2[part * 3 + q];
                        float4 x = make_float4(__half2float(__ushort_as_half((unsigned short)(h.x & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(h.x >> 16))),
                                               __half2float(__ushort_as_half((unsigned short)(h.y & 0xFFFFu))), __half2float(__ushort_as_half((unsigned short)(h.y >> 16))));
                        const int i0 = (part * 3 + q) * 4;
                        if (i0 + 0 >= nb * 3) x.x = 0.f;
                        if (i0 + 1 >= nb * 3) x.y = 0.f;
                        if (i0 + 2 >= nb * 3) x.z = 0.f;
                        if (i0 + 3 >= nb * 3) x.w = 0.f;
                        sdat[buf][el][4 + part * 3 + q] = x;
                    }
                } else if (A.M == 16) {
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
                    const Feat fsh = Feat{A.shs, A.f16 != 0}.at(A.M > 0 ? (size_t)sid * A.M * 3 : 0), fcol = Feat{A.colors, A.f16 != 0}.at(A.M > 0 ? 0 : (size_t)sid * 3);
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int idx = (part * 3 + q) * 4 + c;
                            v[c] = A.M > 0 ? (idx < nb * 3 ? fsh[idx] : 0.f) : (idx < 3 ? fcol[idx] : 0.f);
                        }
                        sdat[buf][el][4 + part * 3 + q] = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
                if (part == 0) {
                    const size_t ci = (size_t)sid * NCOPY + copy;
                    const unsigned rec = A.surf_off[ci] - A.surf_cnt[ci] + (unsigned)(d >> 32);
                    sdesc[buf][el] = (d & 0x3FFFFFFFull) | ((unsigned long long)rec << 32);
                    if constexpr (OTH) sox[buf][el] = reinterpret_cast<const float2 *>(A.others)[sid];
                }
            }
            if (part == 0 && elv) scn[buf][el] = e < NE ? (unsigned)((d >> 24) & 63ull) + 1u : 0u;
            if (lane * 32 < BS_GROUP * 128) {   // clear kmat[buf]: 128 B per entry, 32 B per lane
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
            float4 st0, st1 = make_float4(0.f, 0.f, 0.f, 0.f), st2 = make_float4(0.f, 0.f, 0.f, 0.f);
            // per-hit state of one list position: plane 0 (16 B) -- and, generic form, plane 1: 16 B rows, or 24 B rows = two 12 B halves with `others`
            auto fetch_state = [&](const int k) {
                const float4 *sp = k > 0 ? state + (size_t)(k - 1) : A.state;      // unconditional (idle lanes share one address): no branch, no wait
                if (ENVGS_BSB_KO & 8) st0 = make_float4(0.5f, 0.1f, 0.1f, 0.1f); else st0 = sp[0];
                if constexpr (!RGBO) {
                    if (ENVGS_BSB_KO & 4) return;
                    if constexpr (OTH) {
                        typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));       // (sizeof is 16: the halves are addressed in floats)
                        const float *q = reinterpret_cast<const float *>(plane1 + (size_t)(sp - A.state) * 24);
                        const f3 a = *reinterpret_cast<const f3 *>(q); st1.x = a.x; st1.y = a.y; st1.z = a.z;
                        if (!(ENVGS_BSB_KO & 2)) { const f3 b = *reinterpret_cast<const f3 *>(q + 3); st1.w = b.x; st2.x = b.y; st2.y = b.z; }
                    } else st1 = *reinterpret_cast<const float4 *>(plane1 + (size_t)(sp - A.state) * 16);
                }
            };
            fetch_state(k1);
            // (round 6) the entry's place in its run of five is a COMPILE-TIME constant: the loop body is instantiated five times, so the tile rows of
            // BT(), the `first / last of a run` tests and the run-relative LDS addresses are immediates instead of ~25 scalar / address instructions per entry
            auto entry = [&]<int e5>(const int el) __attribute__((always_inline)) {
                const unsigned long long d = sdesc[buf][el];
                const int sid = (int)(d & 0xFFFFFFull);
                const unsigned long long rec = d >> 32;
                const bool act = k1 > 0;
                // This ray's gradient words.  dL/dcolour (3) goes to the LDS tile of the reduction MFMAs -- columns 3 e5 .. 3 e5 + 2, e5 = the
                // entry's place in a run of FIVE: all entries of a batch share the A operand (the rays' basis values), so one set of sixteen
                // MFMAs reduces the colour columns of five entries (the matrix pipe's 16 x 32 cycles per set are not hidden by the other
                // wavefront: 0.66 of 3.64 ms when every entry had its own set).  The 15 geometry words only ever needed the plain sum over the
                // rays and take the rasterizer's wavefront transpose-reduce instead.  Tile layout: word n of ray j at n*BT_PITCH + j + 2n (round 5: rows of 96 words, the skew no longer wraps -- every operand address is one base + an immediate):
                // conflict-free both for the writes (fixed n, 64 rays) and for the MFMA operand reads (16 words x 4 rays).  One wavefront
                // per workgroup: its LDS operations execute in program order, so no barrier is needed -- and a barrier's vmcnt(0) would
                // drain the state prefetch that is in flight.
                float gw[16];
#pragma unroll
                for (int n = 0; n < 16; n++) gw[n] = 0.f;
                [[maybe_unused]] float gx1 = 0.f;
#define BT(n) btile[3 * e5 + (n)][lane + 2 * (3 * e5 + (n))]
                // The SH colours of the run's five entries for all 64 rays, on the matrix cores (round 5; every lane used to evaluate its own:
                // 48 FMAs and twelve 16 B LDS reads per entry and lane):  [64 rays x 16 basis values] . [16 x 15]  (column 3 e + c = colour c of
                // the run's e-th surfel, straight from the staged SH blocks) = 16 MFMAs per run, written into the tile's columns 3 e + c --
                // exactly the slots that receive the entry's colour GRADIENT a few lines below, once the lane has read its colour from them.
                if (A.M > 0 && e5 == 0) {
                    const int nrun = min(5, ne - el);
                    const int kk = lane >> 4, nn = lane & 15, ee = nn / 3, c2 = nn - 3 * ee;
                    f32x4 cc0 = {0.f, 0.f, 0.f, 0.f}, cc1 = cc0, cc2 = cc0, cc3 = cc0;
                    // (round 6: every B operand of a set is read from LDS BEFORE the set's first MFMA -- the compiler had put one ds_read + s_waitcnt
                    //  lgkmcnt(0) in front of every two to four MFMAs: twenty serialised LDS round trips per run of five entries at 2 wavefronts per SIMD)
                    float bvc[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) {
                        bvc[ks] = 0.f;
                        if (nn < 15 && ee < nrun) bvc[ks] = reinterpret_cast<const float *>(&sdat[buf][el + ee][4])[(4 * ks + kk) * 3 + c2];
                    }
                    ENVGS_LDS_FENCE();
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) {
                        cc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Acol[ks], bvc[ks], cc0, 0, 0, 0);
                        cc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Acol[4 + ks], bvc[ks], cc1, 0, 0, 0);
                        cc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Acol[8 + ks], bvc[ks], cc2, 0, 0, 0);
                        cc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(Acol[12 + ks], bvc[ks], cc3, 0, 0, 0);
                    }
                    if (nn < 15) {
                        float *trow = &btile[nn][0];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int j = 4 * kk + i + 2 * nn;                 // (output layout: element i of lane l = row 4 (l >> 4) + i, column l & 15)
                            trow[j] = cc0[i]; trow[j + 16] = cc1[i]; trow[j + 32] = cc2[i]; trow[j + 48] = cc3[i];
                        }
                    }
                }
                if (act) {
                    const float4 s0 = sdat[buf][el][0], s1 = sdat[buf][el][1], s2 = sdat[buf][el][2], s3 = sdat[buf][el][3];
                    const SurfHit h = hit_surfel(s0, s1, s2, s3, Box, Boy, Boz, Bdx, Bdy, Bdz);
                    float col[3]; bool cl[3] = {false, false, false};
                    if (A.M > 0) {
                        const float rc[3] = {BT(0), BT(1), BT(2)};
#pragma unroll
                        for (int c = 0; c < 3; c++) { const float v = rc[c] + 0.5f; cl[c] = v < 0.f; col[c] = cl[c] ? 0.f : v; }
                    } else { const float4 x = sdat[buf][el][4]; col[0] = x.x; col[1] = x.y; col[2] = x.z; }
                    float x0 = 0.f, x1 = 0.f;                 // (staged with the record: a global load here sat on every entry's critical path)
                    if constexpr (OTH) { const float2 xo = sox[buf][el]; x0 = xo.x; x1 = xo.y; }
                    const float alpha = h.alpha, Tb = st0.x;
                    const float w = alpha * Tb;
                    const float sgn = h.denom < 0.0f ? 1.0f : -1.0f;
                    const float nf0 = sgn * s3.x, nf1 = sgn * s3.y, nf2 = sgn * s3.z;
                    const float inv1m = __builtin_amdgcn_rcpf(1.0f - alpha);          // v_rcp_f32 (1 ulp): gradient-only terms need no IEEE division
                    float gv_ = gR0 * col[0] + gR1 * col[1] + gR2 * col[2], gS = gR0 * st0.y + gR1 * st0.z + gR2 * st0.w;
                    if constexpr (!RGBO) {
                        gv_ += gD * h.t + gA + gN0 * nf0 + gN1 * nf1 + gN2 * nf2 + gX0 * x0 + gX1 * x1;
                        gS += gD * st1.x + gA * (1.0f - Tb * (1.0f - alpha)) + gN0 * st1.y + gN1 * st1.z + gN2 * st1.w + gX0 * st2.x + gX1 * st2.y;
                    }
                    const float dLa = Tb * gv_ - (Fsum - gS) * inv1m;
                    const float dc[3] = {cl[0] ? 0.f : w * gR0, cl[1] ? 0.f : w * gR1, cl[2] ? 0.f : w * gR2};
                    // dL/d(others) = sum over the batch's rays of w g_aux: words 15 / 16 of the (then five-register) transpose-reduce below and ONE two-lane
                    // atomic per entry (round 6; until then every lane added its own term -- 64 same-address atomics per entry and word: 2.1 of 5.1 ms)
                    if constexpr (OTH) { gw[15] = w * gX0; gx1 = w * gX1; }
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
                    gw[0] = -e0; gw[1] = -e1; gw[2] = -e2;
                    gw[3] = cu * qx; gw[4] = cu * qy; gw[5] = cu * qz;
                    gw[6] = cv * qx; gw[7] = cv * qy; gw[8] = cv * qz;
                    const float ws = w * sgn;
                    gw[9] = ws * gN0 - kt * qx; gw[10] = ws * gN1 - kt * qy; gw[11] = ws * gN2 - kt * qz;
                    gw[12] = -cu * h.u * A.mod;
                    gw[13] = -cv * h.v * A.mod;
                    gw[14] = h.G * dLa;
                    dO0 += e0; dO1 += e1; dO2 += e2;
                    dD0 += h.t * e0; dD1 += h.t * e1; dD2 += h.t * e2;
                } else {
                    BT(0) = 0.f; BT(1) = 0.f; BT(2) = 0.f;
                }
#undef BT
                if (el + 1 < ne) {                           // next entry's state: in flight during the reduction below
                    k1 = valid ? (int)kmat[buf][el + 1][lane] : 0;
                    fetch_state(k1);
                }
                // geometry: lane r*16 + k (k < 4) receives the sum over the 64 rays of word k + 4 r; record words 48 .. 62
                {
                    if constexpr (OTH && !(ENVGS_BSB_KO & 1)) {
                        // with dL/d(others): 17 words through the five-register form -- lane r*16 + k (k < 5) receives word k + 5 r; words 15 / 16 (the sums
                        // of w g_aux, lanes 48 / 49) leave as ONE two-lane atomic per entry
                        float g20[20];
#pragma unroll
                        for (int n = 0; n < 20; n++) g20[n] = n < 15 ? gw[n] : 0.f;
                        g20[15] = gw[15]; g20[16] = gx1;
                        const float gsum = wave_transpose_reduce<5>(g20, lane);
                        const int word = (lane & 15) + 5 * (lane >> 4);
                        if ((lane & 15) < 5 && word < 15 && rec < A.num_records) A.records[rec * RECW + 48 + word] = gsum;
                        if ((lane & 15) < 5 && (word == 15 || word == 16) && A.dothers) atomic_add_f32(A.dothers + 2 * sid + (word - 15), gsum);
                    } else {
                        const float gsum = wave_transpose_reduce<4>(gw, lane);
                        const int word = (lane & 15) + 4 * (lane >> 4);
                        if ((lane & 15) < 4 && word < 15 && rec < A.num_records) A.records[rec * RECW + 48 + word] = gsum;
                    }
                }
                // colour: after the fifth entry of a run (or the last of the group) D[16 x 16] = basis^T[16 x 64 rays] . B[64 rays x 16], K = 64 in
                // 16 exact-f32 MFMAs (four independent chains: the dependent latency is 40 cycles); column 3 e + c = colour c of the run's
                // e-th entry, the 16 rows its (16,3) SH gradient block (row 0 = C0 x the plain sum when there are no SH).
                if (e5 == 4 || el + 1 == ne) {
                    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f}, accB = acc4, accC = acc4, accD = acc4;
                    {
                        const int n = lane & 15, j = lane >> 4;
                        const float *brow = &btile[n][0];
                        const int rot = 2 * n + j;
                        float bvr[16];
#pragma unroll
                        for (int sI = 0; sI < 16; sI++) bvr[sI] = brow[4 * sI + rot];
                        ENVGS_LDS_FENCE();
#pragma unroll
                        for (int sI = 0; sI < 16; sI += 4) {
                            acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI], bvr[sI], acc4, 0, 0, 0);
                            accB = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI + 1], bvr[sI + 1], accB, 0, 0, 0);
                            accC = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI + 2], bvr[sI + 2], accC, 0, 0, 0);
                            accD = __builtin_amdgcn_mfma_f32_16x16x4f32(Areg[sI + 3], bvr[sI + 3], accD, 0, 0, 0);
                        }
                    }
                    acc4 = (acc4 + accB) + (accC + accD);
                    const int n = lane & 15, mrow = (lane >> 4) * 4, er = n / 3, cc = n - 3 * er;
                    if (n < 15 && er <= e5) {                                  // this lane's column belongs to an entry of the run
                        const unsigned long long rc = sdesc[buf][el - e5 + er] >> 32;
                        if (rc < A.num_records) {
                            float *ro = A.records + rc * RECW;
                            if (A.M > 0) { ro[(mrow + 0) * 3 + cc] = acc4[0]; ro[(mrow + 1) * 3 + cc] = acc4[1]; ro[(mrow + 2) * 3 + cc] = acc4[2]; ro[(mrow + 3) * 3 + cc] = acc4[3]; }
                            else if (mrow == 0) ro[cc] = acc4[0] * (1.0f / kC0);
                        }
                    }
                    // the same run's contribution to dL/d(basis):  [64 rays x 15 colour-gradient columns] . [15 x 16]  (row 3 e + c = the SH
                    // coefficients of colour c of the run's e-th surfel) -- the tile is the A operand as it stands, the B operand comes straight
                    // from the staged SH blocks: 16 MFMAs per run instead of 48 FMAs and 12 LDS reads per entry and lane
                    if (A.M > 0) {
                        const int kk = lane >> 4, nn = lane & 15;
                        float bvs[4], avs[16];
#pragma unroll
                        for (int ks = 0; ks < 4; ks++) {
                            const int col = 4 * ks + kk, ee = col / 3, c2 = col - 3 * ee;
                            bvs[ks] = 0.f;
                            if (col < 15 && ee <= e5) bvs[ks] = reinterpret_cast<const float *>(&sdat[buf][el - e5 + ee][4])[nn * 3 + c2];
                            const float *arow = &btile[col][0];
#pragma unroll
                            for (int rb = 0; rb < 4; rb++) avs[4 * ks + rb] = arow[16 * rb + nn + 2 * col];
                        }
                        ENVGS_LDS_FENCE();
#pragma unroll
                        for (int ks = 0; ks < 4; ks++)
#pragma unroll
                            for (int rb = 0; rb < 4; rb++)
                                SkM[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(avs[4 * ks + rb], bvs[ks], SkM[rb], 0, 0, 0);
                    }
                }
            };
            for (int el = 0; el < ne; el += 5) {
                entry.template operator()<0>(el);
                if (el + 1 >= ne) break;
                entry.template operator()<1>(el + 1);
                if (el + 2 >= ne) break;
                entry.template operator()<2>(el + 2);
                if (el + 3 >= ne) break;
                entry.template operator()<3>(el + 3);
                if (el + 4 >= ne) break;
                entry.template operator()<4>(el + 4);
            }
        }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int i = 0; i < 4; i++) btile[lane & 15][16 * rb + 4 * (lane >> 4) + i] = SkM[rb][i];
        __syncthreads();
        if (valid) {
            BwdRay B;
            bwd_load_ray(A, r, B);
            BwdAcc acc;
            bwd_init_acc(acc);
            acc.dO0 = dO0; acc.dO1 = dO1; acc.dO2 = dO2; acc.dD0 = dD0; acc.dD1 = dD1; acc.dD2 = dD2;
#pragma unroll
            for (int k = 0; k < 16; k++) acc.Sk[k] = btile[k][lane];
            bwd_store_ray(A, r, B, acc);
        }
    }
}

template __global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64) batch_surfel_bwd<false, false>(const TraceArgs A);
template __global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64) batch_surfel_bwd<false, true>(const TraceArgs A);
template __global__ void __attribute__((amdgpu_waves_per_eu(ENVGS_BSB_WAVES, ENVGS_BSB_WAVES))) __launch_bounds__(64) batch_surfel_bwd<true, false>(const TraceArgs A);

// The hits of SPARSE entries (register_hits; envgs_trace.h: sparse_hits), one LANE per hit: the surfel-major batch kernel above spends a whole
// 64-lane pass on an entry whatever its hit count, so entries that 1-4 of the batch's rays blended (a fifth of the entries of the benchmark
// view, three quarters for incoherent bounce rays) are differentiated here instead, each hit on its own -- its ray's constants, its per-hit
// state row, the surfel's record and SH block gathered per lane -- and write one gradient record PER HIT (no reduction over rays is left to
// do; reduce_surfel_records sums a surfel's records whoever wrote them).  Same formulas as batch_surfel_bwd (IEEE-rounded where that kernel
// uses v_rcp_f32: the contract is 1e-4).  Runs AFTER the batch kernel, which stores the rays' gradients: this one adds to them.
__global__ void __launch_bounds__(256)
sparse_hits_bwd(const TraceArgs A, const int rgbo)
{
    const unsigned filed = __hip_atomic_load(A.counter + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned n = filed < A.sparse_cap ? filed : A.sparse_cap;
    const int nb = (A.D + 1) * (A.D + 1);
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint4 e = A.sparse[i];
        const int slot = (int)e.x, k = (int)e.y, sid = (int)e.z;
        const int r = ray_of(A, slot);
        if (r >= A.R) continue;
        BwdRay B;
        bwd_load_ray(A, r, B);
        float basis[16];
#pragma unroll
        for (int q = 0; q < 16; q++) basis[q] = 0.f;
        sh_basis(A.D, B.ux, B.uy, B.uz, basis);
        const size_t row = state_row0(A, slot, r) + (size_t)k;
        const float4 st0 = A.state[row];
        float4 st1 = make_float4(0.f, 0.f, 0.f, 0.f);
        float sx0 = 0.f, sx1 = 0.f;
        if (!rgbo) {
            if (A.has_others) {
                const float *q = reinterpret_cast<const float *>(state_row1(A, row));
                st1 = make_float4(q[0], q[1], q[2], q[3]); sx0 = q[4]; sx1 = q[5];
            } else st1 = *reinterpret_cast<const float4 *>(state_row1(A, row));
        }
        const float4 *sr = A.srec + (size_t)sid * 4;
        const float4 s0 = sr[0], s1 = sr[1], s2 = sr[2], s3 = sr[3];
        const SurfHit h = hit_surfel(s0, s1, s2, s3, B.ox, B.oy, B.oz, B.dx, B.dy, B.dz);
        float shv[48];
        float col[3]; bool cl[3] = {false, false, false};
        if (A.M > 0) {
            load_sh(A, sid, nb, shv);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; q++) { const float b = basis[q]; c0 += b * shv[q * 3]; c1 += b * shv[q * 3 + 1]; c2 += b * shv[q * 3 + 2]; }
            c0 += 0.5f; c1 += 0.5f; c2 += 0.5f;
            cl[0] = c0 < 0.f; cl[1] = c1 < 0.f; cl[2] = c2 < 0.f;
            col[0] = cl[0] ? 0.f : c0; col[1] = cl[1] ? 0.f : c1; col[2] = cl[2] ? 0.f : c2;
        } else {
            const Feat c = Feat{A.colors, A.f16 != 0}.at((size_t)sid * 3);
            col[0] = c[0]; col[1] = c[1]; col[2] = c[2];
        }
        const float x0 = (!rgbo && A.has_others) ? A.others[2 * sid] : 0.f, x1 = (!rgbo && A.has_others) ? A.others[2 * sid + 1] : 0.f;
        const float alpha = h.alpha, Tb = st0.x, w = alpha * Tb;
        const float sgn = h.denom < 0.0f ? 1.0f : -1.0f;
        const float nf0 = sgn * s3.x, nf1 = sgn * s3.y, nf2 = sgn * s3.z;
        const float inv1m = 1.0f / (1.0f - alpha);
        const float Fsum = B.gR0 * B.fr0 + B.gR1 * B.fr1 + B.gR2 * B.fr2 + B.gD * B.fD + B.gA * B.fA + B.gN0 * B.fN0 + B.gN1 * B.fN1 + B.gN2 * B.fN2 +
                           B.gX0 * B.fX0 + B.gX1 * B.fX1 + B.fT * B.bgdot;
        float gv_ = B.gR0 * col[0] + B.gR1 * col[1] + B.gR2 * col[2], gS = B.gR0 * st0.y + B.gR1 * st0.z + B.gR2 * st0.w;
        if (!rgbo) {
            gv_ += B.gD * h.t + B.gA + B.gN0 * nf0 + B.gN1 * nf1 + B.gN2 * nf2 + B.gX0 * x0 + B.gX1 * x1;
            gS += B.gD * st1.x + B.gA * (1.0f - Tb * (1.0f - alpha)) + B.gN0 * st1.y + B.gN1 * st1.z + B.gN2 * st1.w + B.gX0 * sx0 + B.gX1 * sx1;
        }
        const float dLa = Tb * gv_ - (Fsum - gS) * inv1m;
        const float dc[3] = {cl[0] ? 0.f : w * B.gR0, cl[1] ? 0.f : w * B.gR1, cl[2] ? 0.f : w * B.gR2};
        const float dLG = s0.w * dLa;
        const float dLu = dLG * (-h.G * h.u), dLv = dLG * (-h.G * h.v);
        const float qx = B.ox + h.t * B.dx - s0.x, qy = B.oy + h.t * B.dy - s0.y, qz = B.oz + h.t * B.dz - s0.z;
        const float dq0 = dLu * s1.x + dLv * s2.x, dq1 = dLu * s1.y + dLv * s2.y, dq2 = dLu * s1.z + dLv * s2.z;
        const float cu = dLu / s1.w, cv = dLv / s2.w;
        const float dLt_tot = w * B.gD + dq0 * B.dx + dq1 * B.dy + dq2 * B.dz;
        const float kt = dLt_tot / h.denom;
        const float e0 = dq0 - kt * s3.x, e1 = dq1 - kt * s3.y, e2 = dq2 - kt * s3.z;
        float gw[16];
        gw[0] = -e0; gw[1] = -e1; gw[2] = -e2;
        gw[3] = cu * qx; gw[4] = cu * qy; gw[5] = cu * qz;
        gw[6] = cv * qx; gw[7] = cv * qy; gw[8] = cv * qz;
        const float ws = w * sgn;
        gw[9] = ws * B.gN0 - kt * qx; gw[10] = ws * B.gN1 - kt * qy; gw[11] = ws * B.gN2 - kt * qz;
        gw[12] = -cu * h.u * A.mod; gw[13] = -cv * h.v * A.mod; gw[14] = h.G * dLa; gw[15] = 0.f;
        // the record: (16, 3) SH gradient = basis (x) dL/dcolour (or the 3 colour words), then the 15 geometry words
        const int copy = (slot >> 6) & (NCOPY - 1);
        const size_t ci = (size_t)sid * NCOPY + copy;
        const unsigned long long rec = (unsigned long long)(A.surf_off[ci] - A.surf_cnt[ci]) + (unsigned long long)e.w;
        if (rec < A.num_records) {
            float4 *ro = reinterpret_cast<float4 *>(A.records + rec * RECW);
            if (A.M > 0) {
#pragma unroll
                for (int q = 0; q < 12; q++) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { const int wd = 4 * q + j; v[j] = basis[wd / 3] * dc[wd % 3]; }
                    ro[q] = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else ro[0] = make_float4(dc[0], dc[1], dc[2], 0.f);
#pragma unroll
            for (int q = 0; q < 4; q++) ro[12 + q] = make_float4(gw[4 * q], gw[4 * q + 1], gw[4 * q + 2], gw[4 * q + 3]);
        }
        // the ray's share: added to what batch_surfel_bwd stored (bwd_store_ray is linear in its accumulators)
        float dd0 = 0.f, dd1 = 0.f, dd2 = 0.f;
        if (A.M > 0) {
            float bgx[16], bgy[16], bgz[16];
            sh_basis_grad(A.D, B.ux, B.uy, B.uz, bgx, bgy, bgz);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const float sk = q < nb ? shv[q * 3] * dc[0] + shv[q * 3 + 1] * dc[1] + shv[q * 3 + 2] * dc[2] : 0.f;
                dd0 += bgx[q] * sk; dd1 += bgy[q] * sk; dd2 += bgz[q] * sk;
            }
        }
        const float inv3 = B.il * B.il * B.il;
        const float f0 = h.t * e0 + ((B.dl2 - B.dx * B.dx) * dd0 - B.dy * B.dx * dd1 - B.dz * B.dx * dd2) * inv3;
        const float f1 = h.t * e1 + (-B.dx * B.dy * dd0 + (B.dl2 - B.dy * B.dy) * dd1 - B.dz * B.dy * dd2) * inv3;
        const float f2 = h.t * e2 + (-B.dx * B.dz * dd0 - B.dy * B.dz * dd1 + (B.dl2 - B.dz * B.dz) * dd2) * inv3;
        atomic_add_f32(A.dray_o + 3 * r, e0); atomic_add_f32(A.dray_o + 3 * r + 1, e1); atomic_add_f32(A.dray_o + 3 * r + 2, e2);
        atomic_add_f32(A.dray_d + 3 * r, f0); atomic_add_f32(A.dray_d + 3 * r + 1, f1); atomic_add_f32(A.dray_d + 3 * r + 2, f2);
        if (!rgbo && A.has_others && A.dothers) { atomic_add_f32(A.dothers + 2 * sid, w * B.gX0); atomic_add_f32(A.dothers + 2 * sid + 1, w * B.gX1); }
    }
}

// Stage 2: sum each surfel's (batch, surfel) records into the (zeroed) gradient buffers -- plain stores, every word has one owner; the
// K-buffer pass for overflowed rays runs afterwards and adds to the same buffers atomically (A.reduce_adds, the deferred form: it ran
// BEFORE, and the sum is added to what it left -- still one owner per word).  16 lanes per surfel, 16 B per lane = one
// 256 B record per load instruction; the typical surfel has ~15 records, but a few are seen by thousands of batches: those are deferred
// and summed by the whole workgroup (16 records per instruction) so that no lane group walks a megabyte on its own.
constexpr int RED_LONG = 96;
__device__ __forceinline__ void reduce_store(const TraceArgs &A, const int sid, const int q, const int nb, const float *v)
{
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int wd = 4 * q + e;
        if (wd < 48) {
            float *o = nullptr;
            if (A.M > 0) { if (wd / 3 < nb) o = A.dshs + (size_t)sid * A.M * 3 + wd; }
            else if (wd < 3) o = A.dcolors + (size_t)sid * 3 + wd;
            if (o) *o = A.reduce_adds ? *o + v[e] : v[e];
        } else if (wd < 63) {
            float *o = A.geo_rec + (size_t)sid * GEO + (wd - 48);
            *o = A.reduce_adds ? *o + v[e] : v[e];
        }
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
    const float qq = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    const float inv = qq > 0.0f ? 1.0f / sqrtf(qq) : 0.0f;      // (a zero quaternion has no frame: no ray can have hit it, its gradient is zero -- not 0 * inf)
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


}  // namespace envgs
