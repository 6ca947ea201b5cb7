// trace_kbuffer.hip -- the K-nearest-buffer tracer (T2 trace_fwd, T3 trace_bwd): bounces, rays whose hit list
// overflowed, and callers that pass no list scratch.  One lane = one ray; see trace_common.h.
#include "trace_common.h"

namespace envgs {

// ------------------------------------------------------------------------------------------ T2 ---
__global__ void __launch_bounds__(64)
trace_fwd(const TraceArgs A, const int ray_h, const int ray_w)
{
    __shared__ int stk[STACK][64];
    const int lane = threadIdx.x;
    // overflow pass after the list path: counter[1] holds the longest list of this call, counter[21] the rays that found no room in the compact
    // per-hit buffers -- nothing overflowed, nothing to do
    if (A.only_overflow && (int)__hip_atomic_load(A.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= A.cap &&
        __hip_atomic_load(A.counter + 21, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;      // ([21]: rays the compact row scan handed over)
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
                            atomic_add_f32(A.wet + sid, w);          // every stage: a surfel blended only by bounce rays is visible too
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

// K-buffer backward (re-traces).  Used for bounce-free rays whose hit list overflowed, and when no list was kept.
__global__ void __launch_bounds__(64)
trace_bwd(const TraceArgs A, const int ray_h, const int ray_w)
{
    __shared__ int stk[STACK][64];
    __shared__ float fld[NFLD][65];                 // row stride 65: lanes 48..62 read 15 different rows of one column conflict-free
    const int lane = threadIdx.x;
    if (A.only_overflow && (int)__hip_atomic_load(A.counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= A.cap &&
        __hip_atomic_load(A.counter + 21, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;      // ([21]: rays the compact row scan handed over)
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


}  // namespace envgs
