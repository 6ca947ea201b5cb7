// trace_collect.hip -- list path, step 1: the coherence sort key of a ray and the unordered hit collection (cooperative workgroup-per-batch kernel over the
// 4-wide nodes = the product path; kept for A/B measurements behind envgs_debug_set: per-ray kernel, one-wavefront packet kernel over the
// binary nodes, packet kernel over the 4-wide nodes).
#include "trace_common.h"

namespace envgs {

#ifdef ENVGS_DIAG   // superseded collection kernel: A/B measurements and tests only (libenvgs_hip_diag.so), not in the product library
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
#endif  // ENVGS_DIAG

// Packet form of collect_hits for coherence-sorted rays.  The 64 rays of a batch mostly walk the SAME nodes (measured on the bench
// scene: the union of the surfels a batch finds is 2.4x what one of its rays finds), so the wavefront walks the tree ONCE with a single,
// wave-uniform stack: node and surfel records come in through the scalar unit (s_load: one 64 B fetch per wavefront instead of up to
// 64 gathers), every lane tests its own ray against them with its own termination bound, and control flow never diverges.  A lane
// that is pruned simply stops passing box tests.  Visit order (near child first by majority vote) only affects how early the bounds
// tighten: the lists are sorted afterwards.
// (8 waves per SIMD: with two segments in flight ~10 k wavefronts want a slot; the (n - o) * (1/d) slab form is kept on purpose -- the
//  one-fma form n*(1/d) - o/d needs an error margin proportional to |o/d|, and in a packet ONE ray with a tiny direction component then
//  drags the whole wavefront through nodes nobody hits: measured +0.6 ms)
#ifdef ENVGS_DIAG   // superseded collection kernel: A/B measurements and tests only (libenvgs_hip_diag.so), not in the product library
__global__ void __attribute__((amdgpu_waves_per_eu(8, 8))) __launch_bounds__(64)
collect_hits_packet(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ srec)
{
    __shared__ int stk[PSTACK];
    const int lane = threadIdx.x;
    const int slimit = (A.exp & 1024) ? 2 : PSTACK;       // (test switch: forces the overflow hand-off)
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
        bool ovf = false;                                  // wave-uniform: a child did not fit on the packet stack
        visits += (unsigned)__popcll(__ballot(valid));
        while (true) {
            if (cur < 0) {
                if (sp == 0) break;
                --sp;
                // LDS only (see PSTACK): with an HBM overflow area behind it the compiler merges the two reads into ONE flat load of a selected
                // address, whose s_waitcnt vmcnt(0) also waits for every list store still in flight -- on every pop
                cur = __builtin_amdgcn_readfirstlane(stk[sp]);
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
                if (sp < slimit) stk[sp++] = farc; else ovf = true;
                cur = leftFirst ? lc : rc;
            }
            else if (goL) cur = lc;
            else if (goR) cur = rc;
            else cur = -1;
        }
        if (ovf) {
            // a postponed child was dropped: this batch's lists are incomplete.  Mark every ray as overflowed (hit_cnt > cap) so that it is
            // traced by the K-buffer kernels instead (per-lane stacks, no packet stack), and count the event (counters[20]).
            n = A.cap + 1;
            if (lane == 0) atomicAdd(A.counter + 20, 1u);
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
#endif  // ENVGS_DIAG

// The same traversal over the 4-wide nodes (trace_bvh.hip: node4[i] = the grandchildren of binary node i): half the steps, and each step is
// one scalar-load round trip plus the stack / mask bookkeeping of the scalar unit, which is what the binary walk spends most of its time on.
// Children that any ray hits are entered nearest first, ordered by the entry distance of each child's first hitting lane (the rays of a
// batch are coherent; the order only affects how early the termination bounds tighten).  `visits` counts 64 B units (two per wide node).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef ENVGS_DIAG   // superseded collection kernel: A/B measurements and tests only (libenvgs_hip_diag.so), not in the product library
__global__ void __attribute__((amdgpu_waves_per_eu(6, 8))) __launch_bounds__(64)
collect_hits_packet4(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ nodes4, const float4 *__restrict__ srec)
{
    __shared__ int stk[PSTACK];
    const int lane = threadIdx.x;
    const int slimit = (A.exp & 1024) ? 2 : PSTACK;       // (test switch: forces the overflow hand-off)
    unsigned visits = 0, found_tot = 0;
    unsigned psteps = 0, pleaves = 0;                     // per-PACKET counts: wide nodes fetched, surfel records fetched (deduplicated byte model of bench.py)
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
        bool ovf = false;                                  // wave-uniform: a child did not fit on the packet stack
        const f32x2 o2x = {ox, ox}, o2y = {oy, oy}, o2z = {oz, oz}, i2x = {ix, ix}, i2y = {iy, iy}, i2z = {iz, iz};
        while (true) {
            if (cur < 0) {
                if (sp == 0) break;
                --sp;
                // LDS only (see PSTACK): with an HBM overflow area behind it the compiler merges the two reads into ONE flat load of a selected
                // address, whose s_waitcnt vmcnt(0) also waits for every list store still in flight -- on every pop
                cur = __builtin_amdgcn_readfirstlane(stk[sp]);
            }
            const float4 *nd = nodes4 + (size_t)cur * 8;
            psteps++;
            float4 qa[4], qb[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { qa[c] = nd[2 * c]; qb[c] = nd[2 * c + 1]; }
            int key[4], ref[4];
            int ninner = 0;                                   // internal children some ray enters
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int ch = __builtin_amdgcn_readfirstlane(__float_as_int(qb[c].z));
                // both planes of an axis in one packed instruction: (lo, hi) - o, then * 1/d
                const f32x2 sx = (f32x2{qa[c].x, qa[c].y} - o2x) * i2x, sy = (f32x2{qa[c].z, qa[c].w} - o2y) * i2y,
                            sz = (f32x2{qb[c].x, qb[c].y} - o2z) * i2z;
                const float tn = fmaxf(fmaxf(fminf(sx.x, sx.y), fminf(sy.x, sy.y)), fminf(sz.x, sz.y));
                const float tf = fminf(fminf(fmaxf(sx.x, sx.y), fmaxf(sy.x, sy.y)), fmaxf(sz.x, sz.y));
                const bool hit = valid && (tn <= tf) && (tf >= tmin) && (tn <= tkill);
                const unsigned long long m = __ballot(hit);
                key[c] = 0x7fffffff; ref[c] = -1;
                if (m != 0ull) {
                    if (ch < 0) {
                        const int sid = ~ch;
                        pleaves++;
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
                        ninner++;
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
            if (ninner >= 2) {                                // (most steps enter at most one internal child: nothing to order, nothing to push)
                ENVGS_CSWAP(0, 1) ENVGS_CSWAP(2, 3) ENVGS_CSWAP(0, 2) ENVGS_CSWAP(1, 3) ENVGS_CSWAP(1, 2)
                // nearest next; the others go on the stack far to near
#pragma unroll
                for (int c = 3; c >= 1; c--)
                    if (ref[c] >= 0) { if (sp < slimit) stk[sp++] = ref[c]; else ovf = true; }
                cur = ref[0];
            } else {
                cur = max(max(ref[0], ref[1]), max(ref[2], ref[3]));      // the one entered child, or -1
            }
#undef ENVGS_CSWAP
        }
        if (ovf) {
            // a postponed child was dropped: this batch's lists are incomplete.  Mark every ray as overflowed (hit_cnt > cap) so that it is
            // traced by the K-buffer kernels instead (per-lane stacks, no packet stack), and count the event (counters[20]).
            n = A.cap + 1;
            if (lane == 0) atomicAdd(A.counter + 20, 1u);
        }
        if (valid) { A.hit_cnt[r] = n; found_tot += (unsigned)n; }
        int mx = n;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        if (lane == 0) atomicMax((int *)(A.counter + 1), mx);
    }
    if (A.stats) {
        const float ff = wave_sum((float)found_tot);
        if (lane == 0) {
            atomicAdd(A.stats + 1, (unsigned long long)visits); atomicAdd(A.stats + 3, (unsigned long long)ff);
            atomicAdd(A.stats + 4, (unsigned long long)psteps); atomicAdd(A.stats + 5, (unsigned long long)pleaves);
        }
    }
}
#endif  // ENVGS_DIAG

// Cooperative packet traversal: COOP_W wavefronts share one 64-ray batch.  With one wavefront per batch the kernel lasts as long as its
// longest batch (a chain of ~2000 dependent node steps against a mean of ~830) while most of the chip has already drained; here every
// wavefront of the workgroup holds the SAME 64 rays (lane = ray) and the subtrees are what is divided:
//   phase A  the top of the tree is expanded level by level (the wavefronts split each level's nodes) until at least COOP_FRONT
//            entered subtrees exist; they are ranked by entry distance,
//   phase B  every wavefront pulls the next subtree from a shared counter and walks it depth first, nearest child first, on a private stack.
// What the rays of the batch share lives in LDS and is updated with LDS atomics: the per-ray list cursor (ds_add_rtn), the per-ray optical
// depth bins of the termination bound (ds_add_f32) -- so a wavefront deep in a far subtree is pruned by the hits another one finds
// near the origin.  The bound only ever tightens and a stale (larger) one only collects more, so the collected set is a superset of
// what the compositing needs whatever the interleaving; the lists are sorted afterwards.
constexpr int COOP_W = 4;
constexpr float COOP_REFRESH_OD = 1.0f;   // recompute a ray's bound when its optical depth has grown by this much (the bound cuts at ~9.5).  Round 5, with the
                                          // ray-major bin table: 0.5 -> collect 1.733 ms, 68.9 M hits found; 1.0 -> 1.718 ms, 69.3 M; 2.0 -> 1.745 ms, 70.5 M (profiles/r05_ab_collect.txt)
constexpr int COOP_ODROW = 36;            // floats per ray of the bin table: 32 bins + pad -- rows stay 16 B aligned (ds_read_b128) and 64 rows spread over the banks
constexpr int COOP_NBIN = 32;             // distance bins of the termination bound (LDS: the update is one ds_add_f32 whatever their number)
constexpr int COOP_STK = 96;
constexpr int COOP_FRONT = 64;            // stop expanding once a level has this many entered subtrees (the next level holds at most 4x that).  Measured
                                          // (bench scene): 16 -> 93.8 M hits found, collect 2.70 ms; 32 -> 83.3 M, 1.97 ms; 64 -> 79.1 M, 1.87 ms (one wavefront
                                          // per batch, depth first: 85.2 M, 3.35 ms): the ranked frontier is a better visiting order than the local one
constexpr int COOP_ITEMS = 256;           // >= 4 * (largest front - 1)
constexpr int COOP_QFLUSH = 8;           // deferred exact tests: a wavefront's queue of entered leaf slots is worked off when it holds this many ...
constexpr int COOP_QCAP = COOP_QFLUSH + 3 + 1;   // ... (a step adds at most four)
static_assert(COOP_ODROW % 4 == 0, "the bin rows are read with ds_read_b128");
// Round 6: the optical-depth bins are FIXED POINT (2^-16).  ds_add_f32 is microcoded on gfx950 -- measured (scratch/lds_atomic_rate.hip) 193 cycles of
// the CU's LDS unit per 64-lane instruction, 43 with 14 live lanes, against 7 / 5 for ds_add_u32 -- and the walk issued two of them per leaf test:
// a quarter of the kernel's duration during which every other LDS operation of the CU (stack pops, frontier items) queued behind them.  Each hit's
// -ln(1 - alpha) is rounded DOWN, so the sums stay lower bounds of the true optical depth and the bound stays conservative; integer sums are also
// independent of the order in which the four wavefronts add.
constexpr float COOP_OD_SCALE = 65536.0f;
constexpr unsigned COOP_KILL_Q = (unsigned)(KILL_OD * 65536.0f);
struct CoopLds {                      // (refresh_bound reads od[] as 16 B vectors: the OBJECT is 16 B aligned -- checked in the kernel, see there)
    unsigned od[64][COOP_ODROW];              // ray-major (round 5): a ray's 32 bins are eight ds_read_b128, four in flight at a time, instead of 31 serialised ds_read_b32
    unsigned odtot[64];
    int cnt[64];
    unsigned long long items[2][COOP_ITEMS];      // entry-distance bits << 32 | wide-node index
    int stk[COOP_W][COOP_STK];
    int nitems[2];
    int next, ovf, batch, pad;
};
// the queue of the deferred form: surfel id, the ballot of the rays that passed the slot's slab test, and the staged 64 B surfel records
struct CoopQueue {
    int sid[COOP_W][COOP_QCAP];
    unsigned mlo[COOP_W][COOP_QCAP], mhi[COOP_W][COOP_QCAP];
    float4 rec[COOP_W][COOP_QCAP][4];
};

// DEFER (round 5 experiment, diagnostic library only -- measured SLOWER, profiles/r05_ab_collect.txt): the exact ray / surfel tests leave the
// traversal's dependent chain.  The product form (DEFER = false) tests a leaf slot the moment some ray's slab test enters it: an s_load of the
// 64 B record, ~60 VALU, a returning LDS atomic for the list slot, a global store and two more LDS atomics, all in front of the next node's
// s_load.  The deferred form only QUEUES the slot (surfel id + the ballot of the entering rays, lane 0, LDS); when eight are queued the wavefront
// fetches all their records with ONE vector load (lane l: float4 l & 3 of entry l >> 2), stages them in LDS and runs the exact tests from
// broadcast ds_reads.  Same node steps / leaf tests / hits within 1.5 %, bit-identical composited lists -- and 1.92 ms per launch against 1.77:
// the in-kernel timers of the same round (ENVGS_COOP_TIMING) show why: a node's s_load costs 310 cycles of a 3 500-cycle step and the leaf
// records 600 of the 1 270 cycles of a leaf test; the rest is INSTRUCTION ISSUE (~200 instructions per step at ~1 instruction per 4 cycles per
// SIMD, 4.5 wavefronts taking turns), and the deferred form adds instructions (LDS staging, ballot bits, VGPR operands) while removing latency
// that was not the bound.
template <bool DEFER, int WAVES>
__global__ void __launch_bounds__(64 * COOP_W, WAVES)
collect_hits_coop(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ nodes4, const float4 *__restrict__ srec)
{
    // (refresh_bound's ds_read_b128 of L.od needs L 16 B aligned.  It IS -- L is the kernel's first LDS object -- but SAYING so, on the
    //  type (alignas) or on this declaration, changes the backend's code for the worse: collect 1.59 -> 1.66 ms per launch, both ways, measured
    //  twice in round 6 (profiles/r06_ab_collect.txt).  So the alignment is CHECKED instead: the diagnostic build traps on a misaligned L.)
    __shared__ CoopLds L;
    __shared__ CoopQueue Q;
#ifdef ENVGS_DIAG
    if ((reinterpret_cast<size_t>(&L.od[0][0]) & 15) != 0) __builtin_trap();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slimit = (A.exp & 1024) ? 2 : COOP_STK;     // (test switch: forces the overflow hand-off)
    unsigned found_tot = 0, psteps = 0, pleaves = 0;
    unsigned long long cyc_expand = 0, cyc_walk = 0, cyc_wait = 0;      // where the wavefront's time goes (diagnostics, stats[6..8])
    float rlx, rly, rlz, rhx, rhy, rhz;                   // the scene box = union of the root's two child boxes
    {
        const float4 n0 = nodes[0], n1 = nodes[1], n2 = nodes[2];
        rlx = fminf(n0.x, n1.z); rly = fminf(n0.y, n1.w); rlz = fminf(n0.z, n2.x);
        rhx = fmaxf(n0.w, n2.y); rhy = fmaxf(n1.x, n2.z); rhz = fmaxf(n1.y, n2.w);
    }
    const int home = xcc_id();
    const int nbatch = A.batch1 - A.batch0;
    // (what other wavefronts add to the bins must be re-read: relaxed workgroup-scope atomic loads keep the LDS address space, a volatile
    //  generic pointer would turn them into flat loads)
#define ENVGS_LDS_READ(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    int *stk = L.stk[wave];
    while (true) {
        if (wave == 0) {
            const int b = fetch_batch(A.counter + 32 + 8 * A.seg, nbatch, home, lane);
            if (lane == 0) { L.batch = b; L.nitems[0] = 1; L.nitems[1] = 0; L.items[0][0] = 0ull; L.next = 0; L.ovf = 0; }
        }
#pragma unroll
        for (int q = 0; q < (int)(sizeof(L.od) / sizeof(unsigned)) / (64 * COOP_W); q++) (&L.od[0][0])[q * 64 * COOP_W + tid] = 0u;
        if (tid < 64) { L.odtot[tid] = 0u; L.cnt[tid] = 0; }
        __syncthreads();
        const int fb = L.batch;
        if (fb < 0) break;
        const unsigned long long c0 = __builtin_readcyclecounter();
        const int base = (A.batch0 + fb) << 6;
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
            bin_w = span * (1.00001f / (float)(COOP_NBIN - 1));
            inv_bin_w = 1.0f / bin_w;
        }
        const float tk_open = valid ? 3.0e38f : -3.0e38f;     // lanes without a ray never pass a slab test
        float tkill = tk_open;
        unsigned seen = 0u;                              // the ray's (fixed-point) optical depth when its bound was last recomputed
        int pend = 0;
        uint2 *list = A.hits + (size_t)rr * A.cap;
        const f32x2 o2x = {ox, ox}, o2y = {oy, oy}, o2z = {oz, oz}, i2x = {ix, ix}, i2y = {iy, iy}, i2z = {iz, iz};
        int oct = 8;
#ifndef ENVGS_COOP_NO_OCTANTS
        {
            const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
            auto sign_of = [&](const float d, const float inv) -> int {       // 0: every ray's component > 0, 1: every one < 0, 2: mixed / zero / not finite
                const unsigned long long pos = __builtin_amdgcn_ballot_w64(valid && d > 0.0f && inv < 3.0e38f);
                const unsigned long long neg = __builtin_amdgcn_ballot_w64(valid && d < 0.0f && inv > -3.0e38f);
                return pos == vm ? 0 : (neg == vm ? 1 : 2);
            };
            const int ux = sign_of(dx, ix), uy = sign_of(dy, iy), uz = sign_of(dz, iz);
            if (vm != 0ull && ux < 2 && uy < 2 && uz < 2) oct = ux | (uy << 1) | (uz << 2);
        }
#endif

        // one wide node for all 64 rays: slab tests of its four slots, exact tests of the leaf slots some ray may hit; returns the internal
        // children some ray enters (ref) with the entry distance of each one's first hitting lane (key)
        // The walk of a batch, instantiated per direction OCTANT (round 5): OCT = sign bits of (dx, dy, dz) when all 64 rays share strictly signed
        // finite components (~95 % of the coherence-sorted batches), 8 = the generic form.
        unsigned long long c1 = c0;
        bool ovf = false;
        auto walk = [&]<int OCT>() {
        int qn = 0;                                            // queued leaf slots of this wavefront (DEFER)
        // an accepted hit: list slot from the ray's LDS cursor, optical depth into the ray's distance bin
        auto record_hit = [&](const SurfHit &h, const int sid) {
            const int slot = atomicAdd(&L.cnt[lane], 1);
            if (slot < A.cap) list[slot] = make_uint2(__float_as_uint(h.t), (unsigned)sid);
            const float x = (h.t - tA) * inv_bin_w;
            int b = x <= 0.0f ? 0 : (int)ceilf(x + 1e-3f);
            b = b > COOP_NBIN - 1 ? COOP_NBIN - 1 : b;
            const unsigned dep = (unsigned)(-__logf(1.0f - h.alpha) * COOP_OD_SCALE);      // rounded down: alpha <= 0.99 -> at most 4.61 * 2^16
            atomicAdd(&L.od[lane][b], dep);
            atomicAdd(&L.odtot[lane], dep);
        };
        // look at the bound -- and recompute it (eight ds_read_b128 and ~125 VALU, as much as two leaf tests) only when some ray that can be cut at
        // all has gathered noticeably more optical depth than at its last recomputation
        auto refresh_bound = [&]() {
            const unsigned tot = ENVGS_LDS_READ(L.odtot[lane]);
            if (__builtin_amdgcn_ballot_w64(tot >= COOP_KILL_Q && tot > seen + (unsigned)(COOP_REFRESH_OD * COOP_OD_SCALE)) != 0ull) {
                seen = tot;
                unsigned cum = 0u; int kb = COOP_NBIN - 1;
                // (what other wavefronts add meanwhile may or may not be seen: either way the sums are lower bounds of the true optical depth, the bound stays conservative)
                const uint4 *row = reinterpret_cast<const uint4 *>(&L.od[lane][0]);
                asm volatile("" ::: "memory");                 // (re-read: nothing cached from an earlier refresh)
#pragma unroll
                for (int h = 0; h < COOP_NBIN / 16; h++) {    // sixteen bins per round: four ds_read_b128 in flight, 16 VGPRs
                    uint4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = row[4 * h + q];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const unsigned e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int bin = 16 * h + 4 * q + j;
                            if (bin < COOP_NBIN - 1) { cum += e[j]; kb = (cum >= COOP_KILL_Q && kb == COOP_NBIN - 1) ? bin : kb; }
                        }
                    }
                }
                tkill = kb < COOP_NBIN - 1 ? tA + (float)kb * bin_w : tk_open;
            }
        };
        // DEFER: the queued slots' records in one vector load, staged in LDS, tested from broadcast reads
        auto flush = [&]() {
            if (qn == 0) return;
            const int e = lane >> 2;
            if (e < qn) Q.rec[wave][e][lane & 3] = srec[(size_t)Q.sid[wave][e] * 4 + (lane & 3)];
            __builtin_amdgcn_wave_barrier();                   // (one wavefront: its LDS operations execute in order; this only pins the compiler's order)
            for (int k = 0; k < qn; k++) {
                const float4 s0 = Q.rec[wave][k][0], s1 = Q.rec[wave][k][1], s2 = Q.rec[wave][k][2], s3 = Q.rec[wave][k][3];
                const int sid = Q.sid[wave][k];
                const unsigned mw = lane < 32 ? Q.mlo[wave][k] : Q.mhi[wave][k];
                const bool hit = ((mw >> (lane & 31)) & 1u) != 0u;
                const SurfHit h = hit_surfel(s0, s1, s2, s3, ox, oy, oz, dx, dy, dz);
                if (hit && h.ok && h.t > tmin && h.t <= tkill) record_hit(h, sid);
            }
            __builtin_amdgcn_wave_barrier();
            qn = 0;
            refresh_bound();
        };
        // (kr[c]: entry-distance bits << 32 | child reference of an entered internal child -- the frontier's item format; 0x7fffffff'ffffffff = not entered.
        //  One aligned scalar pair per child: a compare-exchange of the ordering network is one s_cmp on the high words and two 64-bit selects)
        auto step = [&](const int cur_, unsigned long long (&kr)[4]) -> int {
            const int cur = __builtin_amdgcn_readfirstlane(cur_);      // (the walk's `cur` reaches here through a vector phi: without this the node address is computed in VALU and read back lane by lane)
            const float4 *nd = nodes4 + (size_t)cur * 8;
            psteps++;
            float4 qa[4], qb[4];
#if defined(ENVGS_COOP_TIMING) && ENVGS_COOP_TIMING == 2      // scratch/ measurement build: stats[6] = node loads, stats[8] = the four slots (slab tests, ballots, leaf tests)
            const unsigned long long ta = __builtin_readcyclecounter();
            {
                typedef float f8_ __attribute__((ext_vector_type(8)));
                f8_ r0, r1, r2, r3;
                asm volatile("s_load_dwordx8 %0, %4, 0\n s_load_dwordx8 %1, %4, 32\n s_load_dwordx8 %2, %4, 64\n s_load_dwordx8 %3, %4, 96\n s_waitcnt lgkmcnt(0)"
                             : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3) : "s"(nd));
                qa[0] = make_float4(r0[0], r0[1], r0[2], r0[3]); qb[0] = make_float4(r0[4], r0[5], r0[6], r0[7]);
                qa[1] = make_float4(r1[0], r1[1], r1[2], r1[3]); qb[1] = make_float4(r1[4], r1[5], r1[6], r1[7]);
                qa[2] = make_float4(r2[0], r2[1], r2[2], r2[3]); qb[2] = make_float4(r2[4], r2[5], r2[6], r2[7]);
                qa[3] = make_float4(r3[0], r3[1], r3[2], r3[3]); qb[3] = make_float4(r3[4], r3[5], r3[6], r3[7]);
            }
            const unsigned long long tb = __builtin_readcyclecounter();
            cyc_expand += tb - ta;
#else
#pragma unroll
            for (int c = 0; c < 4; c++) { qa[c] = nd[2 * c]; qb[c] = nd[2 * c + 1]; }
#endif
            int ninner = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int ch = __builtin_amdgcn_readfirstlane(__float_as_int(qb[c].z));
                kr[c] = 0x7fffffffffffffffull;
                if (c >= 2 && ch == WIDE_EMPTY) continue;        // (a fifth of the slots of the bench tree: bottom nodes with leaf children; slots 0 and 1 are
                                                                 //  always filled -- fit_nodes -- so their boxes are loaded with the node, not behind a branch)
                const f32x2 sx = (f32x2{qa[c].x, qa[c].y} - o2x) * i2x, sy = (f32x2{qa[c].z, qa[c].w} - o2y) * i2y,
                            sz = (f32x2{qb[c].x, qb[c].y} - o2z) * i2z;
                float tn, tf;
                if constexpr (OCT == 8) {
                    tn = fmaxf(fmaxf(fmaxf(fminf(sx.x, sx.y), fminf(sy.x, sy.y)), fminf(sz.x, sz.y)), tmin);
                    tf = fminf(fminf(fminf(fmaxf(sx.x, sx.y), fmaxf(sy.x, sy.y)), fmaxf(sz.x, sz.y)), tkill);
                } else {
                    // every ray of the batch has the SAME strictly signed, finite direction components (decided per batch): the entry / exit plane of
                    // each axis is known at compile time -- lo for a positive component, hi for a negative one -- and the six min / max that ordered
                    // them are gone (11 instead of 17 VALU per slot; the values, hence the traversal, are bit-identical to the generic form)
                    const float nx = (OCT & 1) ? sx.y : sx.x, fx = (OCT & 1) ? sx.x : sx.y;
                    const float ny = (OCT & 2) ? sy.y : sy.x, fy = (OCT & 2) ? sy.x : sy.y;
                    const float nz = (OCT & 4) ? sz.y : sz.x, fz = (OCT & 4) ? sz.x : sz.y;
                    tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), tmin);
                    tf = fminf(fminf(fminf(fx, fy), fz), tkill);
                }
                const bool hit = tn <= tf;                        // (tmin <= tkill always, so this is the three-way test of the other kernels)
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
#if defined(ENVGS_COOP_TIMING) && ENVGS_COOP_TIMING == 3      // scratch/ measurement build: stats[6] = leaf slots slab-tested, stats[8] = lanes that passed a leaf slot's slab test
                if (ch < 0) { cyc_expand += 1ull; cyc_wait += (unsigned long long)__builtin_popcountll(m); }
#endif
                if (m != 0ull) {
                    if (ch < 0) {
                        const int sid = ~ch;
                        pleaves++;
                        if (DEFER) {
                            if (lane == 0) { Q.sid[wave][qn] = sid; Q.mlo[wave][qn] = (unsigned)m; Q.mhi[wave][qn] = (unsigned)(m >> 32); }
                            qn++;
                        } else {
#if defined(ENVGS_COOP_TIMING) && ENVGS_COOP_TIMING == 1      // scratch/ measurement build: where the immediate form's leaf test spends its cycles (stats[6] = whole leaf tests, stats[8] = their record loads)
                            const unsigned long long ta = __builtin_readcyclecounter();
                            typedef float f8_ __attribute__((ext_vector_type(8)));
                            f8_ r0, r1;
                            const float4 *sr = srec + (size_t)sid * 4;
                            asm volatile("s_load_dwordx8 %0, %2, 0\n s_load_dwordx8 %1, %2, 32\n s_waitcnt lgkmcnt(0)" : "=&s"(r0), "=&s"(r1) : "s"(sr));
                            const unsigned long long tb = __builtin_readcyclecounter();
                            const SurfHit h = hit_surfel(make_float4(r0[0], r0[1], r0[2], r0[3]), make_float4(r0[4], r0[5], r0[6], r0[7]),
                                                         make_float4(r1[0], r1[1], r1[2], r1[3]), make_float4(r1[4], r1[5], r1[6], r1[7]), ox, oy, oz, dx, dy, dz);
                            if (hit && h.ok && h.t > tmin && h.t <= tkill) record_hit(h, sid);
                            pend++;
                            asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
                            const unsigned long long tc = __builtin_readcyclecounter();
                            cyc_expand += tc - ta; cyc_wait += tb - ta;
#else
                            const float4 *sr = srec + (size_t)sid * 4;
                            const SurfHit h = hit_surfel(sr[0], sr[1], sr[2], sr[3], ox, oy, oz, dx, dy, dz);
                            if (hit && h.ok && h.t > tmin && h.t <= tkill) record_hit(h, sid);
                            pend++;
#endif
                        }
                    } else {
                        const int fl = (int)__builtin_ctzll(m);
                        kr[c] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(__float_as_int(tn), fl) << 32) | (unsigned)ch;      // tn >= tmin >= 0: the float bits order like integers
                        ninner++;
                    }
                }
            }
#if defined(ENVGS_COOP_TIMING) && ENVGS_COOP_TIMING == 2
            asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
            cyc_wait += __builtin_readcyclecounter() - tb;
#endif
            if (DEFER) {
                if (qn >= COOP_QFLUSH) flush();
            } else if (pend >= 3) {        // look at the bound every third leaf test (a stale bound only collects a little more)
                pend = 0;
                refresh_bound();
            }
            return ninner;
        };

        // ---- phase A: expand the top of the tree level by level
        int buf = 0;
        for (int level = 0; level < 12; level++) {
            const int ncur = L.nitems[buf];
            if (ncur >= COOP_FRONT || ncur == 0) break;
            for (int i = wave; i < ncur; i += COOP_W) {
                const int cur = __builtin_amdgcn_readfirstlane((int)(unsigned)L.items[buf][i]);
                unsigned long long kr[4];
                if (step(cur, kr) > 0 && lane == 0) {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if ((int)(unsigned)kr[c] >= 0) L.items[buf ^ 1][atomicAdd(&L.nitems[buf ^ 1], 1)] = kr[c];
                }
            }
            if (DEFER) flush();                                 // (the next level's slab tests see this level's hits)
            __syncthreads();
            if (tid == 0) L.nitems[buf] = 0;
            buf ^= 1;
            __syncthreads();
        }
        const int nfin = L.nitems[buf];
        if (wave == 0 && nfin > 1) {           // rank the subtrees by entry distance (keys are unique: the node index is part of them)
            constexpr int PER = COOP_ITEMS / 64;
            unsigned long long mine[PER]; int rank[PER];
#pragma unroll
            for (int e = 0; e < PER; e++) { mine[e] = lane + 64 * e < nfin ? L.items[buf][lane + 64 * e] : ~0ull; rank[e] = 0; }
            for (int j = 0; j < nfin; j++) {
                const unsigned long long o = L.items[buf][j];
#pragma unroll
                for (int e = 0; e < PER; e++) rank[e] += o < mine[e] ? 1 : 0;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): every lane's reads have returned before the slots are rewritten
#pragma unroll
            for (int e = 0; e < PER; e++)
                if (lane + 64 * e < nfin) L.items[buf][rank[e]] = mine[e];
        }
        __syncthreads();

        c1 = __builtin_readcyclecounter();
        // ---- phase B: subtrees from the shared counter, depth first on the private stack
        while (true) {
            int idx = 0;
            if (lane == 0) idx = atomicAdd(&L.next, 1);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= nfin) break;
            int cur = __builtin_amdgcn_readfirstlane((int)(unsigned)L.items[buf][idx]);
            int sp = 0;
            while (true) {
                if (cur < 0) {
                    if (sp == 0) break;
                    --sp;
                    cur = __builtin_amdgcn_readfirstlane(stk[sp]);
                }
                unsigned long long kr[4];
                const int ninner = step(cur, kr);
#define ENVGS_CSWAP(a, b) { const bool sw = (unsigned)(kr[a] >> 32) > (unsigned)(kr[b] >> 32); const unsigned long long lo_ = sw ? kr[b] : kr[a], hi_ = sw ? kr[a] : kr[b]; kr[a] = lo_; kr[b] = hi_; }
                if (ninner >= 2) {
                    ENVGS_CSWAP(0, 1) ENVGS_CSWAP(2, 3) ENVGS_CSWAP(0, 2) ENVGS_CSWAP(1, 3) ENVGS_CSWAP(1, 2)
#pragma unroll
                    for (int c = 3; c >= 1; c--)
                        if ((int)(unsigned)kr[c] >= 0) { if (sp < slimit) stk[sp++] = (int)(unsigned)kr[c]; else ovf = true; }
                    cur = (int)(unsigned)kr[0];
                } else {
                    cur = max(max((int)(unsigned)kr[0], (int)(unsigned)kr[1]), max((int)(unsigned)kr[2], (int)(unsigned)kr[3]));
                }
#undef ENVGS_CSWAP
            }
        }
        if (DEFER) flush();
        };
        switch (oct) {
            case 0: walk.template operator()<0>(); break;
            case 1: walk.template operator()<1>(); break;
            case 2: walk.template operator()<2>(); break;
            case 3: walk.template operator()<3>(); break;
            case 4: walk.template operator()<4>(); break;
            case 5: walk.template operator()<5>(); break;
            case 6: walk.template operator()<6>(); break;
            case 7: walk.template operator()<7>(); break;
            default: walk.template operator()<8>(); break;
        }
        if (ovf && lane == 0) L.ovf = 1;
        const unsigned long long c2 = __builtin_readcyclecounter();
        __syncthreads();
        const unsigned long long c3 = __builtin_readcyclecounter();
#if defined(ENVGS_COOP_TIMING) && ENVGS_COOP_TIMING == 3
        (void)c0; (void)c1; (void)c2; (void)c3;               // (counters, not cycles, in the slots)
#elif defined(ENVGS_COOP_TIMING)
        cyc_walk += c2 - c0; (void)c1; (void)c3;              // expansion + walks; the other two slots hold the leaf-test split
#else
        cyc_expand += c1 - c0; cyc_walk += c2 - c1; cyc_wait += c3 - c2;
#endif
        if (wave == 0) {
            int n = L.cnt[lane];
            if (L.ovf) {
                // a postponed child was dropped: this batch's lists are incomplete.  Mark every ray as overflowed (hit_cnt > cap) so that it is
                // traced by the K-buffer kernels instead (per-lane stacks), and count the event (counters[20]).
                n = A.cap + 1;
                if (lane == 0) atomicAdd(A.counter + 20, 1u);
            }
            if (valid) { A.hit_cnt[r] = n; found_tot += (unsigned)n; }
            int mx = n;
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
            if (lane == 0) atomicMax((int *)(A.counter + 1), mx);
            if (A.batch_cnt) {                         // rows of the compact per-hit buffers this batch needs (envgs_trace.h: compact_rows)
                const float rows = wave_sum((valid && n <= A.cap) ? (float)n : 0.f);
                if (lane == 0) A.batch_cnt[A.batch0 + fb] = (unsigned)rows;
            }
        }
        __syncthreads();
    }
    if (A.stats) {
        const float ff = wave_sum((float)found_tot);
        if (lane == 0) {
            if (wave == 0) atomicAdd(A.stats + 3, (unsigned long long)ff);
            atomicAdd(A.stats + 4, (unsigned long long)psteps); atomicAdd(A.stats + 5, (unsigned long long)pleaves);
            atomicAdd(A.stats + 6, cyc_expand); atomicAdd(A.stats + 7, cyc_walk); atomicAdd(A.stats + 8, cyc_wait);
        }
    }
#undef ENVGS_LDS_READ
}
#define ENVGS_COOP_INST(D, W) template __global__ void __launch_bounds__(64 * COOP_W, W) \
    collect_hits_coop<D, W>(const TraceArgs A, const float4 *__restrict__ nodes, const float4 *__restrict__ nodes4, const float4 *__restrict__ srec);
ENVGS_COOP_INST(false, 8)
#ifdef ENVGS_DIAG
ENVGS_COOP_INST(true, 8) ENVGS_COOP_INST(true, 6)
#endif
#undef ENVGS_COOP_INST

}  // namespace envgs
