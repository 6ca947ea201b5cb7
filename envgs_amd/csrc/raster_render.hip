// raster_render.hip -- R6 (front-to-back compositing) and R7 (its back-to-front gradient).
//
// CDNA4 mapping: one 256-lane workgroup per 16x16 tile (the reference's binning granularity, so tile /
// sort indices stay bit-comparable); each of its 4 wavefronts owns one 8x8 pixel quadrant, so a
// wavefront's 64 lanes stay spatially compact: they agree on "this splat misses us" far more often than
// a 16x4 strip would.  Per-tile splat lists are staged through LDS in batches of 256 records with
// coalesced 16 B loads of the 64 B geom record; the inner loop reads each record as an LDS broadcast.
// Cross-pixel sums (per-surfel weight in R6, the 15+C gradient words in R7) are reduced across the
// wavefront with DPP row operations (no LDS, no per-lane atomics) and leave the wavefront as ONE
// global_atomic_add_f32 instruction whose active lanes hit one 128 B gradient record.
//
// Stands behind GaussianRasterizer forward/backward (easyvolcap/utils/gaussian2d_utils.py:1089-1099);
// output channel order: :1119-1144.  Arithmetic: 2DGS ray-splat intersection (Huang et al. 2024), restated
// in oracle/surfel_raster_oracle.c ("parity unpinned" there).
#include "common.h"

namespace envgs {

struct Hit {
    float sx, sy, inv, kx, ky, kz, lx, ly, lz, rho3d, rho2d, dx, dy, depth, G, alpha;
    bool ok;
};

// ---- the CANONICAL per-(pixel, splat) arithmetic ------------------------------------------------------------------------------------
// The ray/splat intersection is ill-conditioned: k = px*Tw - Tu loses ~2 digits at 800 px, p = k x l -> 0 for a splat seen edge-on, and the
// gradient chain (dk = l x dp, dTw = px dk + py dl + ...) mirrors the same cancellations.  Two fp32 evaluations that round these steps
// differently (FMA contraction or not, association) disagree by 1e-5 .. 1e-3 RELATIVE in alpha and in every gradient term ~ 1/p.z -- which
// is what rounds 1-3 measured against the oracle and covered with an uncertainty floor plus a tail allowance.  So the
// operation order of exactly these steps is FIXED, written with explicit fused multiply-adds, and oracle/surfel_raster_oracle.c:eval_splat /
// orc_render_bwd execute the same sequence: the amplified rounding is then common to both and what remains between them is a few ulp of
// well-conditioned arithmetic (the reciprocal, the exponential, the blend recurrences, summation order).
//   k = fma(px, Tw, -Tu)      l = fma(py, Tw, -Tv)      p = k x l with each component fma(a, b, -(c * d))
//   inv = 1 / p.z             s = p.xy * inv            rho3d = fma(sx, sx, sy * sy)     rho2d = 2 * fma(dx, dx, dy * dy)
//   depth = fma(sx, Tw.x, fma(sy, Tw.y, Tw.z))
// 1 / p.z is v_rcp_f32 + one Newton step (2 FMA: within half an ulp of the correctly rounded quotient in all but a vanishing fraction of the
// cases; an IEEE division is ~10 instructions in a VALU-bound kernel); exp is v_exp_f32 on the argument scaled by log2 e.
// EXACT (diagnostic library only, envgs_debug_set(ENVGS_DBG_RASTER_EXACT, 1)): IEEE divisions and the library expf instead -- the attribution run
// of tests/test_raster_parity.py::test_exact_math_attribution.
template <bool EXACT>
__device__ __forceinline__ float recip(float x)
{
    if (EXACT) return 1.0f / x;
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
template <bool EXACT>
__device__ __forceinline__ float recip1(float x) { return EXACT ? 1.0f / x : __builtin_amdgcn_rcpf(x); }     // well-conditioned uses: 1 ulp is enough
template <bool EXACT>
__device__ __forceinline__ float expneg(float x) { return EXACT ? expf(x) : __expf(x); }

// Ray / splat evaluation for pixel (px,py).  rec = the 16-float geom record (wave-uniform).
template <bool EXACT = false>
__device__ __forceinline__ Hit eval_splat(const float4 r0, const float4 r1, const float4 r2, const float4 r3, float px, float py)
{
    Hit h;
    const float Tux = r0.x, Tuy = r0.y, Tuz = r0.z, Tvx = r0.w, Tvy = r1.x, Tvz = r1.y, Twx = r1.z, Twy = r1.w, Twz = r2.x;
    const float cx = r2.y, cy = r2.z, opa = r3.z;
    h.kx = __builtin_fmaf(px, Twx, -Tux); h.ky = __builtin_fmaf(px, Twy, -Tuy); h.kz = __builtin_fmaf(px, Twz, -Tuz);
    h.lx = __builtin_fmaf(py, Twx, -Tvx); h.ly = __builtin_fmaf(py, Twy, -Tvy); h.lz = __builtin_fmaf(py, Twz, -Tvz);
    const float ppx = __builtin_fmaf(h.ky, h.lz, -(h.kz * h.ly));
    const float ppy = __builtin_fmaf(h.kz, h.lx, -(h.kx * h.lz));
    const float ppz = __builtin_fmaf(h.kx, h.ly, -(h.ky * h.lx));
    h.inv = recip<EXACT>(ppz);
    h.sx = ppx * h.inv; h.sy = ppy * h.inv;
    h.rho3d = __builtin_fmaf(h.sx, h.sx, h.sy * h.sy);
    h.dx = cx - px; h.dy = cy - py;
    h.rho2d = FILTER_INV_SQ * __builtin_fmaf(h.dx, h.dx, h.dy * h.dy);
    const bool use3d = h.rho3d <= h.rho2d;
    const float rho = use3d ? h.rho3d : h.rho2d;
    h.depth = use3d ? __builtin_fmaf(h.sx, Twx, __builtin_fmaf(h.sy, Twy, Twz)) : Twz;
    const float power = -0.5f * rho;
    h.G = expneg<EXACT>(power);
    const float a = opa * h.G;
    h.alpha = a < ALPHA_CAP ? a : ALPHA_CAP;
    h.ok = (ppz != 0.0f) && (h.depth >= NEAR_N) && (power <= 0.0f) && (h.alpha >= ALPHA_MIN);
    return h;
}

template <int C, int B = 256>
struct TileLds {
    float4 rec[4][B];
    float col[C][B];
    uint32_t id[B];
    // qbits[q][w]: bit l set = staged splat 64 w + l can reach alpha >= 1/255 somewhere in quadrant q.  Wavefront q walks the SET BITS of its
    // own row (scalar ctz / clear-lowest), so a splat that cannot touch its quadrant costs it nothing -- 71 % of the tile instances of the
    // 300 k / 800x800 view are blended by no pixel of the tile at all, and per-splat "does this one concern me" checks (an LDS read, a wait and
    // a branch each) were 40 % of the forward's scalar instruction stream.
    unsigned long long qbits[4][(B + 63) / 64];
};

// R7 stages BWD_BATCH splats per round: 192 x (64 B record + 4C B colours + 8 B + (15 + C) x 4 B accumulator) = 33 KB at C = 5, so FOUR tiles
// are resident per CU (160 KB LDS) instead of three with 256-splat batches (44 KB).
constexpr int BWD_BATCH = 192;

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (b % 8), each with a private L2.  Give every
// XCD one contiguous run of tiles so neighbouring tiles -- which share most of their splats -- hit the same L2.
__device__ __forceinline__ int xcd_tile(int b, int ntiles)
{
    const int per = (ntiles + 7) >> 3;
    return (b & 7) * per + (b >> 3);
}

// Exact wavefront-level culling.  A pixel composites a splat only if alpha = min(0.99, o*exp(-rho/2)) >= 1/255, i.e.
// rho = min(rho3d, rho2d) <= tau = 2 ln(255 o).  {rho2d <= tau} is a disc of radius sqrt(tau/2) px around the centre;
// {rho3d <= tau} is the projected ellipse of the uv-disc of radius sqrt(tau), whose screen AABB has the closed form of
// the 3-sigma AABB with 9 replaced by tau (valid while the disc stays in front of the w = 0 plane, d < 0; otherwise no
// culling).  A quadrant whose pixel rect misses the union of both boxes (plus a margin far above fp noise) cannot
// contribute, so skipping it changes no result.
__device__ __forceinline__ uint32_t quadrant_mask(const float4 r0, const float4 r1, const float4 r2, const float4 r3, int tile_px, int tile_py)
{
    const float Tux = r0.x, Tuy = r0.y, Tuz = r0.z, Tvx = r0.w, Tvy = r1.x, Tvz = r1.y, Twx = r1.z, Twy = r1.w, Twz = r2.x;
    const float cx = r2.y, cy = r2.z, opa = r3.z;
    const float tau = 2.0f * __logf(255.0f * opa);
    if (!(tau >= 0.0f)) return 0u;
    const float r2d = sqrtf(0.5f * tau);
    float bx0 = cx - r2d, bx1 = cx + r2d, by0 = cy - r2d, by1 = cy + r2d;
    const float d = tau * (Twx * Twx + Twy * Twy) - Twz * Twz;
    if (!(d < 0.0f)) return 0xFu;
    const float id = 1.0f / d;
    const float f0 = tau * id, f2 = -id;
    const float pxc = f0 * (Tux * Twx + Tuy * Twy) + f2 * (Tuz * Twz);
    const float pyc = f0 * (Tvx * Twx + Tvy * Twy) + f2 * (Tvz * Twz);
    const float hx = pxc * pxc - (f0 * (Tux * Tux + Tuy * Tuy) + f2 * (Tuz * Tuz));
    const float hy = pyc * pyc - (f0 * (Tvx * Tvx + Tvy * Tvy) + f2 * (Tvz * Tvz));
    const float ex = sqrtf(fmaxf(hx, 0.0f)), ey = sqrtf(fmaxf(hy, 0.0f));
    bx0 = fminf(bx0, pxc - ex); bx1 = fmaxf(bx1, pxc + ex); by0 = fminf(by0, pyc - ey); by1 = fmaxf(by1, pyc + ey);
    const float mx = 0.05f + 1e-3f * (bx1 - bx0), my = 0.05f + 1e-3f * (by1 - by0);
    bx0 -= mx; bx1 += mx; by0 -= my; by1 += my;
    if (!(bx0 <= bx1 && by0 <= by1)) return 0xFu;       // NaN guard: never cull on garbage
    // Beyond the box: the ellipse itself.  pp = k x l is LINEAR in the pixel, pp = x a + y b + c with a = Tv x Tw, b = Tw x Tu, so
    // {rho3d <= tau} = {pp.x^2 + pp.y^2 - tau pp.z^2 <= 0} is a conic whose quadratic part (A00, A01, A11) does not depend on where the origin
    // is put; with the origin at its centre (pxc, pyc) it reads  A00 x^2 + 2 A01 x y + A11 y^2 <= ex^2 det / A11  (ex = the half extent
    // found above).  A long thin splat seen at an angle fills a small fraction of its box: most of the quadrants the box touches -- in most
    // of the tiles the (square) binning radius assigned it to -- never see it.  The quadrant's pixel rect misses the ellipse iff the
    // minimum of the form over the rect (the origin if inside, else on one of the four edges) exceeds the bound; 5 % and 0.05 px of margin.
    const float ax = Tvy * Twz - Tvz * Twy, ay = Tvz * Twx - Tvx * Twz, az = Tvx * Twy - Tvy * Twx;
    const float bxx = Twy * Tuz - Twz * Tuy, bxy = Twz * Tux - Twx * Tuz, bxz = Twx * Tuy - Twy * Tux;
    const float A00 = ax * ax + ay * ay - tau * az * az, A11 = bxx * bxx + bxy * bxy - tau * bxz * bxz, A01 = ax * bxx + ay * bxy - tau * az * bxz;
    const float det = A00 * A11 - A01 * A01;
    // (the same bound follows from either extent, hx A00 = hy A11: a conic whose two closed forms disagree is not trusted, nor is anything NaN)
    const bool conic = A00 > 0.0f && A11 > 0.0f && det > 1e-6f * A00 * A11 && hx > 0.0f && hy > 0.0f &&
                       fabsf(hx * A00 - hy * A11) <= 0.02f * (hx * A00 + hy * A11);
    const float bound = conic ? 1.05f * hx * det / A11 : 0.0f;
    const float iA00 = conic ? 1.0f / A00 : 0.0f, iA11 = conic ? 1.0f / A11 : 0.0f;
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float qx = (float)(tile_px + (q & 1) * 8), qy = (float)(tile_py + (q >> 1) * 8);
        if (!(bx0 <= qx + 7.0f && bx1 >= qx && by0 <= qy + 7.0f && by1 >= qy)) continue;
        bool hit = true;
        if (conic) {
            // the low-pass disc of radius r2d around (cx, cy)
            const float ddx = fmaxf(fmaxf(qx - cx, cx - (qx + 7.0f)), 0.0f), ddy = fmaxf(fmaxf(qy - cy, cy - (qy + 7.0f)), 0.0f);
            const bool disc = ddx * ddx + ddy * ddy <= (r2d + 0.05f) * (r2d + 0.05f);
            // the ellipse, in coordinates relative to its centre
            const float x0 = qx - 0.05f - pxc, x1 = qx + 7.05f - pxc, y0 = qy - 0.05f - pyc, y1 = qy + 7.05f - pyc;
            bool ell = x0 <= 0.0f && x1 >= 0.0f && y0 <= 0.0f && y1 >= 0.0f;
            if (!ell) {
                float fmin = 3.0e38f;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float yy = e ? y1 : y0;                                  // horizontal edges: minimise over x
                    const float xs = fminf(fmaxf(-A01 * yy * iA00, x0), x1);
                    fmin = fminf(fmin, (A00 * xs + 2.0f * A01 * yy) * xs + A11 * yy * yy);
                    const float xx = e ? x1 : x0;                                  // vertical edges: minimise over y
                    const float ys = fminf(fmaxf(-A01 * xx * iA11, y0), y1);
                    fmin = fminf(fmin, (A11 * ys + 2.0f * A01 * xx) * ys + A00 * xx * xx);
                }
                ell = fmin <= bound;
            }
            hit = disc || ell || !(bound == bound);
        }
        if (hit) m |= 1u << q;
    }
    return m;
}

// ------------------------------------------------------------------------------------------ R6 ---
// AUDIT = true is the parity-audit instantiation of the SAME kernel (envgs_raster_render_audit): it additionally records, per pixel and
// list entry, whether the entry was blended (contrib[pid][entry] = 1), so that tests can compare contributor SETS with the oracle.
template <int C, bool AUDIT, bool EXACT = false>
__global__ void __launch_bounds__(256, 5)
composite_fwd(int W, int H, int bg_len, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ point_list,
              const float *__restrict__ geom, const Feat colors, const float *__restrict__ bg,
              float *__restrict__ out_color, float *__restrict__ allmap, float *__restrict__ final_T,
              int32_t *__restrict__ n_contrib, float *__restrict__ weight, uint8_t *__restrict__ audit_contrib, int audit_lmax,
              uint8_t *__restrict__ contrib_mask, const uint8_t *__restrict__ audit_skip)
{
    __shared__ TileLds<C> lds;
    __shared__ float wacc[256];            // per-splat weight summed over the 4 wavefronts before it leaves the CU
    __shared__ unsigned char cmk[256][4];  // per splat and quadrant: some pixel of the quadrant blended it (-> contrib_mask, what the backward walks);
                                           // each wavefront owns its byte: plain stores
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int tile = xcd_tile(blockIdx.x, gx * gy);
    if (tile >= gx * gy) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pxi = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int pyi = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi, py = (float)pyi;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];

    // AUDIT only: pixels the caller marks (the oracle's fragile pixels) are left out of the per-surfel weight sums, so that `weight` can be
    // compared on EVERY surfel with an oracle weight that leaves out the same pixels
    const bool wskip = AUDIT && audit_skip && inside && audit_skip[(size_t)pyi * W + pxi] != 0;
    bool done = !inside;
    float T = 1.0f, Cacc[C], N0 = 0.f, N1 = 0.f, N2 = 0.f, D = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med = 0.f;
    int32_t last = 0, medc = -1;
#pragma unroll
    for (int c = 0; c < C; c++) Cacc[c] = 0.f;
    wacc[tid] = 0.f; reinterpret_cast<unsigned *>(&cmk[0][0])[tid] = 0u;

    for (uint32_t base = r0; base < r1; base += 256) {
        if (__syncthreads_and(done)) break;
        uint32_t qm = 0u;
        if (base + tid < r1) {
            const uint32_t g = point_list[base + tid];
            const float4 *gp = reinterpret_cast<const float4 *>(geom + (size_t)g * GEOM);
            const float4 a0 = gp[0], a1 = gp[1], a2 = gp[2], a3 = gp[3];
            lds.id[tid] = g;
            lds.rec[0][tid] = a0; lds.rec[1][tid] = a1; lds.rec[2][tid] = a2; lds.rec[3][tid] = a3;
            qm = quadrant_mask(a0, a1, a2, a3, tx * TILE, ty * TILE);
#pragma unroll
            for (int c = 0; c < C; c++) lds.col[c][tid] = colors[(size_t)g * C + c];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64((qm >> q) & 1u);
            if (lane == 0) lds.qbits[q][wave] = m;
        }
        __syncthreads();
        const int count = (int)min(256u, r1 - base);
        for (int ck = 0; ck * 64 < count && __builtin_amdgcn_ballot_w64(!done) != 0; ck++) {
          unsigned long long todo = lds.qbits[wave][ck];
          todo = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(todo >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)todo);
          while (todo != 0ull) {
            const int j = ck * 64 + (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
            const Hit h = eval_splat<EXACT>(lds.rec[0][j], lds.rec[1][j], lds.rec[2][j], lds.rec[3][j], px, py);
            bool contrib = !done && h.ok;
            const float test_T = T * (1.0f - h.alpha);
            const bool stop = contrib && test_T < T_EPS;
            done = done || stop;
            contrib = contrib && !stop;
            float w = 0.f;
            if (contrib) {
                const float4 r2 = lds.rec[2][j], r3 = lds.rec[3][j];
                w = h.alpha * T;
                const float A = 1.0f - T;
                const float m = FAR_N / (FAR_N - NEAR_N) * (1.0f - NEAR_N / h.depth);
                dist += (m * m * A + M2 - 2.0f * m * M1) * w;
                D += h.depth * w;
                M1 += m * w;
                M2 += m * m * w;
                const int32_t contributor = (int32_t)(base - r0) + j + 1;
                if (AUDIT) { if (contributor <= audit_lmax) audit_contrib[((size_t)pyi * W + pxi) * (size_t)audit_lmax + (contributor - 1)] = 1; }
                if (T > 0.5f) { med = h.depth; medc = contributor; }
                N0 += r2.w * w; N1 += r3.x * w; N2 += r3.y * w;
#pragma unroll
                for (int c = 0; c < C; c++) Cacc[c] += lds.col[c][j] * w;
                T = test_T;
                last = contributor;
            }
            if (__builtin_amdgcn_ballot_w64(contrib) != 0) {
                const float ws = wave_sum((AUDIT && wskip) ? 0.f : w);
                if (lane == 0) { atomic_add_f32(&wacc[j], ws); cmk[j][wave] = 1; }         // LDS atomic: ds_add_f32
            }
          }
        }
        __syncthreads();
        if (tid < count) {
            const float ws = wacc[tid];
            if (ws != 0.0f) { atomic_add_f32(weight + lds.id[tid], ws); wacc[tid] = 0.f; }
            if (contrib_mask) {
                unsigned *cw = reinterpret_cast<unsigned *>(&cmk[0][0]) + tid;
                const unsigned v = *cw;                            // 0x01 in byte q = quadrant q
                contrib_mask[base + tid] = (uint8_t)((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u));
                if (v) *cw = 0u;
            }
        }
    }

    if (inside) {
        const size_t HW = (size_t)H * W, pid = (size_t)pyi * W + pxi;
        final_T[pid] = T; final_T[HW + pid] = M1; final_T[2 * HW + pid] = M2;
        n_contrib[pid] = last; n_contrib[HW + pid] = medc;
#pragma unroll
        for (int c = 0; c < C; c++) out_color[c * HW + pid] = Cacc[c] + T * (c < bg_len ? bg[c] : 0.0f);
        allmap[0 * HW + pid] = D;
        allmap[1 * HW + pid] = 1.0f - T;
        allmap[2 * HW + pid] = N0; allmap[3 * HW + pid] = N1; allmap[4 * HW + pid] = N2;
        allmap[5 * HW + pid] = med;
        allmap[6 * HW + pid] = dist;
    }
}

// ------------------------------------------------------------------------------------------ R7 ---
// The per-pixel state of the back-to-front walk is ONE recurrence.  Every channel the forward blends (C colours, depth, alpha, 3 normal
// components, and -- when the distortion map has an upstream gradient -- the distortion weight) enters dL/dalpha as
// (value_j - blend of the values behind j) * upstream, and all of these "blends behind j" obey the same recurrence
//     acc_j = alpha_{j+1} * value_{j+1} + (1 - alpha_{j+1}) * acc_{j+1},
// so the dot product with the upstream gradients is taken FIRST (X_j = sum_ch value_j[ch] * dL/dch) and a single scalar accX is carried:
// dL/dalpha_j = T_j * (X_j - accX_j) - T_final / (1 - alpha_j) * <bg, dL/dpix>.   (10 + C recurrences of the textbook form -> 1.)
// 1 / (1 - alpha) and 1 / depth are v_rcp_f32 (1 ulp; well-conditioned uses), 1 / p.z is the forward's Newton-refined reciprocal (eval_splat):
// the kernel is VALU-issue bound, and an IEEE division is ~12 instructions.
// DIST = the distortion map carries an upstream gradient somewhere in this tile (decided per tile at run time; the shipped EnvGS
// configuration trains with lambda_dist = 0, configs/models/envgs.yaml:73, and then skips the m_d terms altogether).
template <int C, bool DIST, bool EXACT>
__device__ __forceinline__ void composite_bwd_tile(TileLds<C, BWD_BATCH> &lds, float (*gacc)[15 + C], const int W, const int H, const int bg_len,
                                                   const uint32_t *__restrict__ point_list, const float *__restrict__ geom,
                                                   const Feat colors, const float *__restrict__ bg,
                                                   const float *__restrict__ final_T, const int32_t *__restrict__ n_contrib,
                                                   const float *__restrict__ dL_dcolor, const float *__restrict__ dL_dallmap,
                                                   float *__restrict__ grad_rec, const uint32_t r0, const int tx, const int ty, const int max_last,
                                                   const uint8_t *__restrict__ contrib_mask)
{
    constexpr int V = 15 + C;          // gradient words per surfel
    constexpr int N4 = (V + 3) / 4;    // registers left after the transpose-reduce (4 words each)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pxi = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int pyi = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi, py = (float)pyi;
    const size_t HW = (size_t)H * W, pid = (size_t)(inside ? pyi : 0) * W + (inside ? pxi : 0);

    const float T_final = inside ? final_T[pid] : 0.f;
    float T = T_final;
    const int32_t last = inside ? n_contrib[pid] : 0;
    const int32_t medc = inside ? n_contrib[HW + pid] : 0;
    float dpix[C];
    float bg_dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) {
        dpix[c] = (inside && dL_dcolor) ? dL_dcolor[c * HW + pid] : 0.f;          // a NULL upstream pointer = that output has no gradient
        bg_dot += (c < bg_len ? bg[c] : 0.0f) * dpix[c];
    }
    const float dL_ddepth = (inside && dL_dallmap) ? dL_dallmap[0 * HW + pid] : 0.f;
    const float dL_daccum = (inside && dL_dallmap) ? dL_dallmap[1 * HW + pid] : 0.f;
    const float dL_dn0 = (inside && dL_dallmap) ? dL_dallmap[2 * HW + pid] : 0.f;
    const float dL_dn1 = (inside && dL_dallmap) ? dL_dallmap[3 * HW + pid] : 0.f;
    const float dL_dn2 = (inside && dL_dallmap) ? dL_dallmap[4 * HW + pid] : 0.f;
    const float dL_dmed = (inside && dL_dallmap) ? dL_dallmap[5 * HW + pid] : 0.f;
    const float dL_dreg = (DIST && inside && dL_dallmap) ? dL_dallmap[6 * HW + pid] : 0.f;
    const float final_D = (DIST && inside) ? final_T[HW + pid] : 0.f;
    const float final_D2 = (DIST && inside) ? final_T[2 * HW + pid] : 0.f;
    const float final_A = 1.0f - T_final;
    const float bgT = -T_final * bg_dot;
    float last_alpha = 0.f, lastX = 0.f, accX = 0.f;
    int wmax_last = last;
    wmax_last = max(wmax_last, __shfl_xor(wmax_last, 1)); wmax_last = max(wmax_last, __shfl_xor(wmax_last, 2)); wmax_last = max(wmax_last, __shfl_xor(wmax_last, 4));
    wmax_last = max(wmax_last, __shfl_xor(wmax_last, 8)); wmax_last = max(wmax_last, __shfl_xor(wmax_last, 16)); wmax_last = max(wmax_last, __shfl_xor(wmax_last, 32));
    wmax_last = __builtin_amdgcn_readfirstlane(wmax_last);
    constexpr float MD_A = FAR_N / (FAR_N - NEAR_N), MD_B = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);

    for (int top = max_last; top > 0; top -= BWD_BATCH) {
        // stage entries [top-BWD_BATCH, top) in reverse: LDS slot t holds list index top-1-t
        __syncthreads();
        uint32_t qm = 0u;
        if (tid < BWD_BATCH && top - 1 - tid >= 0) {
            const uint32_t g = point_list[r0 + (uint32_t)(top - 1 - tid)];
            const float4 *gp = reinterpret_cast<const float4 *>(geom + (size_t)g * GEOM);
            const float4 a0 = gp[0], a1 = gp[1], a2 = gp[2], a3 = gp[3];
            lds.id[tid] = g;
            lds.rec[0][tid] = a0; lds.rec[1][tid] = a1; lds.rec[2][tid] = a2; lds.rec[3][tid] = a3;
            // which quadrants blended this entry: recorded exactly by the forward (so every pass of the loop below has an active lane), or,
            // without that record, the conservative geometric test the forward itself culls with
            qm = contrib_mask ? (uint32_t)contrib_mask[r0 + (uint32_t)(top - 1 - tid)] : quadrant_mask(a0, a1, a2, a3, tx * TILE, ty * TILE);
#pragma unroll
            for (int c = 0; c < C; c++) lds.col[c][tid] = colors[(size_t)g * C + c];
        }
        if (wave * 64 < BWD_BATCH) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64((qm >> q) & 1u);
                if (lane == 0) lds.qbits[q][wave] = m;
            }
        }
        __syncthreads();
        const int count = min(BWD_BATCH, top);
        // entries at or behind this wavefront's deepest last-contributor concern none of its pixels: slots j < top - wmax (deepest first)
        const int jmin = max(0, top - wmax_last);
        for (int ck = jmin >> 6; ck * 64 < count; ck++) {
          unsigned long long todo = lds.qbits[wave][ck];
          todo = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(todo >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)todo);
          if (ck == (jmin >> 6)) todo &= ~0ull << (jmin & 63);
          while (todo != 0ull) {
            const int j = ck * 64 + (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int ci = top - 1 - j;                       // 0-based position in the tile list
            const bool cand = ci < last;
            if (__builtin_amdgcn_ballot_w64(cand) == 0) continue;
            const float4 q0 = lds.rec[0][j], q1 = lds.rec[1][j], q2 = lds.rec[2][j], q3 = lds.rec[3][j];
            const Hit h = eval_splat<EXACT>(q0, q1, q2, q3, px, py);
            const bool act = cand && h.ok;
            if (__builtin_amdgcn_ballot_w64(act) == 0) continue;

            float gv[4 * N4];
#pragma unroll
            for (int v = 0; v < 4 * N4; v++) gv[v] = 0.f;
            if (act) {
                const float Twx = q1.z, Twy = q1.w, opa = q3.z;
                const float nrm0 = q2.w, nrm1 = q3.x, nrm2 = q3.y;
                const float alpha = h.alpha, G = h.G;
                const float r1 = recip1<EXACT>(1.0f - alpha);
                T = T * r1;                                    // transmittance in front of this splat
                const float w = alpha * T;
                // X_j: this splat's blended values dotted with their upstream gradients
                float X = h.depth * dL_ddepth + dL_daccum;
                X += nrm0 * dL_dn0; X += nrm1 * dL_dn1; X += nrm2 * dL_dn2;
#pragma unroll
                for (int c = 0; c < C; c++) {
                    X += lds.col[c][j] * dpix[c];
                    gv[15 + c] = w * dpix[c];
                }
                float dL_dz = w * dL_ddepth;
                if (ci == medc - 1) dL_dz += dL_dmed;
                if (DIST) {
                    const float idep = recip1<EXACT>(h.depth);
                    const float m_d = MD_A * (1.0f - NEAR_N * idep);
                    X += (final_D2 + m_d * m_d * final_A - 2.0f * m_d * final_D) * dL_dreg;           // the distortion weight joins the recurrence
                    dL_dz += 2.0f * w * (m_d * final_A - final_D) * dL_dreg * (MD_B * idep * idep);
                }
                accX = last_alpha * lastX + (1.f - last_alpha) * accX;
                lastX = X;
                last_alpha = alpha;
                const float dL_dalpha = (X - accX) * T + bgT * r1;
                gv[9] = w * dL_dn0; gv[10] = w * dL_dn1; gv[11] = w * dL_dn2;
                gv[12] = G * dL_dalpha;
                const float nGd = -(opa * dL_dalpha) * G;        // dL/dG * dG/d(rho/2-ish): -G * dL/dG
                if (h.rho3d <= h.rho2d) {
                    // the canonical gradient chain of the intersection (same operation sequence as oracle/surfel_raster_oracle.c:orc_render_bwd)
                    const float dsx = __builtin_fmaf(nGd, h.sx, dL_dz * Twx);
                    const float dsy = __builtin_fmaf(nGd, h.sy, dL_dz * Twy);
                    const float dpx = dsx * h.inv, dpy = dsy * h.inv;
                    const float dpz = -__builtin_fmaf(dpx, h.sx, dpy * h.sy);
                    const float dkx = __builtin_fmaf(h.ly, dpz, -(h.lz * dpy)), dky = __builtin_fmaf(h.lz, dpx, -(h.lx * dpz)), dkz = __builtin_fmaf(h.lx, dpy, -(h.ly * dpx));
                    const float dlx = __builtin_fmaf(dpy, h.kz, -(dpz * h.ky)), dly = __builtin_fmaf(dpz, h.kx, -(dpx * h.kz)), dlz = __builtin_fmaf(dpx, h.ky, -(dpy * h.kx));
                    gv[0] = -dkx; gv[1] = -dky; gv[2] = -dkz;
                    gv[3] = -dlx; gv[4] = -dly; gv[5] = -dlz;
                    gv[6] = __builtin_fmaf(px, dkx, __builtin_fmaf(py, dlx, dL_dz * h.sx));
                    gv[7] = __builtin_fmaf(px, dky, __builtin_fmaf(py, dly, dL_dz * h.sy));
                    gv[8] = __builtin_fmaf(px, dkz, __builtin_fmaf(py, dlz, dL_dz));
                } else {
                    gv[13] = nGd * (FILTER_INV_SQ * h.dx);
                    gv[14] = nGd * (FILTER_INV_SQ * h.dy);
                    gv[8] = dL_dz;
                }
            }
            // wavefront transpose-reduce (permlane swaps + DPP): lane r*16+k ends up owning the sum of word k + r*N4, then one LDS atomic instruction
            const float mine = wave_transpose_reduce<N4>(gv, lane);
            const int vi = (lane & 15) + (lane >> 4) * N4;
            if ((lane & 15) < N4 && vi < V) atomic_add_f32(&gacc[j][vi], mine);
          }
        }
        __syncthreads();
        // flush the batch: consecutive lanes own consecutive words, so one instruction covers ~3 whole 128 B records
        for (int i = tid; i < count * V; i += 256) {
            const int j = i / V, v = i - j * V;
            const float val = gacc[j][v];
            if (val != 0.0f) { atomic_add_f32(grad_rec + (size_t)lds.id[j] * GREC + v, val); gacc[j][v] = 0.f; }
        }
    }
}

template <int C, bool EXACT = false>
__global__ void __launch_bounds__(256, 4)
composite_bwd(int W, int H, int bg_len, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ point_list,
              const float *__restrict__ geom, const Feat colors, const float *__restrict__ bg,
              const float *__restrict__ final_T, const int32_t *__restrict__ n_contrib,
              const float *__restrict__ dL_dcolor, const float *__restrict__ dL_dallmap, float *__restrict__ grad_rec,
              const uint8_t *__restrict__ contrib_mask)
{
    constexpr int V = 15 + C;
    __shared__ TileLds<C, BWD_BATCH> lds;
    // Per-batch gradient accumulator: the 4 wavefronts add their DPP-reduced words here (ds_add_f32), and the tile sends
    // ONE global atomic per (splat, word).  The L2 atomic units retire roughly one dword per clock per channel
    // (~0.12 T dword-atomics/s measured), so the number of global atomic dwords -- not bytes -- is what R7 pays for.
    __shared__ float gacc[BWD_BATCH][V];
    __shared__ int s_max_last, s_dist;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int tile = xcd_tile(blockIdx.x, gx * gy);
    if (tile >= gx * gy) return;
    const int tx = tile % gx, ty = tile / gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int pxi = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int pyi = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const size_t HW = (size_t)H * W, pid = (size_t)(inside ? pyi : 0) * W + (inside ? pxi : 0);
    const int32_t last = inside ? n_contrib[pid] : 0;
    const bool reg = inside && dL_dallmap && dL_dallmap[6 * HW + pid] != 0.0f;

    // Entries behind the deepest last-contributor of this tile were never blended by any pixel: skip them.
    if (tid == 0) { s_max_last = 0; s_dist = 0; }
    for (int i = tid; i < BWD_BATCH * V; i += 256) (&gacc[0][0])[i] = 0.f;
    __syncthreads();
    {
        int m = last;
        m = max(m, __shfl_xor(m, 1)); m = max(m, __shfl_xor(m, 2)); m = max(m, __shfl_xor(m, 4));
        m = max(m, __shfl_xor(m, 8)); m = max(m, __shfl_xor(m, 16)); m = max(m, __shfl_xor(m, 32));
        if (lane == 0) atomicMax(&s_max_last, m);
        // the ballot is taken by the WHOLE wavefront (inside `lane == 0 && ...` only lane 0 would be active and the mask would reflect one
        // pixel per quadrant: a distortion gradient that is zero at the four quadrant origins but not elsewhere would be dropped)
        const bool any_reg = __builtin_amdgcn_ballot_w64(reg) != 0;
        if (lane == 0 && any_reg) s_dist = 1;
    }
    __syncthreads();
    const int max_last = s_max_last;
    const uint32_t r0 = ranges[2 * tile];
    if (s_dist)
        composite_bwd_tile<C, true, EXACT>(lds, gacc, W, H, bg_len, point_list, geom, colors, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, r0, tx, ty, max_last, contrib_mask);
    else
        composite_bwd_tile<C, false, EXACT>(lds, gacc, W, H, bg_len, point_list, geom, colors, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, r0, tx, ty, max_last, contrib_mask);
}

// ------------------------------------------------------------------------------------ launchers ---
template <int C>
static int run_fwd(const envgs_raster_cfg *cfg, const uint32_t *ranges, const uint32_t *point_list, const float *geom,
                   const float *colors, const float *bg, float *out_color, float *allmap, float *final_T,
                   int32_t *n_contrib, float *weight, uint8_t *audit_contrib, int audit_lmax, hipStream_t stream, int colors_f16, uint8_t *contrib_mask,
                   const uint8_t *audit_skip)
{
    const int gx = (cfg->width + TILE - 1) / TILE, gy = (cfg->height + TILE - 1) / TILE;
    ProfScope prof_(K_COMPOSITE_FWD, stream);
    const dim3 grid(8 * ((gx * gy + 7) / 8)), block(256);
    const Feat colors_{colors, colors_f16 != 0};
#ifdef ENVGS_DIAG
    if (debug_switch(ENVGS_DBG_RASTER_EXACT)) {       // diagnostic library only: IEEE divisions + library expf (the attribution run)
        if (audit_contrib)
            hipLaunchKernelGGL((composite_fwd<C, true, true>), grid, block, 0, stream, cfg->width, cfg->height, cfg->bg_len, ranges,
                               point_list, geom, colors_, bg, out_color, allmap, final_T, n_contrib, weight, audit_contrib, audit_lmax, contrib_mask, audit_skip);
        else
            hipLaunchKernelGGL((composite_fwd<C, false, true>), grid, block, 0, stream, cfg->width, cfg->height, cfg->bg_len, ranges,
                               point_list, geom, colors_, bg, out_color, allmap, final_T, n_contrib, weight, (uint8_t *)nullptr, 0, contrib_mask, (const uint8_t *)nullptr);
        ENVGS_CHECK_LAUNCH(cfg, stream);
        return 0;
    }
#endif
    if (audit_contrib)
        hipLaunchKernelGGL((composite_fwd<C, true>), grid, block, 0, stream, cfg->width, cfg->height, cfg->bg_len, ranges,
                           point_list, geom, colors_, bg, out_color, allmap, final_T, n_contrib, weight, audit_contrib, audit_lmax, contrib_mask, audit_skip);
    else
        hipLaunchKernelGGL((composite_fwd<C, false>), grid, block, 0, stream, cfg->width, cfg->height, cfg->bg_len, ranges,
                           point_list, geom, colors_, bg, out_color, allmap, final_T, n_contrib, weight, (uint8_t *)nullptr, 0, contrib_mask, (const uint8_t *)nullptr);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return 0;
}

int launch_render_fwd(const envgs_raster_cfg *cfg, const uint32_t *ranges, const uint32_t *point_list, const float *geom,
                      const float *colors, const float *bg, float *out_color, float *allmap, float *final_T,
                      int32_t *n_contrib, float *weight, hipStream_t stream, uint8_t *audit_contrib, int audit_lmax, int colors_f16, uint8_t *contrib_mask,
                      const uint8_t *audit_skip)
{
    if (cfg->P > 0) {
        hipError_t e = hipMemsetAsync(weight, 0, sizeof(float) * (size_t)cfg->P, stream);
        if (e != hipSuccess) return (int)e;
    }
    switch (cfg->channels) {
    case 3: return run_fwd<3>(cfg, ranges, point_list, geom, colors, bg, out_color, allmap, final_T, n_contrib, weight, audit_contrib, audit_lmax, stream, colors_f16, contrib_mask, audit_skip);
    case 5: return run_fwd<5>(cfg, ranges, point_list, geom, colors, bg, out_color, allmap, final_T, n_contrib, weight, audit_contrib, audit_lmax, stream, colors_f16, contrib_mask, audit_skip);
    case 7: return run_fwd<7>(cfg, ranges, point_list, geom, colors, bg, out_color, allmap, final_T, n_contrib, weight, audit_contrib, audit_lmax, stream, colors_f16, contrib_mask, audit_skip);
    default: return ENVGS_ERR_BAD_ARG;
    }
}

template <int C>
static int run_bwd(const envgs_raster_cfg *cfg, const uint32_t *ranges, const uint32_t *point_list, const float *geom,
                   const float *colors, const float *bg, const float *final_T, const int32_t *n_contrib,
                   const float *dL_dcolor, const float *dL_dallmap, float *grad_rec, hipStream_t stream, int colors_f16, const uint8_t *contrib_mask)
{
    const int gx = (cfg->width + TILE - 1) / TILE, gy = (cfg->height + TILE - 1) / TILE;
    ProfScope prof_(K_COMPOSITE_BWD, stream);
#ifdef ENVGS_DIAG
    if (debug_switch(ENVGS_DBG_RASTER_EXACT)) {
        hipLaunchKernelGGL((composite_bwd<C, true>), dim3(8 * ((gx * gy + 7) / 8)), dim3(256), 0, stream, cfg->width, cfg->height, cfg->bg_len, ranges,
                           point_list, geom, Feat{colors, colors_f16 != 0}, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, contrib_mask);
        ENVGS_CHECK_LAUNCH(cfg, stream);
        return 0;
    }
#endif
    hipLaunchKernelGGL((composite_bwd<C, false>), dim3(8 * ((gx * gy + 7) / 8)), dim3(256), 0, stream, cfg->width, cfg->height, cfg->bg_len, ranges,
                       point_list, geom, Feat{colors, colors_f16 != 0}, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, contrib_mask);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return 0;
}

int launch_render_bwd(const envgs_raster_cfg *cfg, const uint32_t *ranges, const uint32_t *point_list, const float *geom,
                      const float *colors, const float *bg, const float *final_T, const int32_t *n_contrib,
                      const float *dL_dcolor, const float *dL_dallmap, float *grad_rec, hipStream_t stream, int colors_f16, const uint8_t *contrib_mask)
{
    if (cfg->P <= 0) return 0;
    hipError_t e = hipMemsetAsync(grad_rec, 0, sizeof(float) * GREC * (size_t)cfg->P, stream);
    if (e != hipSuccess) return (int)e;
    switch (cfg->channels) {
    case 3: return run_bwd<3>(cfg, ranges, point_list, geom, colors, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, stream, colors_f16, contrib_mask);
    case 5: return run_bwd<5>(cfg, ranges, point_list, geom, colors, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, stream, colors_f16, contrib_mask);
    case 7: return run_bwd<7>(cfg, ranges, point_list, geom, colors, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, stream, colors_f16, contrib_mask);
    default: return ENVGS_ERR_BAD_ARG;
    }
}

}  // namespace envgs
