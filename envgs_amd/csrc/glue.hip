// glue.hip -- fused caller-side glue (include/envgs_glue.h): per-surfel SH -> colour channels and per-pixel reflected-ray
// construction, forward and backward, one HBM pass each (the reference spends ~125 torch launches on the same expressions).
#include "common.h"
#include <cstdint>

#include "../../include/envgs_glue.h"

namespace envgs {

constexpr float gC0 = 0.28209479177387814f;
constexpr float gC1 = 0.4886025119029199f;
__device__ __constant__ float gC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float gC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f};

// The SH basis and its gradient are evaluated WITHOUT FMA contraction, statement by statement as oracle/surfel_raster_oracle.c does: a basis
// function near one of its zeros (2 zz - xx - yy -> 0) is a cancellation whose relative rounding error is unbounded, so two evaluations agree on
// dL/dSH = basis * dL/dcolour to 1e-4 RELATIVE only if they round the same way (round 4: the 16 of 14.4 M dshs elements beyond tolerance at
// full size were exactly these; the rasterizer's forward colours, raster_project.hip, have been bit-exact this way since round 1).
__device__ __forceinline__ void basis16(int D, float x, float y, float z, float *b)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = 0.f;
    b[0] = gC0;
    if (D > 0) {
        b[1] = -gC1 * y; b[2] = gC1 * z; b[3] = -gC1 * x;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = gC2[0] * xy; b[5] = gC2[1] * yz; b[6] = gC2[2] * (2.0f * zz - xx - yy); b[7] = gC2[3] * xz; b[8] = gC2[4] * (xx - yy);
            if (D > 2) {
                b[9] = gC3[0] * y * (3.0f * xx - yy); b[10] = gC3[1] * xy * z; b[11] = gC3[2] * y * (4.0f * zz - xx - yy);
                b[12] = gC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy); b[13] = gC3[4] * x * (4.0f * zz - xx - yy);
                b[14] = gC3[5] * z * (xx - yy); b[15] = gC3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

__device__ __forceinline__ void basis16_grad(int D, float x, float y, float z, float *gx, float *gy, float *gz)
{
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < 16; k++) { gx[k] = 0.f; gy[k] = 0.f; gz[k] = 0.f; }
    if (D > 0) {
        gy[1] = -gC1; gz[2] = gC1; gx[3] = -gC1;
        if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            gx[4] = gC2[0] * y; gy[4] = gC2[0] * x;
            gy[5] = gC2[1] * z; gz[5] = gC2[1] * y;
            gx[6] = gC2[2] * -2.f * x; gy[6] = gC2[2] * -2.f * y; gz[6] = gC2[2] * 4.f * z;
            gx[7] = gC2[3] * z; gz[7] = gC2[3] * x;
            gx[8] = gC2[4] * 2.f * x; gy[8] = gC2[4] * -2.f * y;
            if (D > 2) {
                gx[9] = gC3[0] * 6.f * xy; gy[9] = gC3[0] * 3.f * (xx - yy);
                gx[10] = gC3[1] * yz; gy[10] = gC3[1] * xz; gz[10] = gC3[1] * xy;
                gx[11] = gC3[2] * -2.f * xy; gy[11] = gC3[2] * (4.f * zz - xx - 3.f * yy); gz[11] = gC3[2] * 8.f * yz;
                gx[12] = gC3[3] * -6.f * xz; gy[12] = gC3[3] * -6.f * yz; gz[12] = gC3[3] * 3.f * (2.f * zz - xx - yy);
                gx[13] = gC3[4] * (4.f * zz - 3.f * xx - yy); gy[13] = gC3[4] * -2.f * xy; gz[13] = gC3[4] * 8.f * xz;
                gx[14] = gC3[5] * 2.f * xz; gy[14] = gC3[5] * -2.f * yz; gz[14] = gC3[5] * (xx - yy);
                gx[15] = gC3[6] * 3.f * (xx - yy); gy[15] = gC3[6] * -6.f * xy;
            }
        }
    }
}

__global__ void __launch_bounds__(256)
sh_colors_fwd(int P, int D, int M, int S, const float *__restrict__ means, const float *__restrict__ shs, const float *__restrict__ campos,
              const float *__restrict__ spec, const float *__restrict__ rough, float *__restrict__ colors, uint8_t *__restrict__ clamped)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int C = 3 + S + 1;
    const float dx = means[3 * i] - campos[0], dy = means[3 * i + 1] - campos[1], dz = means[3 * i + 2] - campos[2];
    const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float b[16];
    basis16(D, dx * il, dy * il, dz * il, b);
    const float *sh = shs + (size_t)i * M * 3;
    const int nb = (D + 1) * (D + 1);
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < nb) { r0 += b[k] * sh[k * 3]; r1 += b[k] * sh[k * 3 + 1]; r2 += b[k] * sh[k * 3 + 2]; }
    r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
    clamped[3 * i] = r0 < 0.f; clamped[3 * i + 1] = r1 < 0.f; clamped[3 * i + 2] = r2 < 0.f;
    float *o = colors + (size_t)i * C;
    o[0] = fmaxf(r0, 0.f); o[1] = fmaxf(r1, 0.f); o[2] = fmaxf(r2, 0.f);
    for (int s = 0; s < S; s++) o[3 + s] = spec[(size_t)i * S + s];
    o[3 + S] = rough[i];
}

__global__ void __launch_bounds__(256)
sh_colors_bwd(int P, int D, int M, int S, const float *__restrict__ means, const float *__restrict__ shs, const float *__restrict__ campos,
              const uint8_t *__restrict__ clamped, const float *__restrict__ dcolors, float *__restrict__ dmeans, float *__restrict__ dshs,
              float *__restrict__ dspec, float *__restrict__ drough)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const int C = 3 + S + 1;
    const float *g = dcolors + (size_t)i * C;
    const float g0 = clamped[3 * i] ? 0.f : g[0], g1 = clamped[3 * i + 1] ? 0.f : g[1], g2 = clamped[3 * i + 2] ? 0.f : g[2];
    const float dx = means[3 * i] - campos[0], dy = means[3 * i + 1] - campos[1], dz = means[3 * i + 2] - campos[2];
    const float sum2 = dx * dx + dy * dy + dz * dz, il = 1.0f / sqrtf(sum2);
    const float x = dx * il, y = dy * il, z = dz * il;
    float b[16], gx[16], gy[16], gz[16];
    basis16(D, x, y, z, b);
    basis16_grad(D, x, y, z, gx, gy, gz);
    const float *sh = shs + (size_t)i * M * 3;
    float *dsh = dshs + (size_t)i * M * 3;
    const int nb = (D + 1) * (D + 1);
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
    for (int k = 0; k < M; k++) {
        if (k < nb && k < 16) {
            dsh[k * 3] = b[k] * g0; dsh[k * 3 + 1] = b[k] * g1; dsh[k * 3 + 2] = b[k] * g2;
            const float sd = sh[k * 3] * g0 + sh[k * 3 + 1] * g1 + sh[k * 3 + 2] * g2;
            ddx += gx[k] * sd; ddy += gy[k] * sd; ddz += gz[k] * sd;
        } else { dsh[k * 3] = 0.f; dsh[k * 3 + 1] = 0.f; dsh[k * 3 + 2] = 0.f; }
    }
    const float inv3 = il * il * il;
    dmeans[3 * i] = ((sum2 - dx * dx) * ddx - dy * dx * ddy - dz * dx * ddz) * inv3;
    dmeans[3 * i + 1] = (-dx * dy * ddx + (sum2 - dy * dy) * ddy - dz * dy * ddz) * inv3;
    dmeans[3 * i + 2] = (-dx * dz * ddx - dy * dz * ddy + (sum2 - dz * dz) * ddz) * inv3;
    for (int s = 0; s < S; s++) dspec[(size_t)i * S + s] = g[3 + s];
    drough[i] = g[3 + S];
}

// The same two kernels for the usual 16-coefficient layout, FOUR LANES PER SURFEL: with one lane per surfel every 4 B access of the 192 B
// SH / gradient block touches 64 different cache lines per instruction (48 such loads and 48 such stores per lane: 0.9 TB/s measured); here
// lane q of a quad owns coefficients 4q .. 4q+3 = 48 contiguous bytes (three 16 B accesses), the quad covers the block, and the per-surfel sums
// are reduced inside the quad with DPP.
template <int CTRL> __device__ __forceinline__ float quad_xchg(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float quad_sum(float v) { v += quad_xchg<0xB1>(v); v += quad_xchg<0x4E>(v); return v; }

__global__ void __launch_bounds__(256)
sh_colors_fwd_q16(int P, int D, int S, const float *__restrict__ means, const float *__restrict__ shs, const float *__restrict__ campos,
                  const float *__restrict__ spec, const float *__restrict__ rough, float *__restrict__ colors, uint8_t *__restrict__ clamped)
{
    const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, q = t & 3;
    const bool live = i < P;
    const int ii = live ? i : 0;
    const int C = 3 + S + 1;
    const float dx = means[3 * ii] - campos[0], dy = means[3 * ii + 1] - campos[1], dz = means[3 * ii + 2] - campos[2];
    const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    float b[16];
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = 0.f;
    basis16(D, dx * il, dy * il, dz * il, b);
    const int nb = (D + 1) * (D + 1);
    float bq[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const float v = q == 0 ? b[m] : q == 1 ? b[4 + m] : q == 2 ? b[8 + m] : b[12 + m];
        bq[m] = (4 * q + m) < nb ? v : 0.f;
    }
    const float4 *sh4 = reinterpret_cast<const float4 *>(shs + (size_t)ii * 48) + 3 * q;
    const float4 x0 = sh4[0], x1 = sh4[1], x2 = sh4[2];
    const float x[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
    float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 12; f++) r[f % 3] += bq[f / 3] * x[f];
    const float r0 = quad_sum(r[0]) + 0.5f, r1 = quad_sum(r[1]) + 0.5f, r2 = quad_sum(r[2]) + 0.5f;
    if (live && q == 0) {
        clamped[3 * i] = r0 < 0.f; clamped[3 * i + 1] = r1 < 0.f; clamped[3 * i + 2] = r2 < 0.f;
        float *o = colors + (size_t)i * C;
        o[0] = fmaxf(r0, 0.f); o[1] = fmaxf(r1, 0.f); o[2] = fmaxf(r2, 0.f);
        for (int s2 = 0; s2 < S; s2++) o[3 + s2] = spec[(size_t)i * S + s2];
        o[3 + S] = rough[i];
    }
}

__global__ void __launch_bounds__(256)
sh_colors_bwd_q16(int P, int D, int S, const float *__restrict__ means, const float *__restrict__ shs, const float *__restrict__ campos,
                  const uint8_t *__restrict__ clamped, const float *__restrict__ dcolors, float *__restrict__ dmeans, float *__restrict__ dshs,
                  float *__restrict__ dspec, float *__restrict__ drough)
{
    const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, q = t & 3;
    const bool live = i < P;
    const int ii = live ? i : 0;
    const int C = 3 + S + 1;
    const float *g = dcolors + (size_t)ii * C;
    const float g0 = clamped[3 * ii] ? 0.f : g[0], g1 = clamped[3 * ii + 1] ? 0.f : g[1], g2 = clamped[3 * ii + 2] ? 0.f : g[2];
    const float gc[3] = {g0, g1, g2};
    const float dx = means[3 * ii] - campos[0], dy = means[3 * ii + 1] - campos[1], dz = means[3 * ii + 2] - campos[2];
    const float sum2 = dx * dx + dy * dy + dz * dz, il = 1.0f / sqrtf(sum2);
    const float x = dx * il, y = dy * il, z = dz * il;
    float b[16], gx[16], gy[16], gz[16];
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = 0.f;
    basis16(D, x, y, z, b);
    basis16_grad(D, x, y, z, gx, gy, gz);
    const int nb = (D + 1) * (D + 1);
    float bq[4], gxq[4], gyq[4], gzq[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const bool in = (4 * q + m) < nb;
        bq[m] = in ? (q == 0 ? b[m] : q == 1 ? b[4 + m] : q == 2 ? b[8 + m] : b[12 + m]) : 0.f;
        gxq[m] = in ? (q == 0 ? gx[m] : q == 1 ? gx[4 + m] : q == 2 ? gx[8 + m] : gx[12 + m]) : 0.f;
        gyq[m] = in ? (q == 0 ? gy[m] : q == 1 ? gy[4 + m] : q == 2 ? gy[8 + m] : gy[12 + m]) : 0.f;
        gzq[m] = in ? (q == 0 ? gz[m] : q == 1 ? gz[4 + m] : q == 2 ? gz[8 + m] : gz[12 + m]) : 0.f;
    }
    const float4 *sh4 = reinterpret_cast<const float4 *>(shs + (size_t)ii * 48) + 3 * q;
    const float4 x0 = sh4[0], x1 = sh4[1], x2 = sh4[2];
    const float xs[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
    float o[12], sd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 12; f++) { o[f] = bq[f / 3] * gc[f % 3]; sd[f / 3] += xs[f] * gc[f % 3]; }
    if (live) {
        float4 *d4 = reinterpret_cast<float4 *>(dshs + (size_t)i * 48) + 3 * q;
        d4[0] = make_float4(o[0], o[1], o[2], o[3]); d4[1] = make_float4(o[4], o[5], o[6], o[7]); d4[2] = make_float4(o[8], o[9], o[10], o[11]);
    }
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
    for (int m = 0; m < 4; m++) { ddx += gxq[m] * sd[m]; ddy += gyq[m] * sd[m]; ddz += gzq[m] * sd[m]; }
    ddx = quad_sum(ddx); ddy = quad_sum(ddy); ddz = quad_sum(ddz);
    if (live && q == 0) {
        const float inv3 = il * il * il;
        dmeans[3 * i] = ((sum2 - dx * dx) * ddx - dy * dx * ddy - dz * dx * ddz) * inv3;
        dmeans[3 * i + 1] = (-dx * dy * ddx + (sum2 - dy * dy) * ddy - dz * dy * ddz) * inv3;
        dmeans[3 * i + 2] = (-dx * dz * ddx - dy * dz * ddy + (sum2 - dz * dz) * ddz) * inv3;
        for (int s2 = 0; s2 < S; s2++) dspec[(size_t)i * S + s2] = g[3 + s2];
        drough[i] = g[3 + S];
    }
}

// The rasterizer's in-kernel SH colours (csrc/raster_project.hip) differentiated the same way: dL/dcolour comes out of the per-surfel gradient
// record R7 accumulated (words 15..17), surfels that were not rendered get zeros; dshs is written, the view-direction term is ADDED to the
// position gradient project_surfels_bwd has just written.  (One lane per surfel inside project_surfels_bwd: 185 us for 300 k surfels, every
// 4 B access of the 192 B blocks on its own cache line; this kernel + the SH-free project_surfels_bwd: see DESIGN.md section 4.)
__global__ void __launch_bounds__(256)
sh_record_bwd_q16(int P, int D, const float *__restrict__ means, const float *__restrict__ shs, const float *__restrict__ campos,
                  const uint8_t *__restrict__ clamped, const int32_t *__restrict__ radii, const float *__restrict__ grad_rec,
                  float *__restrict__ dmeans, float *__restrict__ dshs)
{
    const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 2, q = t & 3;
    const bool live = i < P;
    const int ii = live ? i : 0;
    const bool vis = radii[ii] > 0;
    const float *g = grad_rec + (size_t)ii * GREC + 15;
    const float gc[3] = {(vis && !clamped[3 * ii]) ? g[0] : 0.f, (vis && !clamped[3 * ii + 1]) ? g[1] : 0.f, (vis && !clamped[3 * ii + 2]) ? g[2] : 0.f};
    const float dx = means[3 * ii] - campos[0], dy = means[3 * ii + 1] - campos[1], dz = means[3 * ii + 2] - campos[2];
    float sum2, len;
    {
#pragma clang fp contract(off)
        sum2 = dx * dx + dy * dy + dz * dz;           // the oracle's statements (orc_preprocess_bwd): uncontracted sum, sqrt, three divisions
        len = sqrtf(sum2);
    }
    // a surfel that was not rendered -- or one sitting exactly on the camera centre (0 / 0) -- gets plain zeros, not basis * 0 (a NaN
    // written into dshs of a culled surfel would poison its Adam moments for good)
    const bool dead = !vis || !(sum2 > 0.0f);
    const float il = dead ? 0.0f : 1.0f / len;
    const float x = dead ? 0.0f : dx / len, y = dead ? 0.0f : dy / len, z = dead ? 0.0f : dz / len;
    float b[16], gx[16], gy[16], gz[16];
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = 0.f;
    basis16(D, x, y, z, b);
    basis16_grad(D, x, y, z, gx, gy, gz);
    const int nb = (D + 1) * (D + 1);
    float bq[4], gxq[4], gyq[4], gzq[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const bool in = (4 * q + m) < nb;
        bq[m] = in ? (q == 0 ? b[m] : q == 1 ? b[4 + m] : q == 2 ? b[8 + m] : b[12 + m]) : 0.f;
        gxq[m] = in ? (q == 0 ? gx[m] : q == 1 ? gx[4 + m] : q == 2 ? gx[8 + m] : gx[12 + m]) : 0.f;
        gyq[m] = in ? (q == 0 ? gy[m] : q == 1 ? gy[4 + m] : q == 2 ? gy[8 + m] : gy[12 + m]) : 0.f;
        gzq[m] = in ? (q == 0 ? gz[m] : q == 1 ? gz[4 + m] : q == 2 ? gz[8 + m] : gz[12 + m]) : 0.f;
    }
    const float4 *sh4 = reinterpret_cast<const float4 *>(shs + (size_t)ii * 48) + 3 * q;
    const float4 x0 = sh4[0], x1 = sh4[1], x2 = sh4[2];
    const float xs[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
    float o[12], sd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 12; f++) { o[f] = bq[f / 3] * gc[f % 3]; sd[f / 3] += xs[f] * gc[f % 3]; }
    if (live) {
        float4 *d4 = reinterpret_cast<float4 *>(dshs + (size_t)i * 48) + 3 * q;
        d4[0] = make_float4(o[0], o[1], o[2], o[3]); d4[1] = make_float4(o[4], o[5], o[6], o[7]); d4[2] = make_float4(o[8], o[9], o[10], o[11]);
    }
    float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
    for (int m = 0; m < 4; m++) { ddx += gxq[m] * sd[m]; ddy += gyq[m] * sd[m]; ddz += gzq[m] * sd[m]; }
    ddx = quad_sum(ddx); ddy = quad_sum(ddy); ddz = quad_sum(ddz);
    if (live && q == 0 && vis) {
        const float inv3 = il * il * il;
        dmeans[3 * i] += ((sum2 - dx * dx) * ddx - dy * dx * ddy - dz * dx * ddz) * inv3;
        dmeans[3 * i + 1] += (-dx * dy * ddx + (sum2 - dy * dy) * ddy - dz * dy * ddz) * inv3;
        dmeans[3 * i + 2] += (-dx * dz * ddx - dy * dz * ddy + (sum2 - dz * dz) * ddz) * inv3;
    }
}

int launch_sh_record_bwd(int P, int D, const float *means3D, const float *shs, const float *campos, const uint8_t *clamped, const int32_t *radii,
                         const float *grad_rec, float *dmeans3D, float *dshs, hipStream_t stream)
{
    if (P <= 0) return 0;
    hipLaunchKernelGGL(sh_record_bwd_q16, dim3((4 * (size_t)P + 255) / 256), dim3(256), 0, stream, P, D, means3D, shs, campos, clamped, radii, grad_rec,
                       dmeans3D, dshs);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ reflect
__global__ void __launch_bounds__(256)
reflect_fwd(int HW, float ratio, const float *__restrict__ allmap, const float *__restrict__ ray_o, const float *__restrict__ ray_d,
            const float *__restrict__ V, float *__restrict__ nw, float *__restrict__ depth, float *__restrict__ ref_o, float *__restrict__ ref_d)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float Dw = allmap[p], A = allmap[HW + p], n0 = allmap[2 * HW + p], n1 = allmap[3 * HW + p], n2 = allmap[4 * HW + p], med = allmap[5 * HW + p];
    // normal_world_c = sum_k n_view_k * V[c][k]   (V = world_view_transform, rows 0..2 = R^T)
    const float w0 = n0 * V[0] + n1 * V[1] + n2 * V[2], w1 = n0 * V[4] + n1 * V[5] + n2 * V[6], w2 = n0 * V[8] + n1 * V[9] + n2 * V[10];
    float de = Dw / A;
    if (!(fabsf(de) <= 3.0e38f)) de = 0.f;                                   // nan_to_num(., 0, 0)
    float dm = med;
    if (!(fabsf(dm) <= 3.0e38f)) dm = 0.f;
    const float dep = de * (1.0f - ratio) + dm * ratio;
    nw[p] = w0; nw[HW + p] = w1; nw[2 * HW + p] = w2;
    depth[p] = dep;
    const float len = sqrtf(w0 * w0 + w1 * w1 + w2 * w2), il = 1.0f / (len + 1e-8f);     // math_utils.normalize: x / (|x| + 1e-8)
    const float u0 = w0 * il, u1 = w1 * il, u2 = w2 * il;
    const float o0 = ray_o[3 * p], o1 = ray_o[3 * p + 1], o2 = ray_o[3 * p + 2];
    const float d0 = ray_d[3 * p], d1 = ray_d[3 * p + 1], d2 = ray_d[3 * p + 2];
    const float dn = d0 * u0 + d1 * u1 + d2 * u2;
    ref_d[3 * p] = d0 - 2.0f * dn * u0; ref_d[3 * p + 1] = d1 - 2.0f * dn * u1; ref_d[3 * p + 2] = d2 - 2.0f * dn * u2;
    ref_o[3 * p] = o0 + d0 * dep; ref_o[3 * p + 1] = o1 + d1 * dep; ref_o[3 * p + 2] = o2 + d2 * dep;
}

__global__ void __launch_bounds__(256)
reflect_bwd(int HW, float ratio, const float *__restrict__ allmap, const float *__restrict__ ray_o, const float *__restrict__ ray_d,
            const float *__restrict__ V, const float *__restrict__ dnw, const float *__restrict__ ddepth, const float *__restrict__ dref_o,
            const float *__restrict__ dref_d, float *__restrict__ dallmap, float *__restrict__ dray_o, float *__restrict__ dray_d)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float Dw = allmap[p], A = allmap[HW + p], n0 = allmap[2 * HW + p], n1 = allmap[3 * HW + p], n2 = allmap[4 * HW + p], med = allmap[5 * HW + p];
    const float w0 = n0 * V[0] + n1 * V[1] + n2 * V[2], w1 = n0 * V[4] + n1 * V[5] + n2 * V[6], w2 = n0 * V[8] + n1 * V[9] + n2 * V[10];
    const float d0 = ray_d[3 * p], d1 = ray_d[3 * p + 1], d2 = ray_d[3 * p + 2];
    const float go0 = dref_o ? dref_o[3 * p] : 0.f, go1 = dref_o ? dref_o[3 * p + 1] : 0.f, go2 = dref_o ? dref_o[3 * p + 2] : 0.f;
    const float gd0 = dref_d ? dref_d[3 * p] : 0.f, gd1 = dref_d ? dref_d[3 * p + 1] : 0.f, gd2 = dref_d ? dref_d[3 * p + 2] : 0.f;
    float de = Dw / A; const bool de_ok = fabsf(de) <= 3.0e38f; if (!de_ok) de = 0.f;
    float dm = med; const bool dm_ok = fabsf(dm) <= 3.0e38f; if (!dm_ok) dm = 0.f;
    const float dep = de * (1.0f - ratio) + dm * ratio;
    // ref_o = o + d*dep
    float gdep = (ddepth ? ddepth[p] : 0.f) + go0 * d0 + go1 * d1 + go2 * d2;
    // ref_d = d - 2 (d.u) u,  u = w / max(|w|, eps)
    const float len = sqrtf(w0 * w0 + w1 * w1 + w2 * w2), il = 1.0f / (len + 1e-8f);
    const float u0 = w0 * il, u1 = w1 * il, u2 = w2 * il;
    const float dn = d0 * u0 + d1 * u1 + d2 * u2, gu_dot = gd0 * u0 + gd1 * u1 + gd2 * u2;
    // dL/du = -2 [ (g.u) d + (d.u) g ]
    const float gu0 = -2.0f * (gu_dot * d0 + dn * gd0), gu1 = -2.0f * (gu_dot * d1 + dn * gd1), gu2 = -2.0f * (gu_dot * d2 + dn * gd2);
    // u = w / (|w| + eps):  dL/dw = gu/(|w|+eps) - (gu.w) w / (|w| (|w|+eps)^2)
    float gw0 = gu0 * il, gw1 = gu1 * il, gw2 = gu2 * il;
    if (len > 0.0f) {
        const float k = (gu0 * w0 + gu1 * w1 + gu2 * w2) * il * il / len;
        gw0 -= k * w0; gw1 -= k * w1; gw2 -= k * w2;
    }
    if (dnw) { gw0 += dnw[p]; gw1 += dnw[HW + p]; gw2 += dnw[2 * HW + p]; }
    // w = V3x3 n_view  ->  dL/dn_k = sum_c gw_c V[c][k]
    dallmap[2 * HW + p] = gw0 * V[0] + gw1 * V[4] + gw2 * V[8];
    dallmap[3 * HW + p] = gw0 * V[1] + gw1 * V[5] + gw2 * V[9];
    dallmap[4 * HW + p] = gw0 * V[2] + gw1 * V[6] + gw2 * V[10];
    const float gde = de_ok ? gdep * (1.0f - ratio) : 0.f;
    dallmap[p] = gde / A;
    dallmap[HW + p] = -gde * Dw / (A * A);
    if (!de_ok) { dallmap[p] = 0.f; dallmap[HW + p] = 0.f; }
    dallmap[5 * HW + p] = dm_ok ? gdep * ratio : 0.f;
    dallmap[6 * HW + p] = 0.f;
    if (dray_o) { dray_o[3 * p] = go0; dray_o[3 * p + 1] = go1; dray_o[3 * p + 2] = go2; }
    if (dray_d) {
        // ref_d: (I - 2 u u^T) g ; ref_o: dep * g_o
        dray_d[3 * p] = gd0 - 2.0f * gu_dot * u0 + dep * go0;
        dray_d[3 * p + 1] = gd1 - 2.0f * gu_dot * u1 + dep * go1;
        dray_d[3 * p + 2] = gd2 - 2.0f * gu_dot * u2 + dep * go2;
    }
}


// ------------------------------------------------------------------------------------------------ surface normal (dpt2norm)
// Tail of render() (gaussian2d_utils.py:1125-1142): surface depth = expected depth mixed with the median depth, back-projected through
// the pixel grid (u = x, v = y, no half-pixel offset: dpt2xyz :1158-1187), pseudo normal = normalize(cross(P[y+1] - P[y-1], P[x+1] - P[x-1]))
// on interior pixels (dpt2norm :1190-1206, F.normalize eps 1e-12), zero on the border, times the DETACHED alpha.
struct SurfCam { float fx, fy, cx, cy, r[9]; };       // r = camera-to-world rotation, row major

__device__ __forceinline__ SurfCam surf_cam(int H, int W, float fx, float fy, const float *__restrict__ V)
{
    SurfCam c;
    c.fx = fx; c.fy = fy; c.cx = 0.5f * (float)W; c.cy = 0.5f * (float)H;
    // world_view_transform (row-vector convention) holds R^T in its upper 3x3: that IS the camera-to-world rotation
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.r[i * 3 + j] = V[i * 4 + j];
    return c;
}

__device__ __forceinline__ float surface_depth_at(const float *__restrict__ allmap, int HW, int p, float ratio, bool *de_ok, bool *dm_ok)
{
    float de = allmap[p] / allmap[HW + p];
    *de_ok = fabsf(de) <= 3.0e38f; if (!*de_ok) de = 0.f;               // nan_to_num(., 0, 0)
    float dm = allmap[5 * HW + p];
    *dm_ok = fabsf(dm) <= 3.0e38f; if (!*dm_ok) dm = 0.f;
    return de * (1.0f - ratio) + dm * ratio;
}

__device__ __forceinline__ void pixel_dir(const SurfCam &c, int x, int y, float *d)
{
    const float a = ((float)x - c.cx) / c.fx, b = ((float)y - c.cy) / c.fy;
    d[0] = c.r[0] * a + c.r[1] * b + c.r[2]; d[1] = c.r[3] * a + c.r[4] * b + c.r[5]; d[2] = c.r[6] * a + c.r[7] * b + c.r[8];
}

// normal at interior pixel (x, y) from the four neighbour depths; returns cross product length (0 -> zero normal)
__device__ __forceinline__ float surf_normal_at(const SurfCam &c, int x, int y, float dU, float dD, float dL, float dR, float *n, float *ex, float *ey)
{
    float u[3], d[3], l[3], r[3];
    pixel_dir(c, x, y - 1, u); pixel_dir(c, x, y + 1, d); pixel_dir(c, x - 1, y, l); pixel_dir(c, x + 1, y, r);
#pragma unroll
    for (int k = 0; k < 3; k++) { ex[k] = dD * d[k] - dU * u[k]; ey[k] = dR * r[k] - dL * l[k]; }
    const float c0 = ex[1] * ey[2] - ex[2] * ey[1], c1 = ex[2] * ey[0] - ex[0] * ey[2], c2 = ex[0] * ey[1] - ex[1] * ey[0];
    const float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2), il = 1.0f / fmaxf(len, 1e-12f);
    n[0] = c0 * il; n[1] = c1 * il; n[2] = c2 * il;
    return len;
}

__global__ void __launch_bounds__(256)
surface_normal_fwd(int H, int W, float ratio, float fx, float fy, const float *__restrict__ V, const float *__restrict__ allmap,
                   float *__restrict__ sdepth, float *__restrict__ snormal)
{
    const SurfCam cam = surf_cam(H, W, fx, fy, V);
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    bool a, b;
    sdepth[p] = surface_depth_at(allmap, HW, p, ratio, &a, &b);
    float n[3] = {0.f, 0.f, 0.f};
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        float ex[3], ey[3];
        surf_normal_at(cam, x, y, surface_depth_at(allmap, HW, p - W, ratio, &a, &b), surface_depth_at(allmap, HW, p + W, ratio, &a, &b),
                       surface_depth_at(allmap, HW, p - 1, ratio, &a, &b), surface_depth_at(allmap, HW, p + 1, ratio, &a, &b), n, ex, ey);
    }
    const float al = allmap[HW + p];
    snormal[p] = n[0] * al; snormal[HW + p] = n[1] * al; snormal[2 * HW + p] = n[2] * al;
}

// dL/d(depth of pixel q's neighbour) through the normal at q; which = 0 up (y-1), 1 down (y+1), 2 left, 3 right.  Gather form: every pixel
// asks its four neighbours what they owe it, so nothing is accumulated atomically.
__device__ __forceinline__ float normal_grad_to_neighbour(const SurfCam &c, const float *__restrict__ allmap, const float *__restrict__ gsn, int H, int W,
                                                          float ratio, int qx, int qy, int which)
{
    if (!(qx > 0 && qx < W - 1 && qy > 0 && qy < H - 1)) return 0.f;
    const int HW = H * W, q = qy * W + qx;
    bool a, b;
    float n[3], ex[3], ey[3];
    const float len = surf_normal_at(c, qx, qy, surface_depth_at(allmap, HW, q - W, ratio, &a, &b), surface_depth_at(allmap, HW, q + W, ratio, &a, &b),
                                     surface_depth_at(allmap, HW, q - 1, ratio, &a, &b), surface_depth_at(allmap, HW, q + 1, ratio, &a, &b), n, ex, ey);
    const float al = allmap[HW + q];
    const float g0 = gsn[q] * al, g1 = gsn[HW + q] * al, g2 = gsn[2 * HW + q] * al;
    // n = c / max(|c|, eps): dL/dc = (g - n (n.g)) / |c|   (|c| > eps; below it n = c / eps and dL/dc = g / eps)
    float gc0, gc1, gc2;
    if (len > 1e-12f) {
        const float ng = n[0] * g0 + n[1] * g1 + n[2] * g2, il = 1.0f / len;
        gc0 = (g0 - n[0] * ng) * il; gc1 = (g1 - n[1] * ng) * il; gc2 = (g2 - n[2] * ng) * il;
    } else { gc0 = g0 * 1e12f; gc1 = g1 * 1e12f; gc2 = g2 * 1e12f; }
    // c = ex x ey:  dL/dex = ey x gc,  dL/dey = gc x ex
    float ge[3];
    if (which < 2) { ge[0] = ey[1] * gc2 - ey[2] * gc1; ge[1] = ey[2] * gc0 - ey[0] * gc2; ge[2] = ey[0] * gc1 - ey[1] * gc0; }
    else { ge[0] = gc1 * ex[2] - gc2 * ex[1]; ge[1] = gc2 * ex[0] - gc0 * ex[2]; ge[2] = gc0 * ex[1] - gc1 * ex[0]; }
    float d[3];
    const int nx = which == 2 ? qx - 1 : which == 3 ? qx + 1 : qx, ny = which == 0 ? qy - 1 : which == 1 ? qy + 1 : qy;
    pixel_dir(c, nx, ny, d);
    const float s = (which == 0 || which == 2) ? -1.f : 1.f;             // ex = P[down] - P[up], ey = P[right] - P[left]
    return s * (ge[0] * d[0] + ge[1] * d[1] + ge[2] * d[2]);
}

__global__ void __launch_bounds__(256)
surface_normal_bwd(int H, int W, float ratio, float fx, float fy, const float *__restrict__ V, const float *__restrict__ allmap,
                   const float *__restrict__ gsd, const float *__restrict__ gsn, float *__restrict__ dallmap)
{
    const SurfCam cam = surf_cam(H, W, fx, fy, V);
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    float g = gsd ? gsd[p] : 0.f;
    if (gsn) {
        // this pixel is the UP neighbour of (x, y+1), the DOWN neighbour of (x, y-1), the LEFT neighbour of (x+1, y), the RIGHT one of (x-1, y)
        g += normal_grad_to_neighbour(cam, allmap, gsn, H, W, ratio, x, y + 1, 0);
        g += normal_grad_to_neighbour(cam, allmap, gsn, H, W, ratio, x, y - 1, 1);
        g += normal_grad_to_neighbour(cam, allmap, gsn, H, W, ratio, x + 1, y, 2);
        g += normal_grad_to_neighbour(cam, allmap, gsn, H, W, ratio, x - 1, y, 3);
    }
    bool de_ok, dm_ok;
    surface_depth_at(allmap, HW, p, ratio, &de_ok, &dm_ok);
    const float Dw = allmap[p], A = allmap[HW + p];
    const float gde = de_ok ? g * (1.0f - ratio) : 0.f;
    dallmap[p] = de_ok ? gde / A : 0.f;
    dallmap[HW + p] = de_ok ? -gde * Dw / (A * A) : 0.f;             // (the alpha that scales the normal is detached)
    dallmap[2 * HW + p] = 0.f; dallmap[3 * HW + p] = 0.f; dallmap[4 * HW + p] = 0.f;
    dallmap[5 * HW + p] = dm_ok ? g * ratio : 0.f;
    dallmap[6 * HW + p] = 0.f;
}

// get_disks (optix_utils.py:39-69): one lane per surfel, the four corners of its 3-sigma quad (+ the two triangles' indices)
__global__ void __launch_bounds__(256)
surfel_quads(int P, const float *__restrict__ means, const float *__restrict__ scales, const float *__restrict__ rots, float *__restrict__ v,
             int32_t *__restrict__ f)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float r = rots[4 * i], x = rots[4 * i + 1], y = rots[4 * i + 2], z = rots[4 * i + 3];
    const float inv = 1.0f / sqrtf(r * r + x * x + y * y + z * z);
    r *= inv; x *= inv; y *= inv; z *= inv;
    const float su = 3.0f * scales[2 * i], sv = 3.0f * scales[2 * i + 1];
    const float a[3] = {(1.f - 2.f * (y * y + z * z)) * su, (2.f * (x * y + r * z)) * su, (2.f * (x * z - r * y)) * su};          // rotation column 0
    const float b[3] = {(2.f * (x * y - r * z)) * sv, (1.f - 2.f * (x * x + z * z)) * sv, (2.f * (y * z + r * x)) * sv};          // rotation column 1
    float *o = v + (size_t)i * 12;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float m = means[3 * i + c];
        o[c] = m - a[c] + b[c]; o[3 + c] = m - a[c] - b[c]; o[6 + c] = m + a[c] + b[c]; o[9 + c] = m + a[c] - b[c];
    }
    if (f) {
        int32_t *g = f + (size_t)i * 6;
        g[0] = 4 * i; g[1] = 4 * i + 1; g[2] = 4 * i + 2; g[3] = 4 * i + 1; g[4] = 4 * i + 2; g[5] = 4 * i + 3;
    }
}

// rgb = (1 - s) img[:3] + s rgb_env, one lane per pixel (img is channel-major, rgb_env / rgb pixel-major)
__global__ void __launch_bounds__(256)
blend_fwd(int HW, int C, const float *__restrict__ img, const float *__restrict__ env, float *__restrict__ rgb)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int S = C - 4;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float s = img[(size_t)(3 + (S == 3 ? c : 0)) * HW + p];
        rgb[(size_t)p * 3 + c] = (1.0f - s) * img[(size_t)c * HW + p] + s * env[(size_t)p * 3 + c];
    }
}

__global__ void __launch_bounds__(256)
blend_bwd(int HW, int C, const float *__restrict__ img, const float *__restrict__ env, const float *__restrict__ g, float *__restrict__ dimg,
          float *__restrict__ denv)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int S = C - 4;
    float ds[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int sc = S == 3 ? c : 0;
        const float s = img[(size_t)(3 + sc) * HW + p], gc = g[(size_t)p * 3 + c];
        dimg[(size_t)c * HW + p] = (1.0f - s) * gc;
        if (denv) denv[(size_t)p * 3 + c] = s * gc;
        ds[sc] += (env[(size_t)p * 3 + c] - img[(size_t)c * HW + p]) * gc;
    }
    for (int sc = 0; sc < S; sc++) dimg[(size_t)(3 + sc) * HW + p] = ds[sc];
    dimg[(size_t)(C - 1) * HW + p] = 0.f;
}


// ---- bounce stages of a multi-depth trace (tracing.py:_forward_bounces; gaussian2d_sampler.py:413-426, optix_utils.py:117-118) ----------------
// Stage k's rays that bounce (rows `sel` of its per-ray tensors) become stage k+1's rays:
//     n = norm / |norm|      t = dpt / acc      o2 = o + d t      d2 = d - 2 (d . n) n
// and on the way back the colour of stage k+1 is blended into stage k's:   col[sel] = (1 - s) rgb[sel] + s col_next,  s = aux[sel, 0].
// One launch each way instead of ~20 torch kernels forward and ~30 backward per stage (gathers, a norm, divisions, the index_put chain
// and their scatter-adds); `sel` holds unique row indices (a nonzero() result), so the backward writes each row once -- no atomics.
__global__ void __launch_bounds__(256)
bounce_rays_fwd(int n, const long long *__restrict__ sel, const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ dpt,
                const float *__restrict__ acc, const float *__restrict__ norm, float *__restrict__ o2, float *__restrict__ d2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t r = (size_t)sel[i];
    const float ox = o[3 * r], oy = o[3 * r + 1], oz = o[3 * r + 2], dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
    const float nx = norm[3 * r], ny = norm[3 * r + 1], nz = norm[3 * r + 2];
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    const float hx = nx / len, hy = ny / len, hz = nz / len;
    const float t = dpt[r] / acc[r];
    o2[3 * i] = ox + dx * t; o2[3 * i + 1] = oy + dy * t; o2[3 * i + 2] = oz + dz * t;
    const float dn = dx * hx + dy * hy + dz * hz;
    d2[3 * i] = dx - 2.0f * dn * hx; d2[3 * i + 1] = dy - 2.0f * dn * hy; d2[3 * i + 2] = dz - 2.0f * dn * hz;
}

// g_o / g_d / g_dpt / g_acc / g_norm: (R_k, .) buffers ZEROED by the caller (rows that do not bounce receive nothing)
__global__ void __launch_bounds__(256)
bounce_rays_bwd(int n, const long long *__restrict__ sel, const float *__restrict__ o, const float *__restrict__ d, const float *__restrict__ dpt,
                const float *__restrict__ acc, const float *__restrict__ norm, const float *__restrict__ g_o2, const float *__restrict__ g_d2,
                float *__restrict__ g_o, float *__restrict__ g_d, float *__restrict__ g_dpt, float *__restrict__ g_acc, float *__restrict__ g_norm)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t r = (size_t)sel[i];
    const float dx = d[3 * r], dy = d[3 * r + 1], dz = d[3 * r + 2];
    const float nx = norm[3 * r], ny = norm[3 * r + 1], nz = norm[3 * r + 2];
    const float len = sqrtf(nx * nx + ny * ny + nz * nz), il = 1.0f / len;
    const float hx = nx * il, hy = ny * il, hz = nz * il;
    const float a = acc[r], t = dpt[r] / a;
    const float ax = g_o2 ? g_o2[3 * i] : 0.f, ay = g_o2 ? g_o2[3 * i + 1] : 0.f, az = g_o2 ? g_o2[3 * i + 2] : 0.f;
    const float bx = g_d2 ? g_d2[3 * i] : 0.f, by = g_d2 ? g_d2[3 * i + 1] : 0.f, bz = g_d2 ? g_d2[3 * i + 2] : 0.f;
    const float dn = dx * hx + dy * hy + dz * hz, bn = bx * hx + by * hy + bz * hz;
    if (g_o) { g_o[3 * r] = ax; g_o[3 * r + 1] = ay; g_o[3 * r + 2] = az; }
    if (g_d) {                                            // d o2 / d d = t I ;  d d2 / d d = I - 2 n n^T (symmetric)
        g_d[3 * r] = ax * t + bx - 2.0f * bn * hx; g_d[3 * r + 1] = ay * t + by - 2.0f * bn * hy; g_d[3 * r + 2] = az * t + bz - 2.0f * bn * hz;
    }
    const float gt = ax * dx + ay * dy + az * dz;         // dL/dt
    if (g_dpt) g_dpt[r] = gt / a;
    if (g_acc) g_acc[r] = -gt * t / a;
    if (g_norm) {
        // dL/dn_hat = -2 [(d . n) g_d2 + (g_d2 . n) d], projected off n_hat and scaled by 1 / |norm|
        const float qx = -2.0f * (dn * bx + bn * dx), qy = -2.0f * (dn * by + bn * dy), qz = -2.0f * (dn * bz + bn * dz);
        const float qn = qx * hx + qy * hy + qz * hz;
        g_norm[3 * r] = (qx - qn * hx) * il; g_norm[3 * r + 1] = (qy - qn * hy) * il; g_norm[3 * r + 2] = (qz - qn * hz) * il;
    }
}

// col: a COPY of rgb (R_k, 3) made by the caller; rows sel are overwritten with the blend
__global__ void __launch_bounds__(256)
bounce_blend_fwd(int n, const long long *__restrict__ sel, const float *__restrict__ rgb, const float *__restrict__ aux,
                 const float *__restrict__ col_next, float *__restrict__ col)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t r = (size_t)sel[i];
    const float s = aux[2 * r];
#pragma unroll
    for (int c = 0; c < 3; c++) col[3 * r + c] = (1.0f - s) * rgb[3 * r + c] + s * col_next[3 * i + c];
}

// g_rgb: a COPY of g_col (R_k, 3) made by the caller (rows that do not bounce pass their gradient through); g_aux (R_k, 2) zeroed by the caller
__global__ void __launch_bounds__(256)
bounce_blend_bwd(int n, const long long *__restrict__ sel, const float *__restrict__ rgb, const float *__restrict__ aux,
                 const float *__restrict__ col_next, const float *__restrict__ g_col, float *__restrict__ g_rgb, float *__restrict__ g_aux,
                 float *__restrict__ g_col_next)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t r = (size_t)sel[i];
    const float s = aux[2 * r];
    float gs = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float g = g_col[3 * r + c];
        gs += (col_next[3 * i + c] - rgb[3 * r + c]) * g;
        if (g_rgb) g_rgb[3 * r + c] = (1.0f - s) * g;
        if (g_col_next) g_col_next[3 * i + c] = s * g;
    }
    if (g_aux) g_aux[2 * r] = gs;
}

// mid (R, 16 * stages): the 16 channels [o 3 | d 3 | dpt | acc | norm 3 | aux 2 | rgb 3] of stage k's rays at the rows `idx` (their pixels)
__global__ void __launch_bounds__(256)
bounce_pack_mid(int n, const long long *__restrict__ idx, int stride, int k, const float *__restrict__ o, const float *__restrict__ d,
                const float *__restrict__ dpt, const float *__restrict__ acc, const float *__restrict__ norm, const float *__restrict__ aux,
                const float *__restrict__ rgb, float *__restrict__ mid)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 *m = reinterpret_cast<float4 *>(mid + (size_t)(idx ? idx[i] : i) * stride + 16 * k);
    m[0] = make_float4(o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i]);
    m[1] = make_float4(d[3 * i + 1], d[3 * i + 2], dpt[i], acc[i]);
    m[2] = make_float4(norm[3 * i], norm[3 * i + 1], norm[3 * i + 2], aux[2 * i]);
    m[3] = make_float4(aux[2 * i + 1], rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
}

}  // namespace envgs

using namespace envgs;

extern "C" {

int envgs_sh_colors_forward(int32_t P, int32_t D, int32_t M, int32_t S, const float *means3D, const float *shs, const float *campos,
                            const float *specular, const float *roughness, float *colors, uint8_t *clamped, void *stream)
{
    if (P < 0 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || (S != 1 && S != 3)) return ENVGS_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (!means3D || !shs || !campos || !specular || !roughness || !colors || !clamped) return ENVGS_ERR_BAD_ARG;
    if (M == 16 && (reinterpret_cast<uintptr_t>(shs) & 15) == 0)
        hipLaunchKernelGGL(sh_colors_fwd_q16, dim3((4 * (size_t)P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, D, S, means3D, shs, campos, specular,
                           roughness, colors, clamped);
    else
        hipLaunchKernelGGL(sh_colors_fwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, D, M, S, means3D, shs, campos, specular,
                           roughness, colors, clamped);
    return (int)hipGetLastError();
}

int envgs_sh_colors_backward(int32_t P, int32_t D, int32_t M, int32_t S, const float *means3D, const float *shs, const float *campos,
                             const uint8_t *clamped, const float *dcolors, float *dmeans3D, float *dshs, float *dspecular,
                             float *droughness, void *stream)
{
    if (P < 0 || D < 0 || D > 3 || M < (D + 1) * (D + 1) || (S != 1 && S != 3)) return ENVGS_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (!means3D || !shs || !campos || !clamped || !dcolors || !dmeans3D || !dshs || !dspecular || !droughness) return ENVGS_ERR_BAD_ARG;
    if (M == 16 && ((reinterpret_cast<uintptr_t>(shs) | reinterpret_cast<uintptr_t>(dshs)) & 15) == 0)
        hipLaunchKernelGGL(sh_colors_bwd_q16, dim3((4 * (size_t)P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, D, S, means3D, shs, campos, clamped,
                           dcolors, dmeans3D, dshs, dspecular, droughness);
    else
        hipLaunchKernelGGL(sh_colors_bwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, D, M, S, means3D, shs, campos, clamped,
                           dcolors, dmeans3D, dshs, dspecular, droughness);
    return (int)hipGetLastError();
}

int envgs_reflect_forward(int32_t H, int32_t W, float depth_ratio, const float *allmap, const float *ray_o, const float *ray_d,
                          const float *viewmatrix, float *normal_world, float *depth, float *ref_o, float *ref_d, void *stream)
{
    if (H <= 0 || W <= 0 || !allmap || !ray_o || !ray_d || !viewmatrix || !normal_world || !depth || !ref_o || !ref_d) return ENVGS_ERR_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(reflect_fwd, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, depth_ratio, allmap, ray_o, ray_d, viewmatrix,
                       normal_world, depth, ref_o, ref_d);
    return (int)hipGetLastError();
}

int envgs_reflect_backward(int32_t H, int32_t W, float depth_ratio, const float *allmap, const float *ray_o, const float *ray_d,
                           const float *viewmatrix, const float *dnormal_world, const float *ddepth, const float *dref_o,
                           const float *dref_d, float *dallmap, float *dray_o, float *dray_d, void *stream)
{
    if (H <= 0 || W <= 0 || !allmap || !ray_o || !ray_d || !viewmatrix || !dallmap) return ENVGS_ERR_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(reflect_bwd, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, depth_ratio, allmap, ray_o, ray_d, viewmatrix,
                       dnormal_world, ddepth, dref_o, dref_d, dallmap, dray_o, dray_d);
    return (int)hipGetLastError();
}

int envgs_surface_normal_forward(int32_t H, int32_t W, float depth_ratio, float fx, float fy, const float *allmap, const float *viewmatrix,
                                 float *surf_depth, float *surf_normal, void *stream)
{
    if (H <= 0 || W <= 0 || !(fx > 0.f) || !(fy > 0.f) || !allmap || !viewmatrix || !surf_depth || !surf_normal) return ENVGS_ERR_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(surface_normal_fwd, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, depth_ratio, fx, fy, viewmatrix,
                       allmap, surf_depth, surf_normal);
    return (int)hipGetLastError();
}

int envgs_surface_normal_backward(int32_t H, int32_t W, float depth_ratio, float fx, float fy, const float *allmap, const float *viewmatrix,
                                  const float *dsurf_depth, const float *dsurf_normal, float *dallmap, void *stream)
{
    if (H <= 0 || W <= 0 || !(fx > 0.f) || !(fy > 0.f) || !allmap || !viewmatrix || !dallmap) return ENVGS_ERR_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(surface_normal_bwd, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, depth_ratio, fx, fy, viewmatrix,
                       allmap, dsurf_depth, dsurf_normal, dallmap);
    return (int)hipGetLastError();
}

int envgs_surfel_quads(int32_t P, const float *means3D, const float *scales, const float *rotations, float *vertices, int32_t *faces, void *stream)
{
    if (P < 0) return ENVGS_ERR_BAD_ARG;
    if (P == 0) return 0;
    if (!means3D || !scales || !rotations || !vertices) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(surfel_quads, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, means3D, scales, rotations, vertices, faces);
    return (int)hipGetLastError();
}

int envgs_blend_forward(int32_t H, int32_t W, int32_t channels, const float *img, const float *rgb_env, float *rgb, void *stream)
{
    if (H <= 0 || W <= 0 || (channels != 5 && channels != 7) || !img || !rgb_env || !rgb) return ENVGS_ERR_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(blend_fwd, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, channels, img, rgb_env, rgb);
    return (int)hipGetLastError();
}

int envgs_blend_backward(int32_t H, int32_t W, int32_t channels, const float *img, const float *rgb_env, const float *drgb, float *dimg,
                         float *drgb_env, void *stream)
{
    if (H <= 0 || W <= 0 || (channels != 5 && channels != 7) || !img || !rgb_env || !drgb || !dimg) return ENVGS_ERR_BAD_ARG;
    const int HW = H * W;
    hipLaunchKernelGGL(blend_bwd, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, HW, channels, img, rgb_env, drgb, dimg, drgb_env);
    return (int)hipGetLastError();
}

int envgs_bounce_rays_forward(int32_t n, const int64_t *sel, const float *ray_o, const float *ray_d, const float *dpt, const float *acc,
                              const float *norm, float *o2, float *d2, void *stream)
{
    if (n < 0) return ENVGS_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!sel || !ray_o || !ray_d || !dpt || !acc || !norm || !o2 || !d2) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bounce_rays_fwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const long long *)sel, ray_o, ray_d, dpt, acc, norm, o2, d2);
    return (int)hipGetLastError();
}

int envgs_bounce_rays_backward(int32_t n, const int64_t *sel, const float *ray_o, const float *ray_d, const float *dpt, const float *acc,
                               const float *norm, const float *g_o2, const float *g_d2, float *g_ray_o, float *g_ray_d, float *g_dpt,
                               float *g_acc, float *g_norm, void *stream)
{
    if (n < 0) return ENVGS_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!sel || !ray_o || !ray_d || !dpt || !acc || !norm) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bounce_rays_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const long long *)sel, ray_o, ray_d, dpt, acc, norm,
                       g_o2, g_d2, g_ray_o, g_ray_d, g_dpt, g_acc, g_norm);
    return (int)hipGetLastError();
}

int envgs_bounce_blend_forward(int32_t n, const int64_t *sel, const float *rgb, const float *aux, const float *col_next, float *col, void *stream)
{
    if (n < 0) return ENVGS_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!sel || !rgb || !aux || !col_next || !col) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bounce_blend_fwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const long long *)sel, rgb, aux, col_next, col);
    return (int)hipGetLastError();
}

int envgs_bounce_blend_backward(int32_t n, const int64_t *sel, const float *rgb, const float *aux, const float *col_next, const float *g_col,
                                float *g_rgb, float *g_aux, float *g_col_next, void *stream)
{
    if (n < 0) return ENVGS_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!sel || !rgb || !aux || !col_next || !g_col) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bounce_blend_bwd, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const long long *)sel, rgb, aux, col_next, g_col,
                       g_rgb, g_aux, g_col_next);
    return (int)hipGetLastError();
}

int envgs_bounce_pack_mid(int32_t n, const int64_t *idx, int32_t stages, int32_t k, const float *ray_o, const float *ray_d, const float *dpt,
                          const float *acc, const float *norm, const float *aux, const float *rgb, float *mid, void *stream)
{
    if (n < 0 || stages <= 0 || k < 0 || k >= stages) return ENVGS_ERR_BAD_ARG;
    if (n == 0) return 0;
    if (!ray_o || !ray_d || !dpt || !acc || !norm || !aux || !rgb || !mid) return ENVGS_ERR_BAD_ARG;
    hipLaunchKernelGGL(bounce_pack_mid, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, (const long long *)idx, 16 * stages, k, ray_o, ray_d,
                       dpt, acc, norm, aux, rgb, mid);
    return (int)hipGetLastError();
}

}  // extern "C"
