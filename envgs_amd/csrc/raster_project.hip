// raster_project.hip -- R1: per-surfel projection + Jacobian (transMat), view normal, 3-sigma AABB,
// radius / tile rect, SH -> RGB.  One lane per surfel, 256-lane workgroups.
//
// This translation unit is compiled with -ffp-contract=off: radius, tile rect, tiles_touched and the
// fp32 depth bits feed integer sort keys that must be BIT-EXACT against the oracle, so every expression
// below is written in one fixed evaluation order with no FMA contraction (SURVEY.md section 7,
// "bit-exact tile/sort indices").
//
// Stands behind GaussianRasterizer.forward's preprocess stage; boundary: easyvolcap/utils/gaussian2d_utils.py:1089-1099,
// transMat definition: :1050-1061, quaternion convention: :145-178, SH basis: easyvolcap/utils/sh_utils.py:642-727.
#include "common.h"

namespace envgs {

__device__ __constant__ float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f};
constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;

__device__ __forceinline__ int clampi(int a, int lo, int hi) { return a < lo ? lo : (a > hi ? hi : a); }

__global__ void __launch_bounds__(256)
project_surfels(int P, int D, int M, int f16, int C, int W, int H, float mod,
                const float *__restrict__ means3D, const float *__restrict__ scales,
                const float *__restrict__ rotations, const float *__restrict__ opacities,
                const float *__restrict__ shs, const float *__restrict__ transmat_precomp,
                const float *__restrict__ V, const float *__restrict__ FP, const float *__restrict__ campos,
                float *__restrict__ geom, float *__restrict__ rgb, uint8_t *__restrict__ clamped,
                int32_t *__restrict__ radii, uint32_t *__restrict__ tiles_touched)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    radii[i] = 0;
    tiles_touched[i] = 0;

    const float p0 = means3D[3 * i + 0], p1 = means3D[3 * i + 1], p2 = means3D[3 * i + 2];
    const float pvx = V[0] * p0 + V[4] * p1 + V[8] * p2 + V[12];
    const float pvy = V[1] * p0 + V[5] * p1 + V[9] * p2 + V[13];
    const float pvz = V[2] * p0 + V[6] * p1 + V[10] * p2 + V[14];
    if (pvz <= NEAR_N) return;

    // world -> homogeneous pixel matrix (rows r, columns x*w, y*w, w)
    const float hw = (float)W / 2.0f, hh = (float)H / 2.0f;
    const float cw = (float)(W - 1) / 2.0f, ch = (float)(H - 1) / 2.0f;
    float PM[12];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        PM[r * 3 + 0] = hw * FP[r * 4 + 0] + cw * FP[r * 4 + 3];
        PM[r * 3 + 1] = hh * FP[r * 4 + 1] + ch * FP[r * 4 + 3];
        PM[r * 3 + 2] = FP[r * 4 + 3];
    }

    float T[9], n0, n1, n2;
    if (transmat_precomp) {
#pragma unroll
        for (int c = 0; c < 9; c++) T[c] = transmat_precomp[9 * i + c];
        n0 = 0.f; n1 = 0.f; n2 = 1.f;
    } else {
        const float q0 = rotations[4 * i + 0], q1 = rotations[4 * i + 1], q2 = rotations[4 * i + 2], q3 = rotations[4 * i + 3];
        const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        const float inv = 1.0f / qn;
        const float r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
        float R[9];
        R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
        R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
        R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
        const float s0 = scales[2 * i + 0] * mod, s1 = scales[2 * i + 1] * mod;
        const float a0 = R[0] * s0, a1 = R[3] * s0, a2 = R[6] * s0;
        const float b0 = R[1] * s1, b1 = R[4] * s1, b2 = R[7] * s1;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            T[c * 3 + 0] = a0 * PM[0 + c] + a1 * PM[3 + c] + a2 * PM[6 + c];
            T[c * 3 + 1] = b0 * PM[0 + c] + b1 * PM[3 + c] + b2 * PM[6 + c];
            T[c * 3 + 2] = p0 * PM[0 + c] + p1 * PM[3 + c] + p2 * PM[6 + c] + PM[9 + c];
        }
        const float w0 = R[2], w1 = R[5], w2 = R[8];
        n0 = V[0] * w0 + V[4] * w1 + V[8] * w2;
        n1 = V[1] * w0 + V[5] * w1 + V[9] * w2;
        n2 = V[2] * w0 + V[6] * w1 + V[10] * w2;
    }
    float *gr = geom + (size_t)i * GEOM;
#pragma unroll
    for (int c = 0; c < 9; c++) gr[c] = T[c];

    const float cosv = -(pvx * n0 + pvy * n1 + pvz * n2);
    if (cosv == 0.0f) return;
    if (cosv < 0.0f) { n0 = -n0; n1 = -n1; n2 = -n2; }

    const float *Tu = T, *Tv = T + 3, *Tw = T + 6;
    const float t0 = 9.0f, t1 = 9.0f, t2 = -1.0f;
    const float d = t0 * (Tw[0] * Tw[0]) + t1 * (Tw[1] * Tw[1]) + t2 * (Tw[2] * Tw[2]);
    if (d == 0.0f) return;
    const float f0 = (1.0f / d) * t0, f1 = (1.0f / d) * t1, f2 = (1.0f / d) * t2;
    const float cx = f0 * (Tu[0] * Tw[0]) + f1 * (Tu[1] * Tw[1]) + f2 * (Tu[2] * Tw[2]);
    const float cy = f0 * (Tv[0] * Tw[0]) + f1 * (Tv[1] * Tw[1]) + f2 * (Tv[2] * Tw[2]);
    const float hx = cx * cx - (f0 * (Tu[0] * Tu[0]) + f1 * (Tu[1] * Tu[1]) + f2 * (Tu[2] * Tu[2]));
    const float hy = cy * cy - (f0 * (Tv[0] * Tv[0]) + f1 * (Tv[1] * Tv[1]) + f2 * (Tv[2] * Tv[2]));
    const float ex = sqrtf(hx > 1e-4f ? hx : 1e-4f), ey = sqrtf(hy > 1e-4f ? hy : 1e-4f);
    const float emax = ex > ey ? ex : ey;
    const float rmin = 3.0f * FILTER_SIZE;
    const float radius = ceilf(emax > rmin ? emax : rmin);

    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int x0 = clampi((int)((cx - radius) / (float)TILE), 0, gx);
    const int y0 = clampi((int)((cy - radius) / (float)TILE), 0, gy);
    const int x1 = clampi((int)((cx + radius + (float)(TILE - 1)) / (float)TILE), 0, gx);
    const int y1 = clampi((int)((cy + radius + (float)(TILE - 1)) / (float)TILE), 0, gy);
    if ((x1 - x0) * (y1 - y0) == 0) return;

    if (shs) {
        const float dx = p0 - campos[0], dy = p1 - campos[1], dz = p2 - campos[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx / len, y = dy / len, z = dz / len;
        const Feat sh = Feat{shs, f16 != 0}.at((size_t)i * M * 3);          // fp32 or fp16 storage, converted on load
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float r = kC0 * sh[0 * 3 + c];
            if (D > 0) {
                r = r - kC1 * y * sh[1 * 3 + c] + kC1 * z * sh[2 * 3 + c] - kC1 * x * sh[3 * 3 + c];
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    r = r + kC2[0] * xy * sh[4 * 3 + c] + kC2[1] * yz * sh[5 * 3 + c] +
                        kC2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + kC2[3] * xz * sh[7 * 3 + c] +
                        kC2[4] * (xx - yy) * sh[8 * 3 + c];
                    if (D > 2) {
                        r = r + kC3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] + kC3[1] * xy * z * sh[10 * 3 + c] +
                            kC3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
                            kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
                            kC3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
                            kC3[5] * z * (xx - yy) * sh[14 * 3 + c] + kC3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
                    }
                }
            }
            r += 0.5f;
            clamped[3 * i + c] = (uint8_t)(r < 0.0f);
            rgb[(size_t)i * C + c] = r < 0.0f ? 0.0f : r;
        }
    }

    gr[9] = cx; gr[10] = cy;
    gr[11] = n0; gr[12] = n1; gr[13] = n2;
    gr[14] = opacities[i];
    gr[15] = pvz;
    radii[i] = (int32_t)radius;
    tiles_touched[i] = (uint32_t)((y1 - y0) * (x1 - x0));
}

int launch_project(const envgs_raster_cfg *cfg, const float *means3D, const float *scales, const float *rotations,
                   const float *opacities, const float *shs, const float *transmat_precomp, const float *viewmatrix,
                   const float *projmatrix, const float *campos, float *geom, float *rgb, uint8_t *clamped,
                   int32_t *radii, uint32_t *tiles_touched, hipStream_t stream)
{
    const int P = cfg->P;
    if (P <= 0) return 0;
    const int blocks = (P + 255) / 256;
    ProfScope prof_(K_PROJECT, stream);
    hipLaunchKernelGGL(project_surfels, dim3(blocks), dim3(256), 0, stream, P, cfg->sh_degree, cfg->sh_coeffs, cfg->feature_f16,
                       cfg->channels, cfg->width, cfg->height, cfg->scale_modifier, means3D, scales, rotations,
                       opacities, shs, transmat_precomp, viewmatrix, projmatrix, campos, geom, rgb, clamped, radii,
                       tiles_touched);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    return 0;
}

}  // namespace envgs
