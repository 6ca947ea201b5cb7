// raster_project_bwd.hip -- R8: chain the per-surfel gradient record (transMat 9, normal 3, opacity 1,
// mean2D 2, colour C) to means3D / scales / rotations / SH coefficients (or the precomputed transMat /
// colours), and produce the means2D densification proxy.  One lane per surfel; HBM-bound:
// 128 B record + 40 B params (+192 B SH) in, 40 B (+192 B) out per surfel.
//
// Stands behind GaussianRasterizer.backward's preprocess stage (call site easyvolcap/utils/gaussian2d_utils.py:1089-1099;
// the consumer of means2D.grad is :901-909).  Restated in oracle/surfel_raster_oracle.c::orc_preprocess_bwd.
#include "common.h"

namespace envgs {

constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
__device__ __constant__ float bC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                        -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ float bC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                        0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                        -0.5900435899266435f};

__global__ void __launch_bounds__(256)
project_surfels_bwd(int P, int D, int M, int f16, int C, int W, int H, float mod, const float *__restrict__ geom,
                    const float *__restrict__ means3D, const float *__restrict__ scales,
                    const float *__restrict__ rotations, const float *__restrict__ shs,
                    const uint8_t *__restrict__ clamped, const float *__restrict__ transmat_precomp,
                    const int32_t *__restrict__ radii, const float *__restrict__ V, const float *__restrict__ FP,
                    const float *__restrict__ campos, const float *__restrict__ grad_rec,
                    float *__restrict__ dmeans3D, float *__restrict__ dmeans2D, float *__restrict__ dscales,
                    float *__restrict__ drots, float *__restrict__ dshs, float *__restrict__ dcolors,
                    float *__restrict__ dopacities, float *__restrict__ dtransmat_precomp, int sh_split)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const bool vis = radii[i] > 0;
    const float *rec = grad_rec + (size_t)i * GREC;

    float dm3[3] = {0.f, 0.f, 0.f}, dm2x = 0.f, dm2y = 0.f;
    float dsc0 = 0.f, dsc1 = 0.f, dq[4] = {0.f, 0.f, 0.f, 0.f};
    float dT[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (vis) {
        const float *T = geom + (size_t)i * GEOM;
#pragma unroll
        for (int c = 0; c < 9; c++) dT[c] = rec[c];
        const float raw2 = dT[2], raw5 = dT[5];
        const float dmx = rec[13], dmy = rec[14];
        if (dmx != 0.f || dmy != 0.f) {
            const float t[3] = {9.0f, 9.0f, -1.0f};
            const float *Tu = T, *Tv = T + 3, *Tw = T + 6;
            const float d = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
            const float invd = 1.0f / d;
            float f[3], dTw[3], df[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                f[c] = t[c] * invd;
                dT[c] += dmx * f[c] * Tw[c];
                dT[3 + c] += dmy * f[c] * Tw[c];
                dTw[c] = dmx * f[c] * Tu[c] + dmy * f[c] * Tv[c];
                df[c] = dmx * Tu[c] * Tw[c] + dmy * Tv[c] * Tw[c];
            }
            const float dL_dd = (df[0] * f[0] + df[1] * f[1] + df[2] * f[2]) * (-invd);
#pragma unroll
            for (int c = 0; c < 3; c++) dT[6 + c] += dTw[c] + dL_dd * (t[c] * Tw[c] * 2.0f);
        }

        float hack_x, hack_y;
        if (transmat_precomp) {
            hack_x = dT[2]; hack_y = dT[5];
        } else {
            hack_x = raw2; hack_y = raw5;
            const float hw = (float)W / 2.0f, hh = (float)H / 2.0f;
            const float cw = (float)(W - 1) / 2.0f, ch = (float)(H - 1) / 2.0f;
            float PM[12];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                PM[r * 3 + 0] = hw * FP[r * 4 + 0] + cw * FP[r * 4 + 3];
                PM[r * 3 + 1] = hh * FP[r * 4 + 1] + ch * FP[r * 4 + 3];
                PM[r * 3 + 2] = FP[r * 4 + 3];
            }
            const float p0 = means3D[3 * i], p1 = means3D[3 * i + 1], p2 = means3D[3 * i + 2];
            const float q0 = rotations[4 * i], q1 = rotations[4 * i + 1], q2 = rotations[4 * i + 2], q3 = rotations[4 * i + 3];
            const float inv = 1.0f / sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            const float r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
            float R[9];
            R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z);       R[2] = 2.f * (x * z + r * y);
            R[3] = 2.f * (x * y + r * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
            R[6] = 2.f * (x * z - r * y);       R[7] = 2.f * (y * z + r * x);       R[8] = 1.f - 2.f * (x * x + y * y);
            const float sx = scales[2 * i] * mod, sy = scales[2 * i + 1] * mod;
            float da[3] = {0.f, 0.f, 0.f}, db[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int rr = 0; rr < 3; rr++) {
                    da[rr] += dT[c * 3 + 0] * PM[rr * 3 + c];
                    db[rr] += dT[c * 3 + 1] * PM[rr * 3 + c];
                    dm3[rr] += dT[c * 3 + 2] * PM[rr * 3 + c];
                }
            const float dnv0 = rec[9], dnv1 = rec[10], dnv2 = rec[11];
            float dnw[3] = {V[0] * dnv0 + V[1] * dnv1 + V[2] * dnv2, V[4] * dnv0 + V[5] * dnv1 + V[6] * dnv2,
                            V[8] * dnv0 + V[9] * dnv1 + V[10] * dnv2};
            {
                const float pvx = V[0] * p0 + V[4] * p1 + V[8] * p2 + V[12];
                const float pvy = V[1] * p0 + V[5] * p1 + V[9] * p2 + V[13];
                const float pvz = V[2] * p0 + V[6] * p1 + V[10] * p2 + V[14];
                const float nv0 = V[0] * R[2] + V[4] * R[5] + V[8] * R[8];
                const float nv1 = V[1] * R[2] + V[5] * R[5] + V[9] * R[8];
                const float nv2 = V[2] * R[2] + V[6] * R[5] + V[10] * R[8];
                const float cosv = -(pvx * nv0 + pvy * nv1 + pvz * nv2);
                if (!(cosv > 0.0f)) { dnw[0] = -dnw[0]; dnw[1] = -dnw[1]; dnw[2] = -dnw[2]; }
            }
            float dR[9];
#pragma unroll
            for (int rr = 0; rr < 3; rr++) { dR[rr * 3 + 0] = da[rr] * sx; dR[rr * 3 + 1] = db[rr] * sy; dR[rr * 3 + 2] = dnw[rr]; }
            dsc0 = (da[0] * R[0] + da[1] * R[3] + da[2] * R[6]) * mod;
            dsc1 = (db[0] * R[1] + db[1] * R[4] + db[2] * R[7]) * mod;
#define VR(a, b) dR[(a) * 3 + (b)]
            dq[0] = 2.f * (x * (VR(2, 1) - VR(1, 2)) + y * (VR(0, 2) - VR(2, 0)) + z * (VR(1, 0) - VR(0, 1)));
            dq[1] = 2.f * (-2.f * x * (VR(1, 1) + VR(2, 2)) + y * (VR(1, 0) + VR(0, 1)) + z * (VR(2, 0) + VR(0, 2)) + r * (VR(2, 1) - VR(1, 2)));
            dq[2] = 2.f * (x * (VR(1, 0) + VR(0, 1)) - 2.f * y * (VR(0, 0) + VR(2, 2)) + z * (VR(2, 1) + VR(1, 2)) + r * (VR(0, 2) - VR(2, 0)));
            dq[3] = 2.f * (x * (VR(2, 0) + VR(0, 2)) + y * (VR(2, 1) + VR(1, 2)) - 2.f * z * (VR(0, 0) + VR(1, 1)) + r * (VR(1, 0) - VR(0, 1)));
#undef VR
        }

        if (shs && !sh_split) {       // (sh_split: the SH part runs four lanes per surfel in sh_record_bwd_q16, right after this kernel)
            const float p0 = means3D[3 * i], p1 = means3D[3 * i + 1], p2 = means3D[3 * i + 2];
            const Feat sh = Feat{shs, f16 != 0}.at((size_t)i * M * 3);          // fp32 or fp16 storage, converted on load
            float *dsh = dshs + (size_t)i * M * 3;
            const float dirx = p0 - campos[0], diry = p1 - campos[1], dirz = p2 - campos[2];
            const float sum2 = dirx * dirx + diry * diry + dirz * dirz;
            const float len = sqrtf(sum2), ilen = 1.0f / len;
            const float x = dirx / len, y = diry / len, z = dirz / len;           // (the oracle's normalisation: three divisions)
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float g = clamped[3 * i + c] ? 0.f : rec[15 + c];
                float gx_ = 0.f, gy_ = 0.f, gz_ = 0.f;
                dsh[0 * 3 + c] = kC0 * g;
                if (D > 0) {
                    dsh[1 * 3 + c] = -kC1 * y * g; dsh[2 * 3 + c] = kC1 * z * g; dsh[3 * 3 + c] = -kC1 * x * g;
                    gx_ = -kC1 * sh[3 * 3 + c]; gy_ = -kC1 * sh[1 * 3 + c]; gz_ = kC1 * sh[2 * 3 + c];
                    if (D > 1) {
                        dsh[4 * 3 + c] = bC2[0] * xy * g; dsh[5 * 3 + c] = bC2[1] * yz * g;
                        dsh[6 * 3 + c] = bC2[2] * (2.f * zz - xx - yy) * g; dsh[7 * 3 + c] = bC2[3] * xz * g;
                        dsh[8 * 3 + c] = bC2[4] * (xx - yy) * g;
                        gx_ += bC2[0] * y * sh[4 * 3 + c] + bC2[2] * 2.f * -x * sh[6 * 3 + c] + bC2[3] * z * sh[7 * 3 + c] + bC2[4] * 2.f * x * sh[8 * 3 + c];
                        gy_ += bC2[0] * x * sh[4 * 3 + c] + bC2[1] * z * sh[5 * 3 + c] + bC2[2] * 2.f * -y * sh[6 * 3 + c] + bC2[4] * 2.f * -y * sh[8 * 3 + c];
                        gz_ += bC2[1] * y * sh[5 * 3 + c] + bC2[2] * 4.f * z * sh[6 * 3 + c] + bC2[3] * x * sh[7 * 3 + c];
                        if (D > 2) {
                            dsh[9 * 3 + c] = bC3[0] * y * (3.f * xx - yy) * g; dsh[10 * 3 + c] = bC3[1] * xy * z * g;
                            dsh[11 * 3 + c] = bC3[2] * y * (4.f * zz - xx - yy) * g;
                            dsh[12 * 3 + c] = bC3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * g;
                            dsh[13 * 3 + c] = bC3[4] * x * (4.f * zz - xx - yy) * g;
                            dsh[14 * 3 + c] = bC3[5] * z * (xx - yy) * g; dsh[15 * 3 + c] = bC3[6] * x * (xx - 3.f * yy) * g;
                            gx_ += bC3[0] * sh[9 * 3 + c] * 6.f * xy + bC3[1] * sh[10 * 3 + c] * yz + bC3[2] * sh[11 * 3 + c] * -2.f * xy +
                                   bC3[3] * sh[12 * 3 + c] * -6.f * xz + bC3[4] * sh[13 * 3 + c] * (-3.f * xx + 4.f * zz - yy) +
                                   bC3[5] * sh[14 * 3 + c] * 2.f * xz + bC3[6] * sh[15 * 3 + c] * 3.f * (xx - yy);
                            gy_ += bC3[0] * sh[9 * 3 + c] * 3.f * (xx - yy) + bC3[1] * sh[10 * 3 + c] * xz +
                                   bC3[2] * sh[11 * 3 + c] * (-3.f * yy + 4.f * zz - xx) + bC3[3] * sh[12 * 3 + c] * -6.f * yz +
                                   bC3[4] * sh[13 * 3 + c] * -2.f * xy + bC3[5] * sh[14 * 3 + c] * -2.f * yz + bC3[6] * sh[15 * 3 + c] * -6.f * xy;
                            gz_ += bC3[1] * sh[10 * 3 + c] * xy + bC3[2] * sh[11 * 3 + c] * 8.f * yz +
                                   bC3[3] * sh[12 * 3 + c] * 3.f * (2.f * zz - xx - yy) + bC3[4] * sh[13 * 3 + c] * 8.f * xz +
                                   bC3[5] * sh[14 * 3 + c] * (xx - yy);
                        }
                    }
                }
                ddx += gx_ * g; ddy += gy_ * g; ddz += gz_ * g;
                for (int k = (D + 1) * (D + 1); k < M; k++) dsh[k * 3 + c] = 0.f;
            }
            const float invsum32 = ilen * ilen * ilen;
            dm3[0] += ((sum2 - dirx * dirx) * ddx - diry * dirx * ddy - dirz * dirx * ddz) * invsum32;
            dm3[1] += (-dirx * diry * ddx + (sum2 - diry * diry) * ddy - dirz * diry * ddz) * invsum32;
            dm3[2] += (-dirx * dirz * ddx - diry * dirz * ddy + (sum2 - dirz * dirz) * ddz) * invsum32;
        }
        const float depth = T[8];
        dm2x = hack_x * depth * 0.5f * (float)W;
        dm2y = hack_y * depth * 0.5f * (float)H;
    } else if (shs && !sh_split) {
        float *dsh = dshs + (size_t)i * M * 3;
        for (int k = 0; k < M * 3; k++) dsh[k] = 0.f;
    }

    dmeans2D[3 * i + 0] = dm2x; dmeans2D[3 * i + 1] = dm2y; dmeans2D[3 * i + 2] = 0.f;
    dopacities[i] = vis ? rec[12] : 0.f;
    if (dmeans3D) { dmeans3D[3 * i + 0] = dm3[0]; dmeans3D[3 * i + 1] = dm3[1]; dmeans3D[3 * i + 2] = dm3[2]; }
    if (!shs) {
        for (int c = 0; c < C; c++) dcolors[(size_t)i * C + c] = vis ? rec[15 + c] : 0.f;
    }
    if (transmat_precomp) {
#pragma unroll
        for (int c = 0; c < 9; c++) dtransmat_precomp[9 * i + c] = dT[c];
    } else {
        dscales[2 * i + 0] = dsc0; dscales[2 * i + 1] = dsc1;
        drots[4 * i + 0] = dq[0]; drots[4 * i + 1] = dq[1]; drots[4 * i + 2] = dq[2]; drots[4 * i + 3] = dq[3];
    }
}

int launch_project_bwd(const envgs_raster_cfg *cfg, const float *geom, const float *means3D, const float *scales,
                       const float *rotations, const float *shs, const uint8_t *clamped, const float *transmat_precomp,
                       const int32_t *radii, const float *viewmatrix, const float *projmatrix, const float *campos,
                       const float *grad_rec, float *dmeans3D, float *dmeans2D, float *dscales, float *drots, float *dshs,
                       float *dcolors, float *dopacities, float *dtransmat_precomp, hipStream_t stream)
{
    const int P = cfg->P;
    if (P <= 0) return 0;
    ProfScope prof_(K_PROJECT_BWD, stream);
    // 16 fp32 SH coefficients per surfel (the usual layout): the SH part is split off into a kernel with four lanes per surfel
    const int sh_split = (shs && cfg->sh_coeffs == 16 && !cfg->feature_f16 && dmeans3D && dshs && clamped && campos) ? 1 : 0;
    hipLaunchKernelGGL(project_surfels_bwd, dim3((P + 255) / 256), dim3(256), 0, stream, P, cfg->sh_degree, cfg->sh_coeffs, cfg->feature_f16,
                       cfg->channels, cfg->width, cfg->height, cfg->scale_modifier, geom, means3D, scales, rotations, shs,
                       clamped, transmat_precomp, radii, viewmatrix, projmatrix, campos, grad_rec, dmeans3D, dmeans2D,
                       dscales, drots, dshs, dcolors, dopacities, dtransmat_precomp, sh_split);
    ENVGS_CHECK_LAUNCH(cfg, stream);
    if (sh_split) {
        const int rc = launch_sh_record_bwd(P, cfg->sh_degree, means3D, shs, campos, clamped, radii, grad_rec, dmeans3D, dshs, stream);
        if (rc) return rc;
        ENVGS_CHECK_LAUNCH(cfg, stream);
    }
    return 0;
}

}  // namespace envgs
