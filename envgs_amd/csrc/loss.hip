// loss.hip -- fused L1 + SSIM image loss, forward statistics and gradient (include/envgs_loss.h).
#include "common.h"

#include "../../include/envgs_loss.h"

namespace envgs {

constexpr int LT = 16;            // output tile
constexpr int LR = 5;             // window radius (11 taps)
constexpr int LH = LT + 2 * LR;   // tile + halo = 26

// the reference's float32 window (ssim_utils.py:19-25, _fspecial_gauss_1d(11, 1.5)); pinned by tests/golden/loss_golden.npz["win"]
__device__ __constant__ float kWin[11] = {0x1.0d9570p-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c3ep-3f, 0x1.106560p-2f,
                                          0x1.b43c3ep-3f, 0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d9570p-10f};

__global__ void __launch_bounds__(256)
l1_ssim_fwd(const int H, const int W, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ maps, float *__restrict__ partial)
{
    __shared__ float sx[LH][LH + 1], sy[LH][LH + 1];
    __shared__ float hb[5][LH][LT + 1];
    __shared__ float red[2][4];
    const int c = blockIdx.z, ty0 = blockIdx.y * LT, tx0 = blockIdx.x * LT;
    const int tid = threadIdx.x;
    const float *xc = x + (size_t)c * H * W, *yc = y + (size_t)c * H * W;
    for (int i = tid; i < LH * LH; i += 256) {
        const int r = i / LH, q = i - r * LH;
        const int gy = ty0 + r - LR, gx = tx0 + q - LR;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;               // padding='same': zeros outside the image
        sx[r][q] = in ? xc[(size_t)gy * W + gx] : 0.f;
        sy[r][q] = in ? yc[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    // the reference filters along H first, then W (gaussian_filter loops over input.shape[2:]); the filter is separable and the sums are
    // associative only up to rounding, so keep its order: vertical pass into hb, horizontal pass per output pixel
    for (int i = tid; i < LT * LH; i += 256) {
        const int r = i / LH, q = i - r * LH;                                 // r: output row in the tile, q: column incl. halo
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float u = sx[r + k][q], v = sy[r + k][q], w = kWin[k];
            a += w * u; b += w * v; aa += w * (u * u); bb += w * (v * v); ab += w * (u * v);
        }
        hb[0][q][r] = a; hb[1][q][r] = b; hb[2][q][r] = aa; hb[3][q][r] = bb; hb[4][q][r] = ab;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15;
    const int gy = ty0 + ly, gx = tx0 + lx;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float w = kWin[k];
        mu1 += w * hb[0][lx + k][ly]; mu2 += w * hb[1][lx + k][ly]; e11 += w * hb[2][lx + k][ly]; e22 += w * hb[3][lx + k][ly]; e12 += w * hb[4][lx + k][ly];
    }
    float sval = 0.f, aval = 0.f;
    if (gy < H && gx < W) {
        constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
        const float iB1 = 1.0f / B1, iB2 = 1.0f / B2;
        sval = (A1 * iB1) * (A2 * iB2);
        aval = fabsf(sx[ly + LR][lx + LR] - sy[ly + LR][lx + LR]);
        if (maps) {
            const float d_ex2 = -(A1 * A2) * iB1 * iB2 * iB2;
            const float d_exy = 2.f * A1 * iB1 * iB2;
            const float d_mu1 = 2.f * mu2 * A2 * iB1 * iB2 - 2.f * mu1 * A1 * A2 * iB1 * iB1 * iB2 - 2.f * mu1 * d_ex2 - mu2 * d_exy;
            const size_t plane = (size_t)H * W, o = (size_t)c * plane + (size_t)gy * W + gx, cs = (size_t)gridDim.z * plane;
            maps[o] = d_mu1; maps[cs + o] = d_ex2; maps[2 * cs + o] = d_exy;
        }
    }
    const float ssum = wave_sum(sval), asum = wave_sum(aval);
    if ((tid & 63) == 0) { red[0][tid >> 6] = ssum; red[1][tid >> 6] = asum; }
    __syncthreads();
    if (tid == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partial[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256)
l1_ssim_bwd(const int H, const int W, const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ maps,
            const float *__restrict__ grad_out, const float w_l1, const float w_ssim, float *__restrict__ dx)
{
    __shared__ float sm[3][LH][LH + 1];
    __shared__ float hb[3][LH][LT + 1];
    const int c = blockIdx.z, ty0 = blockIdx.y * LT, tx0 = blockIdx.x * LT;
    const int tid = threadIdx.x;
    const size_t plane = (size_t)H * W, cs = (size_t)gridDim.z * plane;
    const float *m0 = maps + (size_t)c * plane;
    for (int i = tid; i < LH * LH; i += 256) {
        const int r = i / LH, q = i - r * LH;
        const int gy = ty0 + r - LR, gx = tx0 + q - LR;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;               // the adjoint of a zero-padded 'same' filter with a symmetric window
        const size_t o = (size_t)gy * W + gx;                                 // is the same zero-padded filter
        sm[0][r][q] = in ? m0[o] : 0.f; sm[1][r][q] = in ? m0[cs + o] : 0.f; sm[2][r][q] = in ? m0[2 * cs + o] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < LT * LH; i += 256) {
        const int r = i / LH, q = i - r * LH;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) { const float w = kWin[k]; a += w * sm[0][r + k][q]; b += w * sm[1][r + k][q]; d += w * sm[2][r + k][q]; }
        hb[0][q][r] = a; hb[1][q][r] = b; hb[2][q][r] = d;
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15;
    const int gy = ty0 + ly, gx = tx0 + lx;
    if (gy >= H || gx >= W) return;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) { const float w = kWin[k]; a += w * hb[0][lx + k][ly]; b += w * hb[1][lx + k][ly]; d += w * hb[2][lx + k][ly]; }
    const size_t o = (size_t)c * plane + (size_t)gy * W + gx;
    const float xv = x[o], yv = y[o];
    const float df = xv - yv;
    const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
    const float inv_n = 1.0f / ((float)gridDim.z * (float)H * (float)W);
    dx[o] = grad_out[0] * (w_l1 * sgn - w_ssim * (a + 2.f * xv * b + yv * d)) * inv_n;
}

}  // namespace envgs

using namespace envgs;

extern "C" {

int64_t envgs_l1_ssim_partial_count(int32_t C, int32_t H, int32_t W)
{
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)C * ((H + LT - 1) / LT) * ((W + LT - 1) / LT);
}

int envgs_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float *x, const float *y, float *maps, float *partial, void *stream)
{
    if (C <= 0 || C > 65535 || H < 11 || W < 11 || !x || !y || !partial) return ENVGS_ERR_BAD_ARG;
    ProfScope prof_(K_LOSS_FWD, (hipStream_t)stream);
    hipLaunchKernelGGL(l1_ssim_fwd, dim3((W + LT - 1) / LT, (H + LT - 1) / LT, C), dim3(256), 0, (hipStream_t)stream, H, W, x, y, maps, partial);
    return (int)hipGetLastError();
}

int envgs_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float *x, const float *y, const float *maps, const float *grad_out,
                           float w_l1, float w_ssim, float *dx, void *stream)
{
    if (C <= 0 || C > 65535 || H < 11 || W < 11 || !x || !y || !maps || !grad_out || !dx) return ENVGS_ERR_BAD_ARG;
    ProfScope prof_(K_LOSS_BWD, (hipStream_t)stream);
    hipLaunchKernelGGL(l1_ssim_bwd, dim3((W + LT - 1) / LT, (H + LT - 1) / LT, C), dim3(256), 0, (hipStream_t)stream, H, W, x, y, maps, grad_out,
                       w_l1, w_ssim, dx);
    return (int)hipGetLastError();
}

}  // extern "C"
