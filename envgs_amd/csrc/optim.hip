// optim.hip -- sparse fused multi-tensor Adam (include/envgs_optim.h); restates easyvolcap/utils/src/fused_adam.cu:4-32.
#include "common.h"

#include "../../include/envgs_optim.h"

namespace envgs {

struct AdamBatch {
    envgs_adam_tensor t[ENVGS_ADAM_MAX_TENSORS];
    long long chunk_start[ENVGS_ADAM_MAX_TENSORS + 1];     // prefix of 1024-element chunks
    int count;
};

__device__ __forceinline__ void adam_one(float &p, const float g, float &m, float &v, const float beta1, const float beta2,
                                         const float step_size, const float bc2_sqrt, const float eps)
{
    // exp_avg = exp_avg * beta1 + (1.0 - beta1) * grad      (double intermediates: the CUDA literals are double)
    m = (float)((double)(m * beta1) + (1.0 - (double)beta1) * (double)g);
    v = (float)((double)(v * beta2) + (1.0 - (double)beta2) * (double)g * (double)g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p -= (m / denom) * step_size;
}

__global__ void __launch_bounds__(256)
fused_adam_multi(const AdamBatch B, const float beta1, const float beta2, const float eps)
{
    // chunk -> tensor (<= 24 tensors: a linear scan of wave-uniform values)
    const long long chunk = blockIdx.x;
    int ti = 0;
    while (ti + 1 < B.count && chunk >= B.chunk_start[ti + 1]) ti++;
    const envgs_adam_tensor T = B.t[ti];
    const long long base = (chunk - B.chunk_start[ti]) * 1024 + (long long)threadIdx.x * 4;
    if (base >= T.numel) return;
    const float bc1 = (float)(1.0 - (double)powf(beta1, T.step));
    const float bc2 = (float)(1.0 - (double)powf(beta2, T.step));
    const float step_size = T.lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    if (base + 4 <= T.numel && ((((uintptr_t)(T.grad + base)) | ((uintptr_t)(T.param + base)) | ((uintptr_t)(T.exp_avg + base)) | ((uintptr_t)(T.exp_avg_sq + base))) & 15) == 0) {
        const float4 g = *reinterpret_cast<const float4 *>(T.grad + base);
        if (g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f) return;          // untouched quad: 16 B read, nothing else
        float4 p = *reinterpret_cast<float4 *>(T.param + base), m = *reinterpret_cast<float4 *>(T.exp_avg + base),
               v = *reinterpret_cast<float4 *>(T.exp_avg_sq + base);
        if (g.x != 0.f) adam_one(p.x, g.x, m.x, v.x, beta1, beta2, step_size, bc2_sqrt, eps);
        if (g.y != 0.f) adam_one(p.y, g.y, m.y, v.y, beta1, beta2, step_size, bc2_sqrt, eps);
        if (g.z != 0.f) adam_one(p.z, g.z, m.z, v.z, beta1, beta2, step_size, bc2_sqrt, eps);
        if (g.w != 0.f) adam_one(p.w, g.w, m.w, v.w, beta1, beta2, step_size, bc2_sqrt, eps);
        *reinterpret_cast<float4 *>(T.param + base) = p;
        *reinterpret_cast<float4 *>(T.exp_avg + base) = m;
        *reinterpret_cast<float4 *>(T.exp_avg_sq + base) = v;
    } else {
        for (long long i = base; i < base + 4 && i < T.numel; i++) {
            const float g = T.grad[i];
            if (g != 0.f) {
                float p = T.param[i], m = T.exp_avg[i], v = T.exp_avg_sq[i];
                adam_one(p, g, m, v, beta1, beta2, step_size, bc2_sqrt, eps);
                T.param[i] = p; T.exp_avg[i] = m; T.exp_avg_sq[i] = v;
            }
        }
    }
}

}  // namespace envgs

using namespace envgs;

extern "C" int envgs_fused_adam(int32_t count, const envgs_adam_tensor *tensors, float beta1, float beta2, float eps, void *stream)
{
    if (count < 0 || count > ENVGS_ADAM_MAX_TENSORS || (count > 0 && !tensors)) return ENVGS_ERR_BAD_ARG;
    AdamBatch B;
    B.count = 0;
    long long chunks = 0;
    for (int i = 0; i < count; i++) {
        if (tensors[i].numel <= 0) continue;
        if (!tensors[i].param || !tensors[i].grad || !tensors[i].exp_avg || !tensors[i].exp_avg_sq) return ENVGS_ERR_BAD_ARG;
        B.t[B.count] = tensors[i];
        B.chunk_start[B.count] = chunks;
        chunks += (tensors[i].numel + 1023) / 1024;
        B.count++;
    }
    B.chunk_start[B.count] = chunks;
    if (chunks == 0) return 0;
    ProfScope prof_(K_FUSED_ADAM, (hipStream_t)stream);
    hipLaunchKernelGGL(fused_adam_multi, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, B, beta1, beta2, eps);
    return (int)hipGetLastError();
}
