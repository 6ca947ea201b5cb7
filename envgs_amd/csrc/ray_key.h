// ray_key.h -- coherence key of a ray (the order in which the tracer's 64-ray batches are formed).
#pragma once
#include <hip/hip_runtime.h>

namespace envgs {

// Octahedral direction (2 x 8 bits, Morton-interleaved) in the high 16 bits, origin cell inside the scene box (3 x 5 bits) below: 31 bits.
__device__ __forceinline__ unsigned ray_coherence_key(int r, const float *__restrict__ ray_o, const float *__restrict__ ray_d,
                                                      const float4 *__restrict__ nodes, int P)
{
    float lo[3] = {-1.f, -1.f, -1.f}, ext[3] = {2.f, 2.f, 2.f};
    if (P > 0) {
        const float4 n0 = nodes[0], n1 = nodes[1], n2 = nodes[2];
        lo[0] = fminf(n0.x, n1.z); lo[1] = fminf(n0.y, n1.w); lo[2] = fminf(n0.z, n2.x);
        ext[0] = fmaxf(n0.w, n2.y) - lo[0]; ext[1] = fmaxf(n1.x, n2.z) - lo[1]; ext[2] = fmaxf(n1.y, n2.w) - lo[2];
    }
    const float dx = ray_d[3 * r], dy = ray_d[3 * r + 1], dz = ray_d[3 * r + 2];
    const float inv = 1.0f / (fabsf(dx) + fabsf(dy) + fabsf(dz) + 1e-30f);
    float u = dx * inv, v = dy * inv;
    if (dz < 0.f) { const float uu = (1.f - fabsf(v)) * (u >= 0.f ? 1.f : -1.f), vv = (1.f - fabsf(u)) * (v >= 0.f ? 1.f : -1.f); u = uu; v = vv; }
    const unsigned qu = (unsigned)fminf(fmaxf((u * 0.5f + 0.5f) * 256.f, 0.f), 255.f), qv = (unsigned)fminf(fmaxf((v * 0.5f + 0.5f) * 256.f, 0.f), 255.f);
    unsigned dkey = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) dkey |= ((qu >> b) & 1u) << (2 * b) | ((qv >> b) & 1u) << (2 * b + 1);
    unsigned okey = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = ext[c] > 0.f ? (ray_o[3 * r + c] - lo[c]) / ext[c] : 0.f;
        const unsigned q = (unsigned)fminf(fmaxf(t * 32.f, 0.f), 31.f);
#pragma unroll
        for (int b = 0; b < 5; b++) okey |= ((q >> b) & 1u) << (3 * b + c);
    }
    return (dkey << 15) | okey;
}

}  // namespace envgs
