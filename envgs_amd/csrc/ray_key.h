// ray_key.h -- coherence key of a ray (the order in which the tracer's 64-ray batches are formed).
#pragma once
#include <hip/hip_runtime.h>

namespace envgs {

constexpr int RAY_KEY_LEAD = 4;      // direction-only rounds in front of the interleaved ones (see below); envgs_debug_set(ENVGS_DBG_RAYKEY, lead + 1) overrides

// 31-bit key: octahedral direction (u, v: 8 bits each) and origin cell (x, y, z: 5 bits each) INSIDE THE BOUNDING BOX OF THE RAY ORIGINS,
// Morton-interleaved from the top: `lead` rounds of (u, v) alone, then rounds of (u, v, x, y, z) until the origin bits are used up, then the
// remaining direction bits.  Rounds 1-3 normalised the origin by the SCENE box and put all 16 direction bits first: the reflected rays of
// EnvGS leave a +-1.3 object inside a +-50 environment, so every origin fell into one cell and a 64-ray batch -- one direction cell -- held
// rays from all over the object: a bundle 2.6 units wide against surfels of radius ~3, each surfel met by ~25 of the batch's 64 rays.
// A batch should be a THIN bundle at the distances where its rays hit things: its width there is (origin spread) + t x (direction spread),
// and with ~10 k batches over a 4-dimensional ray space both factors matter -- hence the interleaving.  Measured on the bench scene (MI355X,
// round 4, ms per training step; lead = 8 is the old direction-major layout with the origin bits made meaningful):
//   lead 8: 9.53   5: 9.03   4: 8.96   3: 9.23   2: 9.72   1: 10.64   0: 12.10      entries per step 2.37 M (lead 8) -> 2.09 M (lead 4)
__device__ __forceinline__ unsigned ray_coherence_key(int r, const float *__restrict__ ray_o, const float *__restrict__ ray_d,
                                                      const float lo0, const float lo1, const float lo2, const float s0, const float s1, const float s2,
                                                      const int lead)
{
    const float dx = ray_d[3 * r], dy = ray_d[3 * r + 1], dz = ray_d[3 * r + 2];
    const float inv = 1.0f / (fabsf(dx) + fabsf(dy) + fabsf(dz) + 1e-30f);
    float u = dx * inv, v = dy * inv;
    if (dz < 0.f) { const float uu = (1.f - fabsf(v)) * (u >= 0.f ? 1.f : -1.f), vv = (1.f - fabsf(u)) * (v >= 0.f ? 1.f : -1.f); u = uu; v = vv; }
    const unsigned qu = (unsigned)fminf(fmaxf((u * 0.5f + 0.5f) * 256.f, 0.f), 255.f), qv = (unsigned)fminf(fmaxf((v * 0.5f + 0.5f) * 256.f, 0.f), 255.f);
    // origin cell: (o - lo) * (32 / extent), s = 0 for a degenerate extent (camera rays: one origin)
    const unsigned qx = (unsigned)fminf(fmaxf((ray_o[3 * r] - lo0) * s0, 0.f), 31.f);
    const unsigned qy = (unsigned)fminf(fmaxf((ray_o[3 * r + 1] - lo1) * s1, 0.f), 31.f);
    const unsigned qz = (unsigned)fminf(fmaxf((ray_o[3 * r + 2] - lo2) * s2, 0.f), 31.f);
    unsigned key = 0;
    int di = 7, oi = 4;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        if (di >= 0) { key = (key << 2) | (((qv >> di) & 1u) << 1) | ((qu >> di) & 1u); di--; }
        if (k >= lead && oi >= 0) { key = (key << 3) | (((qz >> oi) & 1u) << 2) | (((qy >> oi) & 1u) << 1) | ((qx >> oi) & 1u); oi--; }
    }
    return key;
}

}  // namespace envgs
