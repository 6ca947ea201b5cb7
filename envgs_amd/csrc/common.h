// common.h -- shared device helpers for the gfx950 surfel kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/envgs_raster.h"
#include "prof.h"

namespace envgs {

constexpr int TILE = ENVGS_TILE;              // 16x16 pixel tiles (binning granularity of the reference)
constexpr int GEOM = ENVGS_GEOM_STRIDE;       // floats per surfel record
constexpr int GREC = ENVGS_GRAD_STRIDE;       // floats per gradient record
constexpr float NEAR_N = 0.2f;
constexpr float FAR_N = 100.0f;
constexpr float FILTER_SIZE = 0.707106f;
constexpr float FILTER_INV_SQ = 2.0f;
constexpr float ALPHA_CAP = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_EPS = 0.0001f;

// Per-surfel FEATURE arrays (SH coefficients, precomputed colours) may be stored in fp32 or fp16 (cfg->feature_f16); they are converted on
// load and everything downstream is fp32.  `h` is wave-uniform.
struct Feat {
    const void *p;
    bool h;
    __device__ __forceinline__ float operator[](size_t i) const {
        return h ? __half2float(reinterpret_cast<const __half *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
    }
    __device__ __forceinline__ Feat at(size_t i) const { return Feat{reinterpret_cast<const char *>(p) + i * (h ? 2 : 4), h}; }
};

#define ENVGS_CHECK_LAUNCH(cfg, stream)                                             \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) return (int)e__;                                     \
        if ((cfg)->debug) {                                                         \
            e__ = hipStreamSynchronize((hipStream_t)(stream));                      \
            if (e__ != hipSuccess) return (int)e__;                                 \
        }                                                                           \
    } while (0)

// ---- wave64 cross-lane helpers (DPP; no LDS traffic) -------------------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND));
}

// Sum over the 64 lanes; the result is returned wave-uniform (in an SGPR via readlane 63).
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);                 // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_mov<0x4E>(v);                 // quad_perm [2,3,0,1]  (xor 2)
    v += dpp_mov<0x141>(v);                // row_half_mirror      (xor 4 once quads are uniform)
    v += dpp_mov<0x140>(v);                // row_mirror           (xor 8 once octets are uniform)
    v += dpp_mov<0x142, 0xa, 0xf, false>(v);   // row_bcast15 into rows 1,3
    v += dpp_mov<0x143, 0xc, 0xf, false>(v);   // row_bcast31 into rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}


typedef unsigned envgs_u2 __attribute__((ext_vector_type(2)));

// Sum 64 per-lane values over the wavefront, value v ending in lane v (a full 64 x 64 transpose-reduce in ~140 instructions):
// v_permlane32_swap and v_permlane16_swap fold the four 16-lane rows together two values per instruction (row r then owns values
// 16r..16r+15), and inside a row a butterfly halves the value count per step -- each lane keeps the half its own lane-id bit selects and
// receives the partner's contribution to that half through DPP (row_mirror, row_half_mirror, quad reversals pair lanes across bit 3,2,1,0).
__device__ __forceinline__ float wave_reduce64(const float (&g)[64], const int lane)
{
    float z[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const envgs_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(g[i]), __float_as_uint(g[i + 32]), false, false);
        z[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const envgs_u2 q = __builtin_amdgcn_permlane16_swap(__float_as_uint(z[k]), __float_as_uint(z[k + 16]), false, false);
        w[k] = __uint_as_float(q.x) + __uint_as_float(q.y);
    }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = (b3 ? w[k + 8] : w[k]) + dpp_mov<0x140>(b3 ? w[k] : w[k + 8]);       // partner lane ^ 15
    float b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) b[k] = (b2 ? a[k + 4] : a[k]) + dpp_mov<0x141>(b2 ? a[k] : a[k + 4]);       // partner lane ^ 7
    float c[2];
#pragma unroll
    for (int k = 0; k < 2; k++) c[k] = (b1 ? b[k + 2] : b[k]) + dpp_mov<0x1B>(b1 ? b[k] : b[k + 2]);        // partner lane ^ 3
    return (b0 ? c[1] : c[0]) + dpp_mov<0xB1>(b0 ? c[0] : c[1]);                                            // partner lane ^ 1
}

// Inclusive scans over the 64 lanes (DPP only: row_shr 1,2,4,8 inside each 16-lane row, then row_bcast15 / row_bcast31 carry the row
// totals forward).  `IDENT` fills lanes whose source falls outside the row / the selected rows.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_fill(float v, float ident) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_scan_add(float v) {
    v += dpp_fill<0x111>(v, 0.f); v += dpp_fill<0x112>(v, 0.f); v += dpp_fill<0x114>(v, 0.f); v += dpp_fill<0x118>(v, 0.f);
    v += dpp_fill<0x142, 0xa>(v, 0.f);
    v += dpp_fill<0x143, 0xc>(v, 0.f);
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v) {
    v *= dpp_fill<0x111>(v, 1.f); v *= dpp_fill<0x112>(v, 1.f); v *= dpp_fill<0x114>(v, 1.f); v *= dpp_fill<0x118>(v, 1.f);
    v *= dpp_fill<0x142, 0xa>(v, 1.f);
    v *= dpp_fill<0x143, 0xc>(v, 1.f);
    return v;
}
__device__ __forceinline__ float wave_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// Transpose-reduce of 4*N4 per-lane values over the wavefront in ~(2+1+1)*N4 + ... instructions instead of 7 per value:
//   v_permlane32_swap pairs value i with value i+2*N4 (one add leaves i's 32 partial sums in the low half, the other's in the high half),
//   v_permlane16_swap pairs again (each 16-lane row now owns ONE value), four DPP row rotations finish the row sums, and N4-1 selects
//   merge the registers.  Result: lane r*16+k (k < N4) returns the sum over all 64 lanes of value k + r*N4.
template <int N4>
__device__ __forceinline__ float wave_transpose_reduce(const float (&g)[4 * N4], const int lane)
{
    float z[2 * N4];
#pragma unroll
    for (int i = 0; i < 2 * N4; i++) {
        const envgs_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(g[i]), __float_as_uint(g[i + 2 * N4]), false, false);
        z[i] = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    float w[N4];
#pragma unroll
    for (int k = 0; k < N4; k++) {
        const envgs_u2 q = __builtin_amdgcn_permlane16_swap(__float_as_uint(z[k]), __float_as_uint(z[k + N4]), false, false);
        float v = __uint_as_float(q.x) + __uint_as_float(q.y);
        v += dpp_mov<0x128>(v);                // row_ror:8
        v += dpp_mov<0x124>(v);                // row_ror:4
        v += dpp_mov<0x122>(v);                // row_ror:2
        v += dpp_mov<0x121>(v);                // row_ror:1
        w[k] = v;
    }
    float out = w[0];
#pragma unroll
    for (int k = 1; k < N4; k++) out = ((lane & 15) == k) ? w[k] : out;
    return out;
}

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ void atomic_add_f32(float *p, float v) { unsafeAtomicAdd(p, v); }

// ---- launchers implemented in the kernel translation units ---------------------------------------
int launch_project(const envgs_raster_cfg *cfg, const float *means3D, const float *scales, const float *rotations,
                   const float *opacities, const float *shs, const float *transmat_precomp, const float *viewmatrix,
                   const float *projmatrix, const float *campos, float *geom, float *rgb, uint8_t *clamped,
                   int32_t *radii, uint32_t *tiles_touched, hipStream_t stream);
int launch_scan(const uint32_t *in, uint32_t *out, int n, void *temp, size_t temp_bytes, hipStream_t stream);
size_t scan_temp_bytes(int n);
size_t sort_temp_bytes(uint32_t n, int width, int height);
int launch_bin(const envgs_raster_cfg *cfg, uint32_t N, const float *geom, const int32_t *radii, uint64_t *tile_pairs,
               uint64_t *keys_sorted, uint32_t *point_list, void *bin_temp, size_t bin_temp_bytes, uint32_t *ranges, hipStream_t stream);
size_t key_sort_temp_bytes(int n);
int launch_key_sort(int n, const uint64_t *keys_in, uint64_t *keys_out, int top_bit, void *temp, size_t temp_bytes, hipStream_t stream);
size_t ray_sort_temp_bytes(int R);
int launch_ray_sort(int R, const float *ray_o, const float *ray_d, const float4 *nodes, int P, uint64_t *pairs, uint32_t *order, void *temp,
                    size_t temp_bytes, hipStream_t stream);
int launch_render_fwd(const envgs_raster_cfg *cfg, const uint32_t *ranges, const uint32_t *point_list, const float *geom,
                      const float *colors, const float *bg, float *out_color, float *allmap, float *final_T,
                      int32_t *n_contrib, float *weight, hipStream_t stream, uint8_t *audit_contrib = nullptr, int audit_lmax = 0, int colors_f16 = 0,
                      uint8_t *contrib_mask = nullptr, const uint8_t *audit_skip = nullptr);
int launch_render_bwd(const envgs_raster_cfg *cfg, const uint32_t *ranges, const uint32_t *point_list, const float *geom,
                      const float *colors, const float *bg, const float *final_T, const int32_t *n_contrib,
                      const float *dL_dcolor, const float *dL_dallmap, float *grad_rec, hipStream_t stream, int colors_f16 = 0,
                      const uint8_t *contrib_mask = nullptr);
int launch_project_bwd(const envgs_raster_cfg *cfg, const float *geom, const float *means3D, const float *scales,
                       const float *rotations, const float *shs, const uint8_t *clamped, const float *transmat_precomp,
                       const int32_t *radii, const float *viewmatrix, const float *projmatrix, const float *campos,
                       const float *grad_rec, float *dmeans3D, float *dmeans2D, float *dscales, float *drots, float *dshs,
                       float *dcolors, float *dopacities, float *dtransmat_precomp, hipStream_t stream);

int launch_sh_record_bwd(int P, int D, const float *means3D, const float *shs, const float *campos, const uint8_t *clamped, const int32_t *radii,
                         const float *grad_rec, float *dmeans3D, float *dshs, hipStream_t stream);

__host__ __device__ inline int tile_bits(int width, int height) {
    int tiles = ((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    int b = 0;
    while ((1 << b) < tiles) b++;
    return b < 1 ? 1 : b;
}

}  // namespace envgs
