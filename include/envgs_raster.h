/*
 * envgs_raster.h -- C-ABI of the MI355X-native surfel rasterizer (libenvgs_hip.so).
 *
 * This is the drop-in boundary for the reference's three pybind/torch extensions
 *   diff_surfel_rasterization_wet, _wet_ch05, _wet_ch07
 * whose Python call site is easyvolcap/utils/gaussian2d_utils.py:1025-1038 (settings) and
 * :1089-1099 (call).  The reference has no C layer of its own to mirror (SURVEY.md section 8b:
 * "the FFI is pybind inside each package"), so each entry point below names the reference
 * interface it stands behind.  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * (HBM) unless its name ends in _host.  `stream` is a hipStream_t passed as void*.
 * All functions return 0 on success, or a negative envgs_status / positive hipError_t.
 *
 * Memory is owned by the caller (torch allocates it; the library never calls hipMalloc), so the
 * same buffers can be reused across calls and P / image size may change from call to call
 * (densification re-allocates every parameter: SURVEY.md section 3.6).
 *
 * HBM layouts (fp32 unless noted):
 *   geom      (P,16)  per-surfel record, 64 B aligned:
 *                     [0..2] Tu  [3..5] Tv  [6..8] Tw   (transMat rows: coefficients of u, v, 1)
 *                     [9..10] screen centre x,y  [11..13] view-space normal (camera facing)
 *                     [14] opacity  [15] view depth (sort key)
 *   colors    (P,C)   the caller's colors_precomp (fp32, or half when cfg->feature_f16), or `rgb` (P,3) fp32 produced from SH by _project
 *   grad_rec  (P,32)  per-surfel gradient record, 128 B aligned, accumulated by _backward:
 *                     [0..8] dL/dtransMat  [9..11] dL/dnormal  [12] dL/dopacity  [13..14] dL/dmean2D
 *                     [15..15+C) dL/dcolour
 *   final_T   (3,H,W) T, M1, M2      n_contrib (2,H,W) int32: last contributor, median contributor
 *   allmap    (7,H,W) 0 depth, 1 alpha, 2-4 normal, 5 median depth, 6 distortion
 *                     (order read by gaussian2d_utils.py:1119-1144)
 */
#ifndef ENVGS_RASTER_H
#define ENVGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define ENVGS_API __attribute__((visibility("default")))
#else
#define ENVGS_API
#endif

#define ENVGS_TILE 16
#define ENVGS_GEOM_STRIDE 16
#define ENVGS_GRAD_STRIDE 32

typedef enum envgs_status {
    ENVGS_OK = 0,
    ENVGS_ERR_BAD_ARG = -1,      /* unsupported channel count / SH degree / null pointer */
    ENVGS_ERR_TEMP_TOO_SMALL = -2
} envgs_status;

/* Mirrors GaussianRasterizationSettings (gaussian2d_utils.py:1025-1038) minus the tensors. */
typedef struct envgs_raster_cfg {
    int32_t P;              /* number of surfels */
    int32_t sh_degree;      /* active degree 0..3 (settings.sh_degree) */
    int32_t sh_coeffs;      /* coefficients stored per surfel, shs is (P, sh_coeffs, 3); 0 if colours are precomputed */
    int32_t channels;       /* 3 (-wet) / 5 (-wet-ch05) / 7 (-wet-ch07) */
    int32_t width, height;  /* settings.image_width / image_height */
    int32_t bg_len;         /* entries in bg; channels >= bg_len composite against 0 */
    int32_t debug;          /* settings.debug: synchronise + check after every kernel */
    float scale_modifier;   /* settings.scale_modifier */
    float tanfovx, tanfovy; /* carried for API completeness; the projection uses projmatrix */
    int32_t feature_f16;    /* 1: the per-surfel FEATURE array the caller passes (shs, or colors_precomp when shs == NULL) is stored as IEEE half
                               (same shape); converted on load, all arithmetic, accumulation and every gradient output stay fp32.  The storage
                               variant of BASELINE configs[4]; the reference has no counterpart. */
} envgs_raster_cfg;

/* Bytes of scratch the prefix sum over P counters needs, and the binning of one width x height image (N is not used any more: the scratch
 * is per-tile histograms, see _bin_and_render). */
ENVGS_API size_t envgs_raster_scan_temp_bytes(int32_t P);
ENVGS_API size_t envgs_raster_sort_temp_bytes(uint32_t N, int32_t width, int32_t height);

/*
 * Stage R1+R2 (GaussianRasterizer.forward, first half): per-surfel projection + Jacobian (transMat),
 * view normal, 3-sigma AABB -> radius / tile rect, SH -> RGB; then the inclusive scan of tiles_touched.
 * Exactly one of (scales, rotations) / transmat_precomp and one of shs / colours is used:
 *   shs == NULL  -> colours are the caller's colors_precomp (not touched here)
 *   shs != NULL  -> rgb (P,3) and clamped (P,3) uint8 are written
 * Writes the number of tile instances N to *num_rendered_host after synchronising `stream` (the caller sizes the binning buffers with
 * it) -- or, with num_rendered_host == NULL, returns without a host sync: N is then offsets[P-1] on the device (see _bin_and_render).
 */
ENVGS_API int envgs_raster_project(const envgs_raster_cfg *cfg,
                         const float *means3D, const float *scales, const float *rotations,
                         const float *opacities, const float *shs, const float *transmat_precomp,
                         const float *viewmatrix, const float *projmatrix, const float *campos,
                         float *geom, float *rgb, uint8_t *clamped, int32_t *radii,
                         uint32_t *tiles_touched, uint32_t *offsets,
                         void *scan_temp, size_t scan_temp_bytes,
                         uint32_t *num_rendered_host, void *stream);

/*
 * Stages R3-R6 (GaussianRasterizer.forward, second half).  The reference emits (tile id << 32 | depth bits, surfel id) pairs, radix-sorts
 * all N of them (stable) and detects the per-tile ranges; what that fixes is, per tile, its instances in (depth bits, surfel id) order.
 * Here the instances are counted per tile in LDS histograms, scattered into their tile's segment (tile_pairs: depth bits << 32 | id) and
 * each segment is sorted by one workgroup in LDS -- no device-wide sort (csrc/raster_bin.hip).  Then front-to-back compositing of
 * `channels` colours + the 7 allmap channels + the per-surfel accumulated weight (the "-wet" output, gaussian2d_utils.py:1090,1114).
 * point_list / ranges are outputs the backward pass reads; keys_sorted (tile id << 32 | depth bits per list entry, what the reference's
 * sorted key buffer holds) is written only when non-NULL (parity tests).
 * N is the CAPACITY of the N-sized buffers (tile_pairs, keys_sorted, point_list, contrib_mask): the instance count envgs_raster_project
 * returned, or -- when the caller did not wait for it (num_rendered_host == NULL there) -- a guess.  The count is re-derived on the device;
 * if it exceeds N every range is left empty (nothing is written out of bounds) and the caller repeats the call with the exact size.
 * contrib_mask (N bytes, optional): for every tile instance (= entry of point_list) the set of 8x8 pixel quadrants of its tile in which
 * some pixel blended it (bit q = quadrant q; row-major 2x2).  Passed to envgs_raster_backward it lets the backward visit exactly the
 * (quadrant, entry) pairs the forward blended instead of re-deriving them geometrically (most candidates fail the alpha test everywhere).
 */
ENVGS_API int envgs_raster_bin_and_render(const envgs_raster_cfg *cfg, uint32_t N,
                                const float *geom, const int32_t *radii,
                                const float *colors, const float *bg,
                                uint64_t *tile_pairs, uint64_t *keys_sorted /* may be NULL */, uint32_t *point_list,
                                void *bin_temp, size_t bin_temp_bytes, uint32_t *ranges,
                                float *out_color, float *allmap, float *final_T, int32_t *n_contrib,
                                float *weight, uint8_t *contrib_mask, void *stream);

/*
 * Parity audit of stage R6 (tests only; no reference counterpart): the SAME compositing kernel, instantiated with one extra store --
 * contrib (H*W, lmax) uint8, contrib[pixel][k] = 1 when entry k of the pixel's tile list was blended into it -- so the per-pixel
 * contributor SETS can be compared bit-exactly with the oracle's.  Inputs are the outputs of _project / _bin_and_render; the image
 * outputs are written again (to caller-provided scratch).  lmax >= the longest tile list.
 * skip_px (H*W bytes, optional): pixels marked non-zero are left out of the per-surfel `weight` sums written by this call (and of nothing
 * else) -- the oracle's fragile pixels, so that `weight` is comparable on every surfel, not only on those that touch no fragile pixel.
 */
ENVGS_API int envgs_raster_render_audit(const envgs_raster_cfg *cfg, const float *geom, const float *colors, const float *bg,
                              const uint32_t *point_list, const uint32_t *ranges,
                              float *out_color, float *allmap, float *final_T, int32_t *n_contrib, float *weight,
                              uint8_t *contrib, int32_t lmax, const uint8_t *skip_px, void *stream);

/*
 * Stages R7+R8 (GaussianRasterizer backward): back-to-front gradient of the compositing, reduced per
 * wavefront into grad_rec, then chained to the parameters.  Output pointers that do not apply may be NULL:
 *   dshs (P,sh_coeffs,3) when shs != NULL, else dcolors (P,C);
 *   dscales/drots/dmeans3D when transmat_precomp == NULL, else dtransmat_precomp (P,9).
 * dmeans2D (P,3) receives the densification proxy read by gaussian2d_utils.py:901-909.
 * dL_dcolor (C,H,W) / dL_dallmap (7,H,W): the upstream gradients; either may be NULL = that output is not used by the loss (zero gradient;
 * the autograd node passes undefined gradients through as NULL instead of materialising a buffer of zeros).
 */
ENVGS_API int envgs_raster_backward(const envgs_raster_cfg *cfg, uint32_t N,
                          const float *geom, const float *colors, const float *bg,
                          const uint32_t *point_list, const uint32_t *ranges,
                          const float *final_T, const int32_t *n_contrib, const uint8_t *contrib_mask /* may be NULL */,
                          const float *dL_dcolor, const float *dL_dallmap,
                          const float *means3D, const float *scales, const float *rotations,
                          const float *shs, const uint8_t *clamped, const float *transmat_precomp,
                          const int32_t *radii,
                          const float *viewmatrix, const float *projmatrix, const float *campos,
                          float *grad_rec,
                          float *dmeans3D, float *dmeans2D, float *dscales, float *drots,
                          float *dshs, float *dcolors, float *dopacities, float *dtransmat_precomp,
                          void *stream);

/*
 * Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg; the
 * reference's counterpart is its `timer.record` sections, easyvolcap/utils/console_utils.py:615-693).
 * kernel_id: 0 project_surfels, 1 scan, 2 bin_tile_pairs (histograms + scans + scatter), 3 sort_tile_lists, 4 (unused),
 *            5 composite_fwd, 6 composite_bwd, 7 project_surfels_bwd, 8 bvh_build, 9 trace_fwd (whole forward), 10 trace_bwd
 *            (whole backward), 11 collect_hits, 12 sort_composite_fwd, 13 (unused), 14 K-buffer forward, 15 batch_surfel_bwd,
 *            16 K-buffer backward, 17 reduce_surfel_records, 18 register_hits, 19 fused_adam_multi, 20 l1_ssim_fwd, 21 l1_ssim_bwd.  envgs_prof_kernel_name(id) returns "" past the last id.
 * envgs_prof_read synchronises on the recorded events, returns the summed milliseconds and launch count
 * since the last read, and resets the counter.
 * envgs_prof_select(mask): only the scopes whose bit (1 << kernel_id) is set record while the timers are on (default: all).  An event record is
 * a barrier packet on its stream: with all ~30 scopes of an EnvGS step on, the step is 0.17 ms (2.3 %) longer than with none (measured, round 6) --
 * bench.py keeps the two dominant kernels' scopes inside its timed regions and times the others in a separate pass.
 */
ENVGS_API void envgs_prof_enable(int on);
ENVGS_API void envgs_prof_select(uint64_t kernel_mask);
ENVGS_API int envgs_prof_read(int kernel_id, double *total_ms, int *launches);
ENVGS_API const char *envgs_prof_kernel_name(int kernel_id);

/*
 * Diagnostic switches for experiments and tests (scratch/, tests/, bench.py --debug-*): process-global, all 0 in production.
 * The library never reads the environment; whoever sets a switch is responsible for reporting it (bench.py prints them).
 *   ENVGS_DBG_TRACE  bit mask: 8 = atomic-flush tracer backward instead of records, 16 = binary packet traversal instead of the 4-wide one,
 *                    64 = no coherence sort of the rays, 512 = per-ray collection kernel even when the rays are sorted,
 *                    1024 = packet stack limited to 2 entries (forces the stack-overflow hand-off to the K-buffer path; tests only),
 *                    2048 = one wavefront per 64-ray batch (collect_hits_packet4) instead of the cooperative workgroup (collect_hits_coop),
 *                    4096 / 8192 = cooperative collection with DEFERRED exact tests at 8 / 6 wavefronts per SIMD (round-5 A/B, diagnostic library),
 *                    16384 = forward_prepare on the caller's stream after the coherence sort instead of beside it on the second stream (round-6 A/B)
 *   ENVGS_DBG_SEGMENTS  forward batch segments of the tracer (0 = default 2; 1 = single launch)
 */
#define ENVGS_DBG_TRACE 0
#define ENVGS_DBG_SEGMENTS 1
#define ENVGS_DBG_COLLECT_WGS 2      /* workgroups per CU of the cooperative collection's persistent grid (0 = default: 4 for each of two segments in flight, 8 for a single one) */
#define ENVGS_DBG_RASTER_EXACT 3     /* libenvgs_hip_diag.so only: R6 / R7 with IEEE divisions and the library expf instead of v_rcp_f32 (+ Newton) / v_exp_f32 -- the attribution run of the parity tests; no effect in the product library */
#define ENVGS_DBG_RAYKEY 4           /* ray coherence key (csrc/ray_key.h): value - 1 = direction-only rounds in front of the interleaved (direction, origin) rounds; 0 = default */
#define ENVGS_DBG_SPARSE 5           /* sparse entries of the tracer's record backward (envgs_trace.h: sparse_hits): value - 1 = the largest hit count an entry may have to be filed per hit; 0 = default (4), 1 = off */
#define ENVGS_DBG_COUNT 6
ENVGS_API void envgs_debug_set(int32_t which, int32_t value);
ENVGS_API int32_t envgs_debug_get(int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* ENVGS_RASTER_H */
