/*
 * envgs_glue.h -- C-ABI of the fused caller-side glue (the first "next" row after the two extensions).
 *
 * Between its two extension calls the reference runs ~25 full-image and ~100 per-surfel torch kernels:
 *   - SH -> 5/7-channel colours in Python when pipe.convert_SHs_python is set (EnvGS base pass):
 *       easyvolcap/utils/gaussian2d_utils.py:1071-1084 (+ eval_sh, easyvolcap/utils/sh_utils.py:642-727)
 *   - reflected-ray construction from the base pass' maps:
 *       gaussian2d_utils.py:1119-1136 (normal view->world, depth = expected/alpha mixed with the median by depth_ratio),
 *       easyvolcap/models/samplers/envgs_sampler.py:420-431 (ref_d = d - 2(d.n)n, ref_o = o + d*depth)
 * These entry points compute exactly those expressions (and their gradients) in ONE kernel each.  They are optional: the two
 * extensions do not depend on them, and the unchanged reference loop simply keeps using its torch code.
 * Conventions as in envgs_raster.h (device pointers, hipStream_t as void*, 0 = ok).
 */
#ifndef ENVGS_GLUE_H
#define ENVGS_GLUE_H

#include <stddef.h>
#include <stdint.h>

#include "envgs_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * colours (P, 3+S+1) = [ clamp_min(eval_sh(deg, shs, normalize(means3D - campos)) + 0.5, 0), specular (P,S), roughness (P,1) ].
 * shs is (P, sh_coeffs, 3) (the get_features layout); clamped (P,3) uint8 is kept for the backward.
 */
ENVGS_API int envgs_sh_colors_forward(int32_t P, int32_t sh_degree, int32_t sh_coeffs, int32_t spec_channels,
                                      const float *means3D, const float *shs, const float *campos, const float *specular,
                                      const float *roughness, float *colors, uint8_t *clamped, void *stream);
ENVGS_API int envgs_sh_colors_backward(int32_t P, int32_t sh_degree, int32_t sh_coeffs, int32_t spec_channels,
                                       const float *means3D, const float *shs, const float *campos, const uint8_t *clamped,
                                       const float *dcolors, float *dmeans3D, float *dshs, float *dspecular, float *droughness,
                                       void *stream);

/*
 * Per pixel, from allmap (7,H,W) [0 depth*alpha, 1 alpha, 2-4 view normal, 5 median depth], rays (H,W,3) and the view matrix
 * (4,4, row-vector convention as everywhere): normal_world (3,H,W), depth (1,H,W), ref_o (H,W,3), ref_d (H,W,3).
 */
ENVGS_API int envgs_reflect_forward(int32_t H, int32_t W, float depth_ratio, const float *allmap, const float *ray_o,
                                    const float *ray_d, const float *viewmatrix, float *normal_world, float *depth,
                                    float *ref_o, float *ref_d, void *stream);
/* dallmap (7,H,W) is WRITTEN (channels 0-5; channel 6 = 0); dray_o / dray_d (H,W,3) may be NULL. */
ENVGS_API int envgs_reflect_backward(int32_t H, int32_t W, float depth_ratio, const float *allmap, const float *ray_o,
                                     const float *ray_d, const float *viewmatrix, const float *dnormal_world, const float *ddepth,
                                     const float *dref_o, const float *dref_d, float *dallmap, float *dray_o, float *dray_d,
                                     void *stream);

/*
 * The regulariser maps of render()'s tail (gaussian2d_utils.py:1125-1142): surf_depth (1,H,W) = expected depth (allmap[0] / allmap[1],
 * nan -> 0) mixed with the median depth (allmap[5]) by depth_ratio, and surf_normal (3,H,W) = dpt2norm(surf_depth) * alpha.detach()
 * (dpt2xyz / dpt2norm, :1158-1206: back-projection through the integer pixel grid with fx = W / (2 tan(FoVx/2)), central differences on
 * interior pixels, zero border).  viewmatrix: world_view_transform (4,4) on the device; its upper 3x3 is the camera-to-world rotation
 * (the reference inverts the matrix; for a rigid camera that is the same).  The backward WRITES dallmap (7,H,W): channels 0, 1, 5; the rest zero.
 */
ENVGS_API int envgs_surface_normal_forward(int32_t H, int32_t W, float depth_ratio, float fx, float fy, const float *allmap,
                                           const float *viewmatrix, float *surf_depth, float *surf_normal, void *stream);
ENVGS_API int envgs_surface_normal_backward(int32_t H, int32_t W, float depth_ratio, float fx, float fy, const float *allmap,
                                            const float *viewmatrix, const float *dsurf_depth, const float *dsurf_normal,
                                            float *dallmap, void *stream);

/*
 * The 3-sigma quads the reference hands to SurfelTracer.build_acceleration_structure (easyvolcap/utils/optix_utils.py:39-69, get_disks):
 * vertices (4P,3) = mu + 3 (s_u a (-+) + s_v b (+-)) in the corner order (-,+) (-,-) (+,+) (+,-), a / b = columns 0 / 1 of the rotation of the
 * NORMALISED quaternion; faces (2P,3) int32 = (4i, 4i+1, 4i+2), (4i+1, 4i+2, 4i+3) (may be NULL).  The reference builds them with ~25 torch
 * kernels (a batched matmul among them) every training step, right in front of the trace; no gradient flows through them.
 */
ENVGS_API int envgs_surfel_quads(int32_t P, const float *means3D, const float *scales, const float *rotations, float *vertices, int32_t *faces,
                                 void *stream);

/*
 * The specular blend that closes an EnvGS forward (easyvolcap/models/samplers/envgs_sampler.py:474 with the channel layout of
 * gaussian2d_utils.py:1119-1144): img (C,H,W) is the -ch05 / -ch07 rasterizer output [rgb 3 | specular S = C-4 | roughness 1], rgb_env (H,W,3)
 * the traced colour;  rgb (H,W,3) = (1 - s) * img[:3] + s * rgb_env  with s = the specular channel (S = 1: shared by the three colours,
 * S = 3: per colour).  The torch form is three slices of `img`, four elementwise kernels and, in the backward, the slices' scatter-adds into a
 * zero image: ~20 launches; here one each way.  dimg (C,H,W) is fully written (zeros in the roughness channel).
 */
ENVGS_API int envgs_blend_forward(int32_t H, int32_t W, int32_t channels, const float *img, const float *rgb_env, float *rgb, void *stream);
ENVGS_API int envgs_blend_backward(int32_t H, int32_t W, int32_t channels, const float *img, const float *rgb_env, const float *drgb,
                                   float *dimg, float *drgb_env, void *stream);

/*
 * Bounce stages of a multi-depth trace (gaussian2d_sampler.py:413-426 / optix_utils.py:117-118; envgs_amd/tracing.py:_forward_bounces):
 * rows `sel` (n UNIQUE int64 row indices, e.g. a nonzero() result) of stage k's per-ray tensors -- rays (R_k,3), dpt / acc (R_k,1), norm (R_k,3),
 * aux (R_k,2), rgb (R_k,3) -- become stage k+1's rays:  n^ = norm/|norm|, t = dpt/acc, o2 = o + d t, d2 = d - 2 (d.n^) n^   (o2, d2: (n,3));
 * and stage k+1's colour col_next (n,3) is blended back:  col[sel] = (1 - s) rgb[sel] + s col_next, s = aux[sel,0]  (col: the caller's COPY of rgb).
 * Backwards: g_ray_o / g_ray_d / g_dpt / g_acc / g_norm / g_aux are (R_k, .) buffers the caller has ZEROED (rows outside sel receive nothing);
 * g_rgb is the caller's COPY of g_col (rows outside sel pass through).  Any gradient pointer may be NULL.  One launch each.
 * envgs_bounce_pack_mid: the 16 `mid` channels [o 3 | d 3 | dpt | acc | norm 3 | aux 2 | rgb 3] of stage k (of `stages`) at rows idx (NULL: row i).
 */
ENVGS_API int envgs_bounce_rays_forward(int32_t n, const int64_t *sel, const float *ray_o, const float *ray_d, const float *dpt, const float *acc,
                                        const float *norm, float *o2, float *d2, void *stream);
ENVGS_API int envgs_bounce_rays_backward(int32_t n, const int64_t *sel, const float *ray_o, const float *ray_d, const float *dpt, const float *acc,
                                         const float *norm, const float *g_o2, const float *g_d2, float *g_ray_o, float *g_ray_d, float *g_dpt,
                                         float *g_acc, float *g_norm, void *stream);
ENVGS_API int envgs_bounce_blend_forward(int32_t n, const int64_t *sel, const float *rgb, const float *aux, const float *col_next, float *col,
                                         void *stream);
ENVGS_API int envgs_bounce_blend_backward(int32_t n, const int64_t *sel, const float *rgb, const float *aux, const float *col_next,
                                          const float *g_col, float *g_rgb, float *g_aux, float *g_col_next, void *stream);
ENVGS_API int envgs_bounce_pack_mid(int32_t n, const int64_t *idx, int32_t stages, int32_t k, const float *ray_o, const float *ray_d,
                                    const float *dpt, const float *acc, const float *norm, const float *aux, const float *rgb, float *mid,
                                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVGS_GLUE_H */
