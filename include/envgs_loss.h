/*
 * envgs_loss.h -- C-ABI of the fused image loss (fourth "next" row).
 *
 * loss = w_l1 * mean|x - y| + w_ssim * (1 - ssim(x, y)): the image supervision of EnvGS (configs/models/envgs.yaml:70-72 -> L1 0.8, SSIM 0.2;
 * easyvolcap/models/supervisors/volumetric_video_supervisor.py:40-66,112-144).  ssim = easyvolcap/utils/ssim_utils.py:58-167 as called by
 * easyvolcap/utils/loss_utils.py:547-549: 11-tap sigma-1.5 separable Gaussian built in float32, padding='same' (zeros), data_range 1,
 * K = (0.01, 0.03), mean over all channels and pixels.  The reference runs it as 5 x 2 grouped conv2d plus ~20 elementwise kernels and lets
 * autograd replay them; here one kernel produces the SSIM map statistics (and the three derivative maps the gradient needs) from LDS tiles,
 * and one kernel turns those into dL/dx.  x, y: (C, H, W) fp32 contiguous, H, W >= 11 (the reference skips SSIM below that,
 * volumetric_video_supervisor.py:70).
 */
#ifndef ENVGS_LOSS_H
#define ENVGS_LOSS_H

#include <stddef.h>
#include <stdint.h>

#include "envgs_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Number of (ssim_sum, abs_sum) float pairs envgs_l1_ssim_forward writes (one per 16x16 tile and channel). */
ENVGS_API int64_t envgs_l1_ssim_partial_count(int32_t C, int32_t H, int32_t W);

/* partial: (count, 2) floats out -- sum of the SSIM map and of |x - y| over each tile (the caller adds them up, in double if it likes).
 * maps: (3, C, H, W) floats out, or NULL when no gradient is wanted: d ssim_map / d mu_x (total), / d E[x^2], / d E[xy]. */
ENVGS_API int envgs_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float *x, const float *y, float *maps, float *partial, void *stream);

/* dx = grad_out * ( w_l1 * sign(x - y) - w_ssim * (blur(m0) + 2 x blur(m1) + y blur(m2)) ) / (C H W).  grad_out: device scalar. */
ENVGS_API int envgs_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float *x, const float *y, const float *maps, const float *grad_out,
                                     float w_l1, float w_ssim, float *dx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVGS_LOSS_H */
