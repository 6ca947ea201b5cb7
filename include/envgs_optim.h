/*
 * envgs_optim.h -- C-ABI of the sparse fused Adam (second "next" row).
 *
 * Counterpart of the reference's only in-tree native kernel, easyvolcap/utils/src/fused_adam.cu:4-32 (`adam_kernel`, launched per
 * tensor by easyvolcap/utils/adam_utils.py and selected through `MyFusedAdam`, easyvolcap/runners/optimizers.py:17-75):
 * elementwise Adam without weight decay that SKIPS every element whose gradient is exactly zero (surfels no pixel / ray touched keep
 * their moments).  Same arithmetic, including the double-precision intermediates the CUDA source's `1.0 - beta` literals imply.
 * MI355X-first differences: ONE launch updates up to ENVGS_ADAM_MAX_TENSORS tensors (the reference launches once per tensor: 13
 * launches per step for the two Gaussian sets), 16 B/lane vector access, zero-gradient quads skip their other three streams.
 * HBM-bound: 28 B per updated element (read p,g,m,v; write p,m,v), 4 B per skipped one.
 */
#ifndef ENVGS_OPTIM_H
#define ENVGS_OPTIM_H

#include <stddef.h>
#include <stdint.h>

#include "envgs_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ENVGS_ADAM_MAX_TENSORS 24

typedef struct envgs_adam_tensor {
    float *param;            /* updated in place */
    const float *grad;
    float *exp_avg;          /* updated in place */
    float *exp_avg_sq;       /* updated in place */
    int64_t numel;
    float lr;
    float step;              /* this tensor's step count AFTER the increment (fused_adam.cu: "already updated") */
} envgs_adam_tensor;

/* Update `count` (<= ENVGS_ADAM_MAX_TENSORS) tensors in one launch.  `tensors` is a HOST array (passed by value to the kernel). */
ENVGS_API int envgs_fused_adam(int32_t count, const envgs_adam_tensor *tensors, float beta1, float beta2, float eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVGS_OPTIM_H */
