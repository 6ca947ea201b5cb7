/*
 * envgs_densify.h -- C-ABI of the densify / prune support kernels (third "next" row).
 *
 * The reference prunes with one boolean-mask gather per tensor -- parameter, exp_avg and exp_avg_sq of each of its 8 parameter groups
 * (`_prune_optimizer`, easyvolcap/utils/gaussian2d_utils.py:536-560; `prune_stats` :640-648): 24+ `tensor[mask]` calls, each its own
 * nonzero + gather + sync.  Here: ONE prefix scan of the keep mask (`envgs_compact_scan`, the caller reads the kept count once) and ONE
 * gather launch that compacts up to ENVGS_COMPACT_MAX_TENSORS row-major tensors of the per-Gaussian SoA (`envgs_compact_gather`).
 * Rows keep their relative order, exactly like `tensor[mask]`.
 *
 * `envgs_knn3_mean_dist2` is the initialisation helper the reference takes from `simple_knn.distCUDA2`
 * (gaussian2d_utils.py:432-440: scales = sqrt(mean squared distance to the 3 nearest neighbours)): exact brute force, LDS-tiled.
 */
#ifndef ENVGS_DENSIFY_H
#define ENVGS_DENSIFY_H

#include <stddef.h>
#include <stdint.h>

#include "envgs_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ENVGS_COMPACT_MAX_TENSORS 32

typedef struct envgs_rows_tensor {
    const void *src;         /* (P, row_bytes) contiguous */
    void *dst;               /* (>= kept, row_bytes) contiguous */
    int64_t row_bytes;       /* multiple of 4 */
} envgs_rows_tensor;

/* Scratch bytes of envgs_compact_scan for P rows. */
ENVGS_API size_t envgs_compact_temp_bytes(int64_t P);

/* positions[i] = number of kept rows before row i (P uint32, device); *n_kept (device uint32) = total kept.  keep: P bytes, non-zero = keep. */
ENVGS_API int envgs_compact_scan(int64_t P, const uint8_t *keep, uint32_t *positions, uint32_t *n_kept, void *temp, size_t temp_bytes,
                                 void *stream);

/* dst_t[positions[i]] = src_t[i] for every kept row i and every tensor t (host array of `count` descriptors, passed by value). */
ENVGS_API int envgs_compact_gather(int32_t count, const envgs_rows_tensor *tensors, int64_t P, const uint8_t *keep, const uint32_t *positions,
                                   void *stream);

/* out[i] = mean of the squared distances from xyz[i] to its 3 nearest OTHER points (fewer if P < 4; 0 for P == 1).  xyz (P,3), out (P). */
ENVGS_API int envgs_knn3_mean_dist2(int32_t P, const float *xyz, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVGS_DENSIFY_H */
