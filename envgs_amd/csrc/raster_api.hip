// raster_api.hip -- the extern "C" entry points declared in include/envgs_raster.h.
// No torch types, no allocation: the caller owns every buffer (see the header for layouts).
#include "common.h"

using namespace envgs;

extern "C" {

size_t envgs_raster_scan_temp_bytes(int32_t P) { return scan_temp_bytes(P); }

size_t envgs_raster_sort_temp_bytes(uint32_t N, int32_t width, int32_t height)
{
    return sort_temp_bytes(N, width, height);
}

static int check_cfg(const envgs_raster_cfg *cfg)
{
    if (!cfg) return ENVGS_ERR_BAD_ARG;
    if (cfg->channels != 3 && cfg->channels != 5 && cfg->channels != 7) return ENVGS_ERR_BAD_ARG;
    if (cfg->sh_degree < 0 || cfg->sh_degree > 3) return ENVGS_ERR_BAD_ARG;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->P < 0) return ENVGS_ERR_BAD_ARG;
    return 0;
}

int envgs_raster_project(const envgs_raster_cfg *cfg, const float *means3D, const float *scales, const float *rotations,
                         const float *opacities, const float *shs, const float *transmat_precomp,
                         const float *viewmatrix, const float *projmatrix, const float *campos, float *geom, float *rgb,
                         uint8_t *clamped, int32_t *radii, uint32_t *tiles_touched, uint32_t *offsets, void *scan_temp,
                         size_t scan_temp_bytes_, uint32_t *num_rendered_host, void *stream_)
{
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (num_rendered_host) *num_rendered_host = 0;
    if (cfg->P == 0) return 0;
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !geom || !radii || !tiles_touched || !offsets) return ENVGS_ERR_BAD_ARG;
    if (!transmat_precomp && (!scales || !rotations)) return ENVGS_ERR_BAD_ARG;
    if (shs && (cfg->channels != 3 || !rgb || !clamped || !campos || cfg->sh_coeffs < (cfg->sh_degree + 1) * (cfg->sh_degree + 1))) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    rc = launch_project(cfg, means3D, scales, rotations, opacities, shs, transmat_precomp, viewmatrix, projmatrix, campos,
                        geom, rgb, clamped, radii, tiles_touched, stream);
    if (rc) return rc;
    if (scan_temp_bytes_ < scan_temp_bytes(cfg->P)) return ENVGS_ERR_TEMP_TOO_SMALL;
    rc = launch_scan(tiles_touched, offsets, cfg->P, scan_temp, scan_temp_bytes_, stream);
    if (rc) return rc;
    if (!num_rendered_host) return 0;                       // the caller reads offsets[P-1] itself, when it suits it (no host sync here)
    hipError_t e = hipMemcpyAsync(num_rendered_host, offsets + (cfg->P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(stream);
    return (int)e;
}

int envgs_raster_bin_and_render(const envgs_raster_cfg *cfg, uint32_t N, const float *geom, const int32_t *radii,
                                const float *colors, const float *bg, uint64_t *tile_pairs, uint64_t *keys_sorted, uint32_t *point_list,
                                void *bin_temp, size_t bin_temp_bytes_, uint32_t *ranges, float *out_color, float *allmap,
                                float *final_T, int32_t *n_contrib, float *weight, uint8_t *contrib_mask, void *stream_)
{
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!ranges || !out_color || !allmap || !final_T || !n_contrib || !bg) return ENVGS_ERR_BAD_ARG;
    if (cfg->P > 0 && (!geom || !radii || !colors || !weight)) return ENVGS_ERR_BAD_ARG;
    if (N > 0 && (!tile_pairs || !point_list || !bin_temp)) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    rc = launch_bin(cfg, N, geom, radii, tile_pairs, keys_sorted, point_list, bin_temp, bin_temp_bytes_, ranges, stream);
    if (rc) return rc;
    // `colors` is the caller's (possibly half) colors_precomp only when no SH were given; the SH -> RGB result of _project is fp32
    return launch_render_fwd(cfg, ranges, point_list, geom, colors, bg, out_color, allmap, final_T, n_contrib, weight, stream, nullptr, 0,
                             (cfg->feature_f16 && cfg->sh_coeffs == 0) ? 1 : 0, contrib_mask);
}

int envgs_raster_render_audit(const envgs_raster_cfg *cfg, const float *geom, const float *colors, const float *bg,
                              const uint32_t *point_list, const uint32_t *ranges, float *out_color, float *allmap, float *final_T,
                              int32_t *n_contrib, float *weight, uint8_t *contrib, int32_t lmax, const uint8_t *skip_px, void *stream_)
{
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!ranges || !out_color || !allmap || !final_T || !n_contrib || !bg || !contrib || lmax <= 0) return ENVGS_ERR_BAD_ARG;
    if (cfg->P > 0 && (!geom || !colors || !weight || !point_list)) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(contrib, 0, (size_t)cfg->width * cfg->height * (size_t)lmax, stream);
    if (e != hipSuccess) return (int)e;
    return launch_render_fwd(cfg, ranges, point_list, geom, colors, bg, out_color, allmap, final_T, n_contrib, weight, stream, contrib, lmax,
                             (cfg->feature_f16 && cfg->sh_coeffs == 0) ? 1 : 0, nullptr, skip_px);
}

int envgs_raster_backward(const envgs_raster_cfg *cfg, uint32_t N, const float *geom, const float *colors, const float *bg,
                          const uint32_t *point_list, const uint32_t *ranges, const float *final_T,
                          const int32_t *n_contrib, const uint8_t *contrib_mask, const float *dL_dcolor, const float *dL_dallmap, const float *means3D,
                          const float *scales, const float *rotations, const float *shs, const uint8_t *clamped,
                          const float *transmat_precomp, const int32_t *radii, const float *viewmatrix,
                          const float *projmatrix, const float *campos, float *grad_rec, float *dmeans3D, float *dmeans2D,
                          float *dscales, float *drots, float *dshs, float *dcolors, float *dopacities,
                          float *dtransmat_precomp, void *stream_)
{
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (cfg->P == 0) return 0;
    if (!geom || !colors || !bg || !ranges || !final_T || !n_contrib || !grad_rec || !radii ||
        !dmeans2D || !dopacities || !viewmatrix || !projmatrix)
        return ENVGS_ERR_BAD_ARG;
    if (N > 0 && !point_list) return ENVGS_ERR_BAD_ARG;
    if (shs ? (!dshs || !clamped || !campos || !means3D) : !dcolors) return ENVGS_ERR_BAD_ARG;
    if (transmat_precomp ? !dtransmat_precomp : (!scales || !rotations || !means3D || !dmeans3D || !dscales || !drots)) return ENVGS_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    rc = launch_render_bwd(cfg, ranges, point_list, geom, colors, bg, final_T, n_contrib, dL_dcolor, dL_dallmap, grad_rec, stream,
                           (cfg->feature_f16 && !shs) ? 1 : 0, contrib_mask);
    if (rc) return rc;
    return launch_project_bwd(cfg, geom, means3D, scales, rotations, shs, clamped, transmat_precomp, radii, viewmatrix,
                              projmatrix, campos, grad_rec, dmeans3D, dmeans2D, dscales, drots, dshs, dcolors, dopacities,
                              dtransmat_precomp, stream);
}

}  // extern "C"
