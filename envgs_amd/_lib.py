"""ctypes loader for libenvgs_hip.so (the C-ABI of include/*.h).  Fails loudly: there is no CPU or
PyTorch fallback for the product path -- if the HIP library is missing or does not load, importing an
operator raises."""
import ctypes
import os

# torch FIRST: PyTorch-ROCm wheels carry their own libamdhip64, and whichever HIP runtime is loaded first serves every later library with
# the same SONAME.  If this library (linked against /opt/rocm) were opened before torch, torch would run on a runtime it was not built with
# and the stream handles it passes here would belong to a different runtime instance ("hipErrorNoDevice" at the first launch).
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libenvgs_hip.so")
_lib = None

c_void_p, c_int, c_uint32, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_size_t


class RasterCfg(ctypes.Structure):
    """struct envgs_raster_cfg (include/envgs_raster.h)."""
    _fields_ = [("P", ctypes.c_int32), ("sh_degree", ctypes.c_int32), ("sh_coeffs", ctypes.c_int32),
                ("channels", ctypes.c_int32), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("bg_len", ctypes.c_int32), ("debug", ctypes.c_int32), ("scale_modifier", ctypes.c_float),
                ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("feature_f16", ctypes.c_int32)]


class AdamTensor(ctypes.Structure):
    """struct envgs_adam_tensor (include/envgs_optim.h)."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("numel", ctypes.c_int64), ("lr", ctypes.c_float), ("step", ctypes.c_float)]


class RowsTensor(ctypes.Structure):
    """struct envgs_rows_tensor (include/envgs_densify.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("row_bytes", ctypes.c_int64)]


class TraceLists(ctypes.Structure):
    """struct envgs_trace_lists (include/envgs_trace.h)."""
    _fields_ = [("hit_lists", ctypes.c_void_p), ("hit_cnt", ctypes.c_void_p), ("n_used", ctypes.c_void_p), ("cap", ctypes.c_int32),
                ("stack_spill", ctypes.c_void_p), ("surf_acc", ctypes.c_void_p), ("surf_cnt", ctypes.c_void_p), ("surf_off", ctypes.c_void_p),
                ("scan_temp", ctypes.c_void_p), ("scan_temp_bytes", ctypes.c_size_t), ("ray_keys", ctypes.c_void_p),
                ("ray_order", ctypes.c_void_p), ("ray_sort_temp", ctypes.c_void_p), ("ray_sort_temp_bytes", ctypes.c_size_t),
                ("records", ctypes.c_void_p),
                ("num_records", ctypes.c_uint64), ("hit_state", ctypes.c_void_p), ("entries", ctypes.c_void_p), ("pairs", ctypes.c_void_p),
                ("n_entries", ctypes.c_void_p), ("compact_rows", ctypes.c_uint64), ("row_off", ctypes.c_void_p), ("batch_rows", ctypes.c_void_p),
                ("row_blk", ctypes.c_void_p), ("sh_perm", ctypes.c_void_p), ("state_planes", ctypes.c_int32),
                ("sparse_hits", ctypes.c_void_p), ("sparse_cap", ctypes.c_uint64),
                ("defer_reduce", ctypes.c_uint32), ("reserved0", ctypes.c_uint32)]


class TraceCfg(ctypes.Structure):
    """struct envgs_trace_cfg (include/envgs_trace.h)."""
    _fields_ = [("P", ctypes.c_int32), ("num_rays", ctypes.c_int32), ("sh_degree", ctypes.c_int32),
                ("sh_coeffs", ctypes.c_int32), ("max_trace_depth", ctypes.c_int32), ("start_from_first", ctypes.c_int32),
                ("has_others", ctypes.c_int32), ("bg_len", ctypes.c_int32), ("debug", ctypes.c_int32),
                ("ray_h", ctypes.c_int32), ("ray_w", ctypes.c_int32),
                ("scale_modifier", ctypes.c_float), ("specular_threshold", ctypes.c_float), ("feature_f16", ctypes.c_int32)]


# every symbol include/*.h declares: name -> (restype, argtypes)
_P = c_void_p
SYMBOLS = {
    "envgs_raster_scan_temp_bytes": (c_size_t, [ctypes.c_int32]),
    "envgs_raster_sort_temp_bytes": (c_size_t, [c_uint32, ctypes.c_int32, ctypes.c_int32]),
    "envgs_raster_project": (c_int, [ctypes.POINTER(RasterCfg)] + [_P] * 15 + [_P, c_size_t, ctypes.POINTER(c_uint32), _P]),
    "envgs_raster_bin_and_render": (c_int, [ctypes.POINTER(RasterCfg), c_uint32] + [_P] * 7 + [_P, c_size_t] + [_P] * 7 + [_P]),
    "envgs_raster_render_audit": (c_int, [ctypes.POINTER(RasterCfg)] + [_P] * 11 + [ctypes.c_int32, _P, _P]),
    "envgs_raster_backward": (c_int, [ctypes.POINTER(RasterCfg), c_uint32] + [_P] * 29 + [_P]),
    "envgs_bvh_temp_bytes": (c_size_t, [ctypes.c_int32]),
    "envgs_bvh_node_floats": (c_size_t, [ctypes.c_int32]),
    "envgs_bvh_build": (c_int, [ctypes.c_int32, _P, _P, _P, _P, c_size_t, ctypes.c_int32, _P]),
    "envgs_bvh_refit": (c_int, [ctypes.c_int32, _P, _P, _P, _P, _P, c_size_t, ctypes.c_int32, _P]),
    "envgs_bvh_quality": (c_int, [ctypes.c_int32, _P, _P, _P]),
    "envgs_trace_stack_spill_ints": (c_size_t, [ctypes.c_int32]),
    "envgs_trace_ray_sort_temp_bytes": (c_size_t, [ctypes.c_int32]),
    "envgs_trace_ray_order": (c_int, [ctypes.c_int32, _P, _P, _P, ctypes.c_int32, _P, _P, _P, c_size_t, _P]),
    "envgs_trace_forward": (c_int, [ctypes.POINTER(TraceCfg)] + [_P] * 22 + [ctypes.POINTER(TraceLists), _P]),
    "envgs_trace_backward": (c_int, [ctypes.POINTER(TraceCfg)] + [_P] * 35 + [ctypes.POINTER(TraceLists), _P]),
    "envgs_trace_backward_join": (c_int, [_P]),
    "envgs_sh_colors_forward": (c_int, [ctypes.c_int32] * 4 + [_P] * 7 + [_P]),
    "envgs_sh_colors_backward": (c_int, [ctypes.c_int32] * 4 + [_P] * 9 + [_P]),
    "envgs_surfel_quads": (c_int, [ctypes.c_int32] + [_P] * 5 + [_P]),
    "envgs_blend_forward": (c_int, [ctypes.c_int32] * 3 + [_P] * 3 + [_P]),
    "envgs_blend_backward": (c_int, [ctypes.c_int32] * 3 + [_P] * 5 + [_P]),
    "envgs_bounce_rays_forward": (c_int, [ctypes.c_int32] + [_P] * 8 + [_P]),
    "envgs_bounce_rays_backward": (c_int, [ctypes.c_int32] + [_P] * 13 + [_P]),
    "envgs_bounce_blend_forward": (c_int, [ctypes.c_int32] + [_P] * 5 + [_P]),
    "envgs_bounce_blend_backward": (c_int, [ctypes.c_int32] + [_P] * 8 + [_P]),
    "envgs_bounce_pack_mid": (c_int, [ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_int32] + [_P] * 8 + [_P]),
    "envgs_reflect_forward": (c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_float] + [_P] * 8 + [_P]),
    "envgs_reflect_backward": (c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_float] + [_P] * 11 + [_P]),
    "envgs_surface_normal_forward": (c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float] + [_P] * 4 + [_P]),
    "envgs_surface_normal_backward": (c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float] + [_P] * 5 + [_P]),
    "envgs_fused_adam": (c_int, [ctypes.c_int32, ctypes.POINTER(AdamTensor), ctypes.c_float, ctypes.c_float, ctypes.c_float, _P]),
    "envgs_compact_temp_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "envgs_compact_scan": (c_int, [ctypes.c_int64, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    "envgs_compact_gather": (c_int, [ctypes.c_int32, ctypes.POINTER(RowsTensor), ctypes.c_int64, _P, _P, _P]),
    "envgs_knn3_mean_dist2": (c_int, [ctypes.c_int32, _P, _P, _P]),
    "envgs_l1_ssim_partial_count": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "envgs_l1_ssim_forward": (c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P, _P]),
    "envgs_l1_ssim_backward": (c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _P]),
    "envgs_debug_set": (None, [ctypes.c_int32, ctypes.c_int32]),
    "envgs_debug_get": (ctypes.c_int32, [ctypes.c_int32]),
    "envgs_prof_enable": (None, [c_int]),
    "envgs_prof_select": (None, [ctypes.c_uint64]),
    "envgs_prof_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
    "envgs_prof_kernel_name": (ctypes.c_char_p, [c_int]),
}


LIB_DIAG_PATH = os.path.join(HERE, "libenvgs_hip_diag.so")
_libs = {}
_selected = {"kind": "product"}


def _open(path):
    if not os.path.exists(path):
        raise RuntimeError(
            "envgs_amd: %s is missing. Build it with `python -m envgs_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the render-and-trace path." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """The library every entry point of the package calls into: the PRODUCT build, unless a test / `bench.py --diag` selected the diagnostic one."""
    global _lib
    kind = _selected["kind"]
    lib = _libs.get(kind)
    if lib is None:
        lib = _libs[kind] = _open(LIB_PATH if kind == "product" else LIB_DIAG_PATH)
    _lib = lib
    return lib


def select(kind):
    """"product" (default) or "diag": the diagnostic build carries, behind envgs_debug_set, the superseded A/B kernels the product library was
    trimmed of (csrc: ENVGS_DIAG).  Both export the same C-ABI; each keeps its own diagnostic switches and timers.  Returns the previous kind."""
    if kind not in ("product", "diag"):
        raise ValueError(kind)
    old = _selected["kind"]
    _selected["kind"] = kind
    load()
    return old


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        raise RuntimeError("envgs_amd: %s failed with code %d (%s)" % (
            what, rc, {-1: "bad argument", -2: "temp buffer too small"}.get(rc, "hipError_t")))
