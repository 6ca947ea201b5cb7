"""envgs_amd -- MI355X-native (gfx950, hand-written HIP) render-and-trace hot path of EnvGS.

Only what the path needs lives here: csrc/ (HIP kernels + the C-ABI of include/*.h), the ctypes loader,
and the host-side mirrors of the reference's operator interface (raster.py, tracing.py).  The drop-in
import names (diff_surfel_rasterization_wet{,_ch05,_ch07}, diff_surfel_tracing) are thin top-level packages.
"""
__version__ = "0.1.0"


def set_feature_storage(kind="f32"):
    """"f16": the three raster packages and the tracer keep HALF copies of the per-surfel feature arrays they read (shs, colors_precomp) --
    BASELINE configs[4]'s storage variant: half the gather bytes, fp32 arithmetic -- while accepting fp32 tensors and returning fp32
    gradients (the copy is made inside the autograd node; a half INPUT would have its gradient cast to half by autograd, which underflows
    at training magnitudes).  "f32" (default): the arrays are read as passed.  Process-wide, like the reference's own build-time choice."""
    if kind not in ("f32", "f16"):
        raise ValueError("feature storage must be 'f32' or 'f16', got %r" % (kind,))
    from . import raster
    raster.FEATURE_STORAGE["f16"] = kind == "f16"


def prefer_rocblas():
    """OPT-IN (never called on import): select rocBLAS instead of hipBLASLt for torch's own matmuls in this process, announced once on the
    `envgs_amd` logger.  Only the unchanged EasyVolcap caller's get_disks batched matmul cares (optix_utils.py:59: 8.7 -> 1.0 ms per step on
    MI355X); the extensions contain no BLAS call.  Returns the previous setting (None if torch has no such switch / no ROCm device)."""
    import logging
    import torch
    if not (torch.cuda.is_available() and getattr(torch.version, "hip", None)):
        return None
    log = logging.getLogger("envgs_amd")
    try:
        before = torch.backends.cuda.preferred_blas_library()
        torch.backends.cuda.preferred_blas_library("cublas")          # "cublas" IS rocBLAS on ROCm builds ("cublaslt" = hipBLASLt)
    except (RuntimeError, AttributeError, ValueError) as e:
        log.warning("envgs_amd.prefer_rocblas: could not select rocBLAS for torch matmuls (%s)", e)
        return None
    log.warning("envgs_amd.prefer_rocblas: torch.backends.cuda.preferred_blas_library %s -> rocBLAS for this process", before)
    return before
