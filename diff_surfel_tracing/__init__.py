"""Drop-in for the reference's `diff_surfel_tracing` extension (imported at easyvolcap/utils/optix_utils.py:7).
MI355X-native: the OptiX GAS + any-hit pipeline is replaced by a hand-written HIP LBVH (include/envgs_trace.h).

One process-wide setting is made here, on import: torch's BLAS backend for its own matmuls is switched from hipBLASLt to rocBLAS.  The
UNCHANGED EasyVolcap caller builds the surfel quads with a (4P,4,4) @ (4P,4,1) batched matmul right before every traced call
(easyvolcap/utils/optix_utils.py:59); through hipBLASLt that costs 8.7 ms per training step at 163 840 env surfels on MI355X (more than the
whole trace), through rocBLAS 1.0 ms (scratch/blas_probe.py).  The render-and-trace path itself contains no BLAS call.
Set ENVGS_KEEP_BLAS=1 to leave torch's choice alone."""
import os as _os

from envgs_amd.tracing import SurfelTracer, SurfelTracingSettings


def _prefer_rocblas():
    if _os.environ.get("ENVGS_KEEP_BLAS"):
        return
    try:
        import torch
        if torch.cuda.is_available() and getattr(torch.version, "hip", None):
            torch.backends.cuda.preferred_blas_library("cublas")          # "cublas" IS rocBLAS on ROCm builds ("cublaslt" = hipBLASLt)
    except Exception:
        pass


_prefer_rocblas()

__all__ = ["SurfelTracer", "SurfelTracingSettings"]
