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
    """Process-wide and therefore announced: one log line (logger `envgs_amd`, WARNING) says what was changed and how to opt out."""
    if _os.environ.get("ENVGS_KEEP_BLAS"):
        return
    import logging
    import torch
    if not (torch.cuda.is_available() and getattr(torch.version, "hip", None)):
        return
    log = logging.getLogger("envgs_amd")
    try:
        before = torch.backends.cuda.preferred_blas_library()
        torch.backends.cuda.preferred_blas_library("cublas")          # "cublas" IS rocBLAS on ROCm builds ("cublaslt" = hipBLASLt)
    except (RuntimeError, AttributeError, ValueError) as e:            # a torch build without the switch: leave it alone, but say so
        log.warning("diff_surfel_tracing: could not select rocBLAS for torch matmuls (%s); get_disks' batched matmul may cost ~8 ms per step", e)
        return
    log.warning("diff_surfel_tracing: torch.backends.cuda.preferred_blas_library %s -> rocBLAS for this process (the caller's get_disks batched "
                "matmul: 8.7 -> 1.0 ms per step on MI355X); set ENVGS_KEEP_BLAS=1 to keep torch's choice", before)


_prefer_rocblas()

__all__ = ["SurfelTracer", "SurfelTracingSettings"]
