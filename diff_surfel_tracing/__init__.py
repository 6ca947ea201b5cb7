"""Drop-in for the reference's `diff_surfel_tracing` extension (imported at easyvolcap/utils/optix_utils.py:7).
MI355X-native: the OptiX GAS + any-hit pipeline is replaced by a hand-written HIP LBVH (include/envgs_trace.h).

Importing this package changes NO process-wide state (round 3 switched torch's BLAS backend here).  One thing a
maintainer should know about the UNCHANGED EasyVolcap caller on this stack: it builds the surfel quads with a (4P,4,4) @ (4P,4,1) batched
matmul right before every traced call (easyvolcap/utils/optix_utils.py:59); through hipBLASLt (torch's default on ROCm) that costs 8.7 ms
per training step at 163 840 env surfels on MI355X -- more than the whole trace --, through rocBLAS 1.0 ms (scratch/blas_probe.py).
Recommended: pin it where the caller lives (one line in the sampler's __init__), or OPT IN here with
ENVGS_PREFER_ROCBLAS=1 / envgs_amd.prefer_rocblas().  The render-and-trace path itself contains no BLAS call."""
import os as _os

from envgs_amd.tracing import SurfelTracer, SurfelTracingSettings

if _os.environ.get("ENVGS_PREFER_ROCBLAS") == "1":
    import envgs_amd as _pkg
    _pkg.prefer_rocblas()

__all__ = ["SurfelTracer", "SurfelTracingSettings"]
