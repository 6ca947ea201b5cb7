"""Drop-in for the reference's `diff_surfel_tracing` extension (imported at easyvolcap/utils/optix_utils.py:7).
MI355X-native: the OptiX GAS + any-hit pipeline is replaced by a hand-written HIP LBVH (include/envgs_trace.h)."""
from envgs_amd.tracing import SurfelTracer, SurfelTracingSettings

__all__ = ["SurfelTracer", "SurfelTracingSettings"]
