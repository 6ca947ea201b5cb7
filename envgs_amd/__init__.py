"""envgs_amd -- MI355X-native (gfx950, hand-written HIP) render-and-trace hot path of EnvGS.

Only what the path needs lives here: csrc/ (HIP kernels + the C-ABI of include/*.h), the ctypes loader,
and the host-side mirrors of the reference's operator interface (raster.py, tracing.py).  The drop-in
import names (diff_surfel_rasterization_wet{,_ch05,_ch07}, diff_surfel_tracing) are thin top-level packages.
"""
__version__ = "0.1.0"
