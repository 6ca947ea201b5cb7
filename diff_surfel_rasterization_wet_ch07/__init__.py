"""Drop-in for the reference's `diff_surfel_rasterization_wet_ch07` extension (7 colour channels), imported at
easyvolcap/utils/gaussian2d_utils.py:1013-1015.  MI355X-native: hand-written HIP (gfx950) behind include/envgs_raster.h."""
from envgs_amd.raster import make_package as _make_package

GaussianRasterizationSettings, GaussianRasterizer = _make_package(7)
NUM_CHANNELS = 7
__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "NUM_CHANNELS"]
