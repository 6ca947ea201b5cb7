"""Drop-in for the reference's `diff_surfel_rasterization_wet` extension (3 colour channels), imported at
easyvolcap/utils/gaussian2d_utils.py:1013-1015.  MI355X-native: hand-written HIP (gfx950) behind include/envgs_raster.h."""
from envgs_amd.raster import make_package as _make_package

GaussianRasterizationSettings, GaussianRasterizer = _make_package(3)
NUM_CHANNELS = 3
__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "NUM_CHANNELS"]
