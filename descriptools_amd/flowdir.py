"""D8 flow direction (net-new): ESRI code of the neighbour that realises
slope.slope_gpu's maximum, first in its scan order NW,N,NE,W,E,SW,S,SE; 0 = nodata / interior
pit; a raster-border cell with no lower neighbour drains outward."""
import numpy as np

from . import _lib
from ._lib import c_f32p, c_u8p, check, dem_f32, ptr


def d8(dem, px, return_slope=False):
    dem32 = dem_f32(dem)
    H, W = dem32.shape
    fdr = np.empty((H, W), np.uint8)
    sl = np.empty((H, W), np.float32) if return_slope else None
    check(_lib.lib().dt_d8_f32(ptr(dem32, c_f32p), H, W, float(px), ptr(fdr, c_u8p), ptr(sl, c_f32p)))
    return (fdr, sl) if return_slope else fdr


def d8_conditioned(dem, px, return_filled=False):
    """D8 for DEMs with pits and flats (the reference takes such an `fdr` from a GIS tool,
    Example/example.py:36): depressions filled, D8 on the filled surface, flats routed to their nearest outlet
    (dt_d8_conditioned_f32).  Every valid cell gets a code; no cycles."""
    dem32 = dem_f32(dem)
    H, W = dem32.shape
    fdr = np.empty((H, W), np.uint8)
    filled = np.empty((H, W), np.float32) if return_filled else None
    info = np.zeros(3, np.int32)
    check(_lib.lib().dt_d8_conditioned_f32(ptr(dem32, c_f32p), H, W, float(px), ptr(fdr, c_u8p), ptr(filled, c_f32p),
                                           info.ctypes.data_as(_lib.c_i32p)))
    if info[0]:
        raise RuntimeError("%d flat cells could not be routed" % int(info[0]))
    return (fdr, filled) if return_filled else fdr
