"""Steepest-descent slope -- HIP replacement of descriptools/slope.py.

The reference tiles the raster on the host to fit a small GPU (`division_*`); one MI355X holds
16384^2 rasters many times over, so the raster is processed as ONE tile.  Tiled == untiled is the
reference's own contract, so the `division_*` arguments are accepted and ignored.
"""
import numpy as np

from . import _lib
from ._lib import C, c_f32p, check, dem_f32, ptr
from .device import host_empty, widen64


def _slope(dem32, px):
    H, W = dem32.shape
    out = host_empty((H, W), np.float32)
    check(_lib.lib().dt_slope_f32(ptr(dem32, c_f32p), H, W, float(px), ptr(out, c_f32p)))
    return out


def sloper(dem, px, division_column=0, division_row=0):
    """slope.py:96-149.  Returns float64 [H, W] holding float32 values, nodata -100 (slope.py:119)."""
    dem32 = dem_f32(dem)
    if dem32.ndim != 2:
        raise ValueError("dem must be 2-D")
    return widen64(_slope(dem32, px))


def slope_cpu(dem, px, extra, blocks=0, threads=0):
    """slope.py:152-206: `dem` is a tile with a 1-cell halo on the sides where extra[k] == 0
    (up, left, right, down); sides with extra[k] == 1 are raster edges and get the -100 ring.
    Returns the float32 slope of the tile without its ring.  blocks/threads accepted, ignored."""
    d = dem_f32(dem)
    pad = ((1 if extra[0] == 1 else 0, 1 if extra[3] == 1 else 0),
           (1 if extra[1] == 1 else 0, 1 if extra[2] == 1 else 0))
    d = np.pad(d, pad, constant_values=-100.0)
    return np.ascontiguousarray(_slope(np.ascontiguousarray(d), px)[1:-1, 1:-1])


def slope_sequential_jit(dem, px):
    """Name kept for importers of slope.py:9; runs the HIP path with the normative semantics of
    slope_gpu (the reference twin differs on the nodata test)."""
    return _slope(dem_f32(dem), px)


slope_sequential = slope_sequential_jit
