"""Steepest-descent slope -- HIP replacement of descriptools/slope.py.

The reference tiles the raster on the host to fit a small GPU (`division_*`); one MI355X holds
16384^2 rasters many times over, so the raster is processed as ONE tile.  Tiled == untiled is the
reference's own contract, so the `division_*` arguments are accepted and ignored.
"""
import numpy as np

from . import _lib
from ._lib import C, c_f32p, c_f64p, check, heights, ptr
from .device import host_empty, widen64


def _slope(dem, px, pad=None):
    """float32 slope (%) of a 2-D raster: the float32 stencil when every height is a float32 value, the float64
    kernel otherwise (_lib.heights); pad: np.pad widths of a -100 ring to add first (slope_cpu)"""
    d, wide = heights(dem)
    if d.ndim != 2:
        raise ValueError("dem must be 2-D")
    if pad is not None:
        d = np.ascontiguousarray(np.pad(d, pad, constant_values=-100.0))
    H, W = d.shape
    out = host_empty((H, W), np.float32)
    if wide:
        check(_lib.lib().dt_slope_f64(ptr(d, c_f64p), H, W, float(px), ptr(out, c_f32p)))
    else:
        check(_lib.lib().dt_slope_f32(ptr(d, c_f32p), H, W, float(px), ptr(out, c_f32p)))
    return out


def sloper(dem, px, division_column=0, division_row=0):
    """slope.py:96-149.  Returns float64 [H, W] holding float32 values, nodata -100 (slope.py:119)."""
    return widen64(_slope(dem, px))


def slope_cpu(dem, px, extra, blocks=0, threads=0):
    """slope.py:152-206: `dem` is a tile with a 1-cell halo on the sides where extra[k] == 0
    (up, left, right, down); sides with extra[k] == 1 are raster edges and get the -100 ring.
    Returns the float32 slope of the tile without its ring.  blocks/threads accepted, ignored."""
    pad = ((1 if extra[0] == 1 else 0, 1 if extra[3] == 1 else 0),
           (1 if extra[1] == 1 else 0, 1 if extra[2] == 1 else 0))
    return np.ascontiguousarray(_slope(dem, px, pad)[1:-1, 1:-1])


def slope_sequential_jit(dem, px):
    """Name kept for importers of slope.py:9; runs the HIP path with the normative semantics of
    slope_gpu (the reference twin differs on the nodata test)."""
    return _slope(dem, px)


slope_sequential = slope_sequential_jit
