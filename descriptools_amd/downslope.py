"""Downslope index -- HIP replacement of descriptools/downslope.py.  The reference runs a GPU walk
that marks failures -50 and then a single-threaded CPU repair over the whole raster; here ONE
kernel restates both (a walk the kernel completes is identical in the repair)."""
import numpy as np

from . import _lib
from ._lib import c_f32p, c_f64p, c_u8p, check, heights, ptr
from .device import host_empty, widen64


def _run(dem, flow_direction, px, elevation_difference, raw):
    d, wide = heights(dem)
    fdr = np.ascontiguousarray(flow_direction, np.uint8)
    H, W = d.shape
    out = host_empty((H, W), np.float32)
    if wide:  # heights that float32 cannot hold: the walk on float64 heights (downslope.py:468 in the DEM's own dtype)
        check(_lib.lib().dt_downslope_f64(ptr(d, c_f64p), ptr(fdr, c_u8p), H, W, float(px),
                                          float(elevation_difference), raw, ptr(out, c_f32p)))
    else:
        check(_lib.lib().dt_downslope(ptr(d, c_f32p), ptr(fdr, c_u8p), H, W, float(px),
                                      float(elevation_difference), raw, ptr(out, c_f32p)))
    return out


def downsloper(dem, flow_direction, px, elevation_difference, column_division=0, row_division=0):
    """downslope.py:317-376 -> float32."""
    return _run(dem, flow_direction, px, elevation_difference, 0)


def downslope_cpu(dem, flow_direction, px, elevation_difference, blocks=0, threads=0):
    """downslope.py:379-431 alone: float64, walks the kernel cannot finish are the marker -50."""
    return widen64(_run(dem, flow_direction, px, elevation_difference, 1))


def downslope_sequential_jit(dem, flow_direction, px, elevation_difference, downslope=None):
    """downslope.py:161-314: with `downslope` given, repairs its -50 cells (and writes -100 where
    dem == -100); without, computes every cell."""
    full = _run(dem, flow_direction, px, elevation_difference, 0)
    if downslope is None or np.asarray(downslope).size == 0:
        return full
    d = np.asarray(dem)
    downslope[...] = np.where(d == -100, -100, np.where(downslope == -50, full, downslope))
    return downslope


downslope_sequential = downslope_sequential_jit
