"""Flow accumulation (net-new): number of upstream cells EXCLUDING self, the
convention of the bundled 12_fac.tif; int64 like the `fac` the reference's callers pass."""
import numpy as np

from . import _lib
from ._lib import c_f32p, c_i64p, c_u8p, check, ptr


def accumulate(fdr, dem=None):
    fdr = np.ascontiguousarray(fdr, np.uint8)
    H, W = fdr.shape
    d = None
    if dem is not None:
        # the DEM is only a nodata mask here (dem <= -100 -> -100): taken in the raster's own dtype, whatever it is
        d = np.where(np.asarray(dem) <= -100, np.float32(-100), np.float32(0)).astype(np.float32)
    acc = np.empty((H, W), np.int64)
    check(_lib.lib().dt_flowacc_u8(ptr(fdr, c_u8p), ptr(d, c_f32p), H, W, ptr(acc, c_i64p)))
    return acc
