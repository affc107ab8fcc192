import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def load_example():
    """Bundled Example rasters exactly as Example/example.py:33-43,52,104 prepares them."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    ex = os.path.join(GOLD, "example")
    dem_f = np.array(Image.open(os.path.join(ex, "12_dem.tif")))
    fac_f = np.array(Image.open(os.path.join(ex, "12_fac.tif")))
    fdr = np.array(Image.open(os.path.join(ex, "12_fdr.tif"))).astype(np.uint8)
    flood = np.array(Image.open(os.path.join(ex, "WB_12_100y.tif"))).astype(np.int8)
    klass = np.array(Image.open(os.path.join(ex, "hand_class.tif"))).astype(np.uint8)
    dem = np.where(dem_f < -1e30, -100, dem_f).astype(np.int16)
    fac = np.where(fac_f < -1e30, -100, fac_f).astype(np.int64)
    river = np.where(fac > 128000, 1, 0).astype(np.int8)
    return dem, fdr, fac, river, flood, klass


def assert_float_close(got, ref, rtol=1e-5, atol=0.0, what=""):
    """north_star tolerance for float descriptors: 1e-5 relative, nodata (-100) exact."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    nod = ref == -100
    assert np.array_equal(got == -100, nod), what + ": nodata mask differs"
    both_nan = np.isnan(got) & np.isnan(ref)
    ok = both_nan | nod | (got == ref) | (np.abs(got - ref) <= rtol * np.abs(ref) + atol)
    if not ok.all():
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        raise AssertionError("%s: %d cells differ, first at %s got %r ref %r" %
                             (what, len(bad), i, got[i], ref[i]))
