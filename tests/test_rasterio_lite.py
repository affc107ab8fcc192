"""CPU: the GeoTIFF helper either side of the hot path on the bundled Example rasters."""
import os

import numpy as np

from conftest import GOLD, load_example
from descriptools_amd import rasterio_lite as rio

EX = os.path.join(GOLD, "example")


def test_read_example_rasters_like_the_reference_prepares_them():
    dem, fdr, fac, river, flood, klass = load_example()
    d, meta = rio.read_masked(os.path.join(EX, "12_dem.tif"), -100, "int16")      # example.py:33,42
    assert np.array_equal(d, dem) and d.dtype == np.int16
    assert abs(meta["pixel"][0] - 12.499164189981625) < 1e-12 and meta["nodata"] < -1e38
    f, _ = rio.read_masked(os.path.join(EX, "12_fac.tif"), -100, "int64")        # example.py:39,43
    assert np.array_equal(f, fac)
    r, m2 = rio.read(os.path.join(EX, "12_fdr.tif"))
    assert np.array_equal(r, fdr) and m2["nodata"] == 0.0
    assert rio.MODEL_TIEPOINT in meta["tags"] and rio.GEO_KEY_DIRECTORY in meta["tags"]


def test_write_round_trip_keeps_values_and_georeferencing(tmp_path):
    klass, meta = rio.read(os.path.join(EX, "hand_class.tif"))
    for arr, nod in ((klass.astype(np.uint8), 0), ((klass.astype(np.float32) - 100.0) * 0.5, -100.0),
                     (klass.astype(np.int16) - 3, -3), (klass.astype(np.int32) * 70000, None)):
        p = str(tmp_path / ("out_%s.tif" % arr.dtype.name))
        rio.write(p, arr, like=meta, nodata=nod)
        back, m = rio.read(p)
        assert np.array_equal(back, arr), arr.dtype
        for t in (rio.MODEL_PIXEL_SCALE, rio.MODEL_TIEPOINT, rio.GEO_KEY_DIRECTORY, rio.GEO_ASCII_PARAMS):
            assert tuple(np.atleast_1d(m["tags"][t])) == tuple(np.atleast_1d(meta["tags"][t])) or \
                str(m["tags"][t]).strip("\x00") == str(meta["tags"][t]).strip("\x00"), t
        if nod is not None:
            assert m["nodata"] == float(nod)
