#!/usr/bin/env python3
"""Headless counterpart of the reference's Example/example.py (config #1 of BASELINE.json) on the
bundled rasters: same steps (example.py:33-147), PIL instead of rasterio, no plotting; asserts the
reference's known answers (SURVEY.md 8c / BASELINE.md 2).  Needs a GPU (no CPU fallback)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# the reference example's own import block (Example/example.py:11-16), unchanged: `descriptools` is the alias
# package of descriptools_amd at the repository root
import descriptools.topoindexes as topoindexes  # noqa: E402
import descriptools.downslope as downslope  # noqa: E402
import descriptools.slope as slope  # noqa: E402
import descriptools.flowhand as flowhand  # noqa: E402
import descriptools.gfi as gfi  # noqa: E402
import descriptools.evaluation as evaluation  # noqa: E402


import descriptools_amd.rasterio_lite as rio  # noqa: E402

EX = os.path.join(ROOT, "tests", "golden", "example")


def read(name):
    return rio.read(os.path.join(EX, name))[0]


def main():
    dem, meta = rio.read_masked(os.path.join(EX, "12_dem.tif"), -100, "int16")   # example.py:33,42
    fdr = read("12_fdr.tif")
    fac, _ = rio.read_masked(os.path.join(EX, "12_fac.tif"), -100, "int")        # example.py:39,43
    px = 12.5
    river = np.where(fac > 128000, 1, 0).astype("int8")             # example.py:52
    t0 = time.time()
    sl = slope.sloper(dem, px).astype("float32")
    slr = np.arctan(sl / 100).astype("float32")
    slr = np.where(dem == -100, -100, slr)
    TopoI, ModTi = topoindexes.topographic_index(fac, slr, px, 0.1)
    down = downslope.downsloper(dem, fdr, px, 5)
    flow, indices, hand = flowhand.flow_hand_index(dem, fdr, river, px)
    geofi = gfi.gfi_calculator(hand, fac, indices, 0.4, 0.1, px)
    lnhlh = gfi.ln_hl_H_calculator(hand, fac, 0.4, 0.1, px)
    t1 = time.time()
    flood = read("WB_12_100y.tif").astype("int8")
    elements, count = np.unique(hand, return_counts=True)
    mx, mn = elements[-1], elements[1]
    desc = evaluation.minMaxScale(hand, mn, mx, -100)
    th = evaluation.calibration(desc, flood, 'under')
    binary = evaluation.binary_map(desc, th, 'under')
    c, f, class_map = evaluation.avaliacao(binary, flood)
    t2 = time.time()
    v = dem != -100
    print("descriptors %.2fs, evaluation %.2fs (host API, PCIe and numpy included)" % (t1 - t0, t2 - t1))
    print("slope %% max/mean %.5f %.5f | TI min/max/mean %.5f %.5f %.6f | down max/mean %.7f %.8f"
          % (sl[v].max(), sl[v].mean(), TopoI[v].min(), TopoI[v].max(), TopoI[v].mean(), down[v].max(),
             down[v].mean()))
    print("HAND min/max %s %s | threshold %r | correctness %r | fit %r" % (mn, mx, th, c, f))
    klass = read("hand_class.tif")
    mism = int((class_map.astype(np.uint8) != klass).sum())
    print("class map vs Example/output/hand_class.tif: %d / %d mismatches" % (mism, klass.size))
    assert (mn, mx) == (0, 259) and th == 0.012 and mism == 0
    assert c == 0.8581615676712259 and f == 0.7240945135019289
    assert abs(float(sl[v].max()) - 192.33304) < 1e-4 and abs(float(down[v].max()) - 1.9233304) < 1e-6
    assert abs(float(TopoI[v].max()) - 26.16775) < 1e-4 and abs(float(geofi[hand != -100].max()) - 10.92763) < 1e-4
    # example.py:201-217 writes the classified map as a GeoTIFF aligned with the inputs
    out_dir = os.environ.get("DT_EXAMPLE_OUT")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        rio.write(os.path.join(out_dir, "hand_class.tif"), class_map.astype(np.uint8), like=meta, nodata=0)
        rio.write(os.path.join(out_dir, "hand.tif"), hand.astype(np.float32), like=meta, nodata=-100.0)
    print("example OK")
    return {"descriptors_s": round(t1 - t0, 3), "evaluation_s": round(t2 - t1, 3), "cells": int(dem.size),
            "threshold": th, "fit": f, "class_map_mismatches": mism}


if __name__ == "__main__":
    main()
