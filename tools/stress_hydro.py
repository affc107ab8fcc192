"""Randomised differential test of the hydrological conditioning, GPU vs the oracle's priority flood + BFS flat routing,
over many shapes / seeds, in the one-launch-per-round form and in the coloured form (debug key 8), synchronous and
asynchronous.  Exits non-zero on the first mismatch.   python tools/stress_hydro.py [cases] [seed0] [max_side]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from descriptools_amd import _lib, flowdir
from descriptools_amd.device import Context

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 300
max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
L = _lib.lib()
ctx = Context()
for k in range(cases):
    rng = np.random.default_rng(seed0 + k)
    H, W = int(rng.integers(1, max_side)), int(rng.integers(1, max_side))
    if k % 7 == 0:
        W = max(4, W // 4 * 4)          # rows aligned to 16 bytes
    if k % 11 == 0:
        H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))   # at most four tiles
    px = float(rng.choice([10.0, 12.5, 30.0]))
    dem = oracle.synth_dem(seed0 + k, 4096, 4096, int(rng.integers(0, 2500)), int(rng.integers(0, 2500)), H, W, int(rng.integers(0, 6)))
    nod = dem == -100
    kind = k % 4
    if kind == 0:    # integer heights + noise + pits: flats and depressions everywhere
        dem = np.floor(dem + rng.normal(0, 6.0, dem.shape)).astype(np.float32)
        dem[rng.random(dem.shape) < 0.02] -= 40
    elif kind == 1:  # coarse quantum: large flats across tile borders
        dem = (np.floor(dem / 8.0) * 8.0).astype(np.float32)
    elif kind == 2:  # smooth terrain with a few wide basins
        yy, xx = np.mgrid[0:H, 0:W]
        for _ in range(4):
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(5, 200)
            dem = np.where((yy - cy) ** 2 + (xx - cx) ** 2 < r * r, dem - np.float32(30.0), dem).astype(np.float32)
    else:            # no interior outlet at all but the nodata blobs
        dem = (dem + rng.normal(0, 1.0, dem.shape)).astype(np.float32)
    dem[nod] = -100
    fdr_o, filled_o = oracle.condition_d8(dem, px)
    for colour_min in (0, 1):
        _lib.check(L.dt_debug_set(8, colour_min))
        fdr, filled = flowdir.d8_conditioned(dem, px, return_filled=True)
        assert np.array_equal(filled, filled_o), (k, H, W, kind, colour_min, "filled", int((filled != filled_o).sum()))
        assert np.array_equal(fdr, fdr_o), (k, H, W, kind, colour_min, "fdr", int((fdr != fdr_o).sum()))
        d, f, c = ctx.to_device(dem), ctx.empty((H, W), np.float32), ctx.empty((H, W), np.uint8)
        _lib.check(L.dt_dev_condition_d8_async(ctx.h, d.ptr, H, W, px, f.ptr, c.ptr, 400))
        st = ctx.status()
        assert st == 0, (k, H, W, kind, colour_min, "status", st)
        assert np.array_equal(f.to_host(), filled_o) and np.array_equal(c.to_host(), fdr_o), (k, H, W, kind, colour_min, "async")
        for b in (d, f, c):
            b.free()
    _lib.check(L.dt_debug_set(8, 0))
    if k % 10 == 0:
        print("case", k, H, W, "kind", kind, "ok", flush=True)
print("cases", cases, "mismatches 0")
