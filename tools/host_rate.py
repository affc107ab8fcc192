"""PCIe-inclusive rate of the host-tier API (numpy in / numpy out) for the whole chain, DESIGN.md 6"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from descriptools_amd import chain, _lib
from descriptools_amd.device import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
L = _lib.lib(); ctx = Context()
d = ctx.empty((n, n), np.float32)
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, n, n, 0, 0, n, n, 0, d.ptr)); dem = d.to_host(); d.free()
for it in range(3):
    t0 = time.perf_counter(); out = chain.run_host(dem, 10.0, want_slope_rad=False); t1 = time.perf_counter()
    print("run_host %dx%d: %.3f s -> %.1f Mcells/s (H2D of the DEM, 13 rasters D2H, int64 widening included)" % (n, n, t1 - t0, n * n / (t1 - t0) / 1e6))
