"""Downslope on REAL terrain: the bundled Example raster with its GIS D8 raster (flats, valley floors: walks of
thousands of moves), alone and tiled 4 x 4, plain kernel against the long-walk workspace (dt_dev_downslope_lift)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
from conftest import load_example  # noqa: E402
from descriptools_amd import _lib  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

L = _lib.lib()
ex = load_example()
dem0, fdr0 = np.asarray(ex[0], np.float32), np.ascontiguousarray(ex[1], np.uint8)
ctx = Context()
for rep in ((1, 4, 8) if len(sys.argv) > 1 and sys.argv[1] == 'big' else (1, 4)):
    dem, fdr = np.tile(dem0, (rep, rep)), np.tile(fdr0, (rep, rep))
    H, W = dem.shape
    d, f = ctx.to_device(dem), ctx.to_device(fdr)
    a, b = ctx.empty((H, W), np.float32), ctx.empty((H, W), np.float32)
    nb = int(L.dt_downslope_lift_workspace(H, W))
    work = ctx.empty((nb,), np.uint8)
    for dz in (5.0, 1.0):
        for _ in range(2):
            t0 = time.perf_counter()
            _lib.check(L.dt_dev_downslope(ctx.h, d.ptr, f.ptr, H, W, 12.5, dz, 0, a.ptr))
            ctx.sync()
            t_plain = time.perf_counter() - t0
        for _ in range(2):
            t0 = time.perf_counter()
            _lib.check(L.dt_dev_downslope_lift(ctx.h, d.ptr, f.ptr, H, W, 12.5, dz, 0, b.ptr, work.ptr, nb))
            ctx.sync()
            t_lift = time.perf_counter() - t0
        same = np.array_equal(a.to_host().view(np.int32), b.to_host().view(np.int32))
        queued = int(np.frombuffer(work.to_host()[:4].tobytes(), np.uint32)[0])
        print("Example x %d (%d x %d), dz %.0f: plain %.2f ms, with the long-walk workspace %.2f ms (%d walks queued); identical: %s"
              % (rep * rep, H, W, dz, t_plain * 1e3, t_lift * 1e3, queued, same), flush=True)
    for x in (d, f, a, b, work):
        x.free()
