"""the conditioning with one launch per round against coloured rounds (debug key 8) at several sizes"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from descriptools_amd import _lib
from descriptools_amd.device import Context
import condition_bench as cb
L = _lib.lib()
for n in (2048, 4096, 6144, 8192):
    dem = cb.rough(n)
    ctx = Context()
    d, f, c = ctx.to_device(dem), ctx.empty((n, n), np.float32), ctx.empty((n, n), np.uint8)
    for cm, name in ((1 << 30, "one launch per round"), (1, "coloured")):
        _lib.check(L.dt_debug_set(8, cm))
        for _ in range(2):
            _lib.check(L.dt_dev_condition_d8_async(ctx.h, d.ptr, n, n, 10.0, f.ptr, c.ptr, 40)); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            _lib.check(L.dt_dev_condition_d8_async(ctx.h, d.ptr, n, n, 10.0, f.ptr, c.ptr, 40))
        ctx.sync()
        assert ctx.status() == 0
        print("%5d^2 (%6d tiles) %-22s %.3f ms" % (n, (n // 64) ** 2, name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    _lib.check(L.dt_debug_set(8, 0))
    for b in (d, f, c): b.free()
    ctx.close()
