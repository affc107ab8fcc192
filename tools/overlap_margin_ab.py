"""the overlapped chain step at 16384^2 with the downslope window's margin 24 (default) / 20 / 16 (debug key 4: less
LDS per downslope workgroup leaves room for the flow kernels' tiles beside it), and serial for reference"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from descriptools_amd import _lib, chain  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = Context()
d = ctx.empty((S, S), np.float32)
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, d.ptr))
for overlap in (True, False):
    ch = chain.Chain(S, S, ctx=ctx, px=10.0, overlap=overlap, tune_placement=False)
    for m in (0, 20, 16):
        _lib.check(L.dt_debug_set(4, m))
        for _ in range(3):
            ch.run(d.ptr)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            ch.run(d.ptr)
        ctx.sync()
        print("overlap %s, downslope margin %d: %.3f ms/step" % (overlap, m or 24, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    _lib.check(L.dt_debug_set(4, 0))
    ch.free()
