"""Third probe: does the slow mode of the fused stencil depend on the row stride (pages touched per workgroup)?
Same cell count, three shapes, several allocation generations each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import _lib
from descriptools_amd.device import Context
L = _lib.lib()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
n = 16384 * 16384
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
keep = []
for g in range(5):
    bufs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(5)]
    keep.append(bufs)
    dem, fac, slope, ti, mti = bufs
    line = "generation %d:" % g
    for H, W in ((65536, 4096), (16384, 16384), (4096, 65536)):
        _lib.check(L.dt_dev_synth_dem(ctx.h, 1, H, W, 0, 0, H, W, 0, dem.data_ptr()))
        fac.view(torch.int32).random_(0, 5000)
        ms = timed(lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), H, W, 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr()))
        line += "   %dx%d %.3f ms" % (H, W, ms)
    print(line, flush=True)
