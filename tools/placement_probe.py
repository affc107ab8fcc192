"""Probe of the fused stencil's two timing modes: within ONE process, several generations of the five
rasters (earlier generations kept alive, so every generation sits on different physical memory), each timed; then
the same generation again with per-raster address skews inside one slab.  usage: placement_probe.py [generations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import _lib
from descriptools_amd.device import Context
L = _lib.lib()
S = 16384
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
n = S * S
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 6
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
keep = []
for g in range(gens):
    bufs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(5)]
    keep.append(bufs)
    dem, fac, slope, ti, mti = bufs
    _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
    fac.view(torch.int32).random_(0, 5000)
    ms = timed(lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr()))
    cp = timed(lambda: L.dt_dev_membench_copy(ctx.h, dem.data_ptr(), slope.data_ptr(), n, -1))
    print("generation %d  slope+ti+mti %.3f ms (%.0f GB/s)   copy %.3f ms   dem at %#x" % (g, ms, n * 20 / ms / 1e6, cp, dem.data_ptr()), flush=True)
# mixing generations: inputs of one, outputs of another
for a, b in ((0, gens - 1), (gens - 1, 0)):
    dem, fac = keep[a][0], keep[a][1]
    slope, ti, mti = keep[b][2], keep[b][3], keep[b][4]
    ms = timed(lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr()))
    print("inputs of generation %d, outputs of %d: %.3f ms" % (a, b, ms), flush=True)
