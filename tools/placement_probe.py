"""Probe of the fused stencil's timing modes: one process, the five rasters carved from one slab with a
per-raster skew; one line per skew.  usage: placement_probe.py [skew_bytes ...]"""
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
PAD = 64 << 20
slab = torch.empty(5 * (n * 4 + PAD), dtype=torch.uint8, device="cuda")
skews = [int(a, 0) for a in sys.argv[1:]] or [0, 256, 4096, 65536, 65536 + 4096, 1 << 20, (2 << 20) + 4096, (6 << 20) + 8192 + 256]
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rep in range(2):
    for skew in skews:
        assert skew * 4 <= PAD and skew % 16 == 0
        bufs = [slab[i * (n * 4 + PAD) + i * skew:][:n * 4].view(torch.float32) for i in range(5)]
        dem, fac, slope, ti, mti = bufs
        _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
        fac.view(torch.int32).random_(0, 5000)
        ms = timed(lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr()))
        print("skew %9d B  slope+ti+mti %.3f ms (%.0f GB/s)" % (skew, ms, n * 20 / ms / 1e6), flush=True)
