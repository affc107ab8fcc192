"""Fifth probe: the row stride.  All rasters of the chain have rows of 16384 floats = 64 KiB, a power of two: the 16
rows of a stencil tile differ only in address bits >= 16.  Same kernel through the windowed entry point with rows
padded by 64 / 192 floats, on several allocation generations."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import _lib
from descriptools_amd.device import Context
L = _lib.lib()
S = 16384
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
def timed(fn, reps=8):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
keep = []
for g in range(5):
    line = "generation %d:" % g
    for pad in (0, 64, 192):
        ld = S + pad
        bufs = [torch.zeros(S * ld, dtype=torch.float32, device="cuda") for _ in range(5)]
        keep.append(bufs)
        dem, fac, slope, ti, mti = bufs
        tmp = torch.empty(S * S, dtype=torch.float32, device="cuda")
        _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, tmp.data_ptr()))
        dem.view(S, ld)[:, :S].copy_(tmp.view(S, S)); del tmp
        fac.view(torch.int32).view(S, ld)[:, :S].random_(0, 5000)
        win = _lib.Window(S, S, ld, 0, 0, S, S, 0)
        ms = timed(lambda: _lib.check(L.dt_dev_slope_twi_w(ctx.h, C.byref(win), dem.data_ptr(), fac.data_ptr(), 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr())))
        line += "   ld=%d %.3f ms" % (ld, ms)
    print(line, flush=True)
