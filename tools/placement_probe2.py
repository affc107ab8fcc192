"""Second probe of the fused stencil's timing modes: the three OUTPUT rasters decide (tools/placement_probe.py).
Here they are carved from one slab, 1 GiB + s apart, for a list of s; dem / fac stay where they are."""
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
def timed(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
dem = torch.empty(n, dtype=torch.float32, device="cuda")
fac = torch.empty(n, dtype=torch.int32, device="cuda")
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
fac.random_(0, 5000)
MB = 1 << 20
slab = torch.empty(3 * n * 4 + 2 * 520 * MB + 4 * MB, dtype=torch.uint8, device="cuda")
base = (slab.data_ptr() + 2 * MB - 1) // (2 * MB) * (2 * MB) - slab.data_ptr()
skews = [0, 2 * MB, 8 * MB, 16 * MB, 24 * MB, 32 * MB, 48 * MB, 64 * MB, 96 * MB, 128 * MB, 160 * MB, 192 * MB, 256 * MB, 320 * MB,
         384 * MB, 448 * MB, 512 * MB, 4096, 65536, 64 * MB + 4096, 128 * MB + 65536]
for rep in range(2):
    for s in skews:
        outs = [slab[base + k * (n * 4 + s):][:n * 4].view(torch.float32) for k in range(3)]
        slope, ti, mti = outs
        ms = timed(lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr()))
        print("outputs 1 GiB + %9.3f MiB apart: %.3f ms (%.0f GB/s)" % (s / MB, ms, n * 20 / ms / 1e6), flush=True)
