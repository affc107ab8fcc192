"""A/B of the downslope kernels at 16384^2 on the chain's own D8 raster: time + equality of the rasters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import _lib
from descriptools_amd.device import Context
L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
dem = torch.empty((S, S), dtype=torch.float32, device="cuda")
fdr = torch.empty((S, S), dtype=torch.uint8, device="cuda")
outs = [torch.empty((S, S), dtype=torch.float32, device="cuda") for _ in range(2)]
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
_lib.check(L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, None, fdr.data_ptr(), None))
def run(o): _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), S, S, 10.0, 5.0, 0, o.data_ptr()))
for rep in range(2):
    for old in (1, 0):  # 1 = the one-thread-per-cell walk on global memory, 0 = the windowed kernel
        _lib.check(L.dt_set_flow_impl(1 if old else 2))
        for _ in range(2): run(outs[old])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): run(outs[old])
        e1.record(st); torch.cuda.synchronize()
        print("old=%d  %.3f ms" % (old, e0.elapsed_time(e1) / 5), flush=True)
_lib.check(L.dt_set_flow_impl(2))
print("identical:", bool(torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))))
# staging + store alone: dz = 0 stops every walk before its first move
for old in (0,):
    for dz in (0.0, 1.0, 2.0, 5.0, 10.0):
        f = lambda: _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), S, S, 10.0, dz, 0, outs[0].data_ptr()))
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): f()
        e1.record(st); torch.cuda.synchronize()
        print("old=%d dz=%4.1f  %.3f ms" % (old, dz, e0.elapsed_time(e1) / 5), flush=True)
