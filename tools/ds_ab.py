"""A/B of one debug knob of the downslope kernel at 16384^2 (same process, interleaved): tools/ds_ab.py [key=6] [size]
key 6 = DT_DBG_DS_NO_RTAB (1: the reciprocal per cell as before round 4; 0: the (moves, diagonal moves) table)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from descriptools_amd import _lib  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

L = _lib.lib()
KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
dem = torch.empty((S, S), dtype=torch.float32, device='cuda')
fdr = torch.empty((S, S), dtype=torch.uint8, device='cuda')
outs = [torch.empty((S, S), dtype=torch.float32, device='cuda') for _ in range(2)]
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
_lib.check(L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, None, fdr.data_ptr(), None))


def timed(v, dz, reps=10):
    L.dt_debug_set(KEY, v)
    o = outs[v]
    _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), S, S, 10.0, dz, 0, o.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), S, S, 10.0, dz, 0, o.data_ptr()))
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dz in (0.001, 5.0):
    for rep in range(3):
        a, b = timed(0, dz), timed(1, dz)
        same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
        print("dz %g  key %d = 0: %.3f ms   = 1: %.3f ms   bit-identical: %s" % (dz, KEY, a, b, same), flush=True)
L.dt_debug_set(KEY, 0)
