"""downslope kernel at 16384^2: time vs elevation difference (= walk length: fixed staging cost vs per-move cost), and
vs the margin of the LDS window (DT_DBG_DS_MARGIN 24 / 20 / 16; results must be bit-identical)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from descriptools_amd import _lib  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
dem = torch.empty((S, S), dtype=torch.float32, device='cuda')
fdr = torch.empty((S, S), dtype=torch.uint8, device='cuda')
out = torch.empty((S, S), dtype=torch.float32, device='cuda')
ref = torch.empty((S, S), dtype=torch.float32, device='cuda')
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
_lib.check(L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, None, fdr.data_ptr(), None))


def run(dz, o):
    _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), S, S, 10.0, dz, 0, o.data_ptr()))


def timed(dz, o, reps=5):
    run(dz, o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        run(dz, o)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for margin in (24, 20, 16):
    L.dt_debug_set(4, margin)
    line = []
    for dz in (0.001, 1.0, 5.0, 7.5):
        line.append("dz %g: %.3f ms" % (dz, timed(dz, out if margin != 24 else ref)))
    if margin == 24:
        run(5.0, ref)
    else:
        run(5.0, out)
    torch.cuda.synchronize()
    same = margin == 24 or torch.equal(out.view(torch.int32), ref.view(torch.int32))
    print("margin %d  %s   bit-identical to margin 24: %s" % (margin, "  ".join(line), same), flush=True)
L.dt_debug_set(4, 0)
