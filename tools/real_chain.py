"""The flow kernels on REAL terrain: the bundled Example (tiled N x N) with its GIS D8 raster -- long flow paths through
flats and valley floors -- op by op, beside the same ops on the chain's own D8 of the same raster.
   python tools/real_chain.py [N]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from conftest import load_example  # noqa: E402
from descriptools_amd import _lib  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = _lib.lib()
ex = load_example()
dem = np.tile(np.asarray(ex[0], np.float32), (rep, rep))
fdr_gis = np.tile(np.ascontiguousarray(ex[1], np.uint8), (rep, rep))
H, W = dem.shape
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
d = torch.as_tensor(dem, device="cuda")
f_gis = torch.as_tensor(fdr_gis, device="cuda")
f_own = torch.empty((H, W), dtype=torch.uint8, device="cuda")
_lib.check(L.dt_dev_slope_d8(ctx.h, d.data_ptr(), H, W, 12.5, None, f_own.data_ptr(), None))
n = H * W
thr = n // 512
T = lambda dt: torch.empty((H, W), dtype=dt, device="cuda")
fac, river, fdist, idx, hand, gfi, lnh, down = T(torch.int32), T(torch.int8), T(torch.float32), T(torch.int32), T(torch.float32), T(torch.float32), T(torch.float32), T(torch.float32)
full = _lib.Window(H, W, W, 0, 0, H, W, 0)
nb = int(L.dt_downslope_lift_workspace(H, W))
work = torch.empty(nb, dtype=torch.uint8, device="cuda")


def ops(f):
    return [
        ("flowacc_flowhand_local", lambda: L.dt_dev_flowacc_river_flowhand_local(ctx.h, f.data_ptr(), d.data_ptr(), H, W, thr, fac.data_ptr(), river.data_ptr())),
        ("flowhand_gfi_finish", lambda: L.dt_dev_flowhand_gfi_finish_w(ctx.h, C.byref(full), d.data_ptr(), f.data_ptr(), river.data_ptr(), fac.data_ptr(), 12.5, 0.4, 0.1, None, None, None, None, None, None, fdist.data_ptr(), idx.data_ptr(), None, hand.data_ptr(), None, gfi.data_ptr(), lnh.data_ptr())),
        ("downslope", lambda: L.dt_dev_downslope(ctx.h, d.data_ptr(), f.data_ptr(), H, W, 12.5, 5.0, 0, down.data_ptr())),
        ("downslope, long-walk workspace", lambda: L.dt_dev_downslope_lift(ctx.h, d.data_ptr(), f.data_ptr(), H, W, 12.5, 5.0, 0, down.data_ptr(), work.data_ptr(), nb)),
    ]


for name, f in (("GIS D8 raster", f_gis), ("the chain's own D8 (pits, no long paths)", f_own)):
    print("%s, %d x %d (%.1f M cells):" % (name, H, W, n / 1e6))
    for op, call in ops(f):
        for _ in range(2):
            _lib.check(call())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            _lib.check(call())
        e1.record(st)
        torch.cuda.synchronize()
        print("   %-34s %.3f ms" % (op, e0.elapsed_time(e1) / 3), flush=True)
