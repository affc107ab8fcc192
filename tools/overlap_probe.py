"""Where should downslope's second stream fork?  Times the whole step at 16384^2: sequential, forked after D8
(bench.py --overlap), forked before HAND's last pass (beside the two bandwidth-bound kernels of the chain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from descriptools_amd import _lib, chain
from descriptools_amd.device import Context
L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
st = torch.cuda.Stream(priority=-1); st2 = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream); ctx2 = Context(0, st2.cuda_stream)
bufs = {}
def alloc_ptr(shape, dt):
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8, np.int32: torch.int32}[dt]
    t = torch.empty(shape, dtype=tdt, device="cuda"); bufs[len(bufs)] = t
    return t.data_ptr()
dem = torch.empty((S, S), dtype=torch.float32, device="cuda")
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
ch = chain.Chain(S, S, ctx=ctx, px=10.0, river_threshold=S * S // 512, alloc=alloc_ptr, side_ctx=ctx2, overlap=True, want_slope_rad=False)
calls = {n: f for n, c, f in ch.ops(dem.data_ptr(), want_a_river=False)}
def go(name): _lib.check(calls[name]())
ORDER = ["d8", "flowacc_river", "flowhand_local", "flowhand_gfi_finish", "slope_twi"]
def step(fork_before):
    for n in ORDER:
        if n == fork_before or (fork_before == "after_d8" and n == "flowacc_river"):
            ctx.fork(ctx2); go("downslope")
            if fork_before == "seq_after_d8": pass
        go(n)
    ctx.join(ctx2)
def step_seq():
    go("d8"); ctx.fork(ctx2); go("downslope"); ctx.join(ctx2)
    for n in ORDER[1:]: go(n)
modes = [("sequential", step_seq), ("fork after d8", lambda: step("after_d8")), ("fork before flowhand_local", lambda: step("flowhand_local")),
         ("fork before flowhand_gfi_finish", lambda: step("flowhand_gfi_finish")), ("fork before slope_twi", lambda: step("slope_twi"))]
for rep in range(2):
    for name, fn in modes:
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): fn()
        e1.record(st); torch.cuda.synchronize()
        print("%-34s %.3f ms/step" % (name, e0.elapsed_time(e1) / 5), flush=True)
