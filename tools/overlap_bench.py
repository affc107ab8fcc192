"""Experiment: run downslope on a side stream concurrently with the flow kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from descriptools_amd import _lib, chain
from descriptools_amd.device import Context
L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.set_stream(s1)
c1, c2 = Context(0, s1.cuda_stream), Context(0, s2.cuda_stream)
keep = []
def alloc(shape, dt):
    t = torch.empty(shape, dtype={np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8, np.int32: torch.int32}[dt], device=dev)
    keep.append(t); return t.data_ptr()
dem = torch.empty((S, S), dtype=torch.float32, device=dev)
_lib.check(L.dt_dev_synth_dem(c1.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
ch = chain.Chain(S, S, ctx=c1, px=10.0, river_threshold=S * S // 512, alloc=alloc)
p, N, H, W = ch.p, S * S, S, S
def step(overlap):
    _lib.check(L.dt_dev_slope_d8(c1.h, dem.data_ptr(), H, W, ch.px, None, p("fdr"), None))
    if overlap:
        ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
        _lib.check(L.dt_dev_downslope(c2.h, dem.data_ptr(), p("fdr"), H, W, ch.px, ch.dz, 0, p("down")))
    _lib.check(L.dt_dev_flowacc_river(c1.h, p("fdr"), dem.data_ptr(), H, W, ch.river_threshold, p("fac"), p("river")))
    if overlap == 2:  # slope+TI+MTI also on the side stream, behind flow accumulation
        ev3 = torch.cuda.Event(); ev3.record(s1); s2.wait_event(ev3)
        _lib.check(L.dt_dev_slope_twi(c2.h, dem.data_ptr(), p("fac"), H, W, ch.px, ch.n_top, p("slope"), None, p("ti"), p("mti")))
    _lib.check(L.dt_dev_flowhand(c1.h, dem.data_ptr(), p("fdr"), p("river"), p("fac"), H, W, ch.px, p("fdist"), p("idx"), p("hand"), p("a_river")))
    if overlap != 2:
        _lib.check(L.dt_dev_slope_twi(c1.h, dem.data_ptr(), p("fac"), H, W, ch.px, ch.n_top, p("slope"), None, p("ti"), p("mti")))
    _lib.check(L.dt_dev_gfi_lnhlh(c1.h, p("hand"), p("a_river"), p("fac"), N, ch.n_gfi, ch.b, ch.px, p("gfi"), p("lnhlh")))
    if overlap:
        ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
    else:
        _lib.check(L.dt_dev_downslope(c1.h, dem.data_ptr(), p("fdr"), H, W, ch.px, ch.dz, 0, p("down")))
for overlap in (0, 1, 2, 0, 1, 2):
    for _ in range(2): step(overlap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step(overlap)
    torch.cuda.synchronize(); print("overlap", overlap, "ms/step %.3f" % ((time.perf_counter() - t0) / 5 * 1e3))
