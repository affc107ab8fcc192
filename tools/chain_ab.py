"""Per-op timings of the 16384^2 chain, in chain order and in isolation (the same op five times back to back),
for the library's A/B knobs (dt_debug_set).  python tools/chain_ab.py [size]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from descriptools_amd import _lib, chain
from descriptools_amd.device import Context

L = _lib.lib()
S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
H = W = S
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
keep = []
POOL = "--pool" in sys.argv
STAG = int(sys.argv[sys.argv.index("--stagger") + 1]) if "--stagger" in sys.argv else (2 << 20) + (68 << 10)
if POOL:  # one allocation, rasters staggered
    pool0 = torch.empty(16 * (H * W * 4 + STAG) + (1 << 30), dtype=torch.uint8, device="cuda")
    cur = [pool0.data_ptr() + (-pool0.data_ptr()) % (1 << 30), 0]


def alloc(shape, dt):
    if POOL:
        nb = int(np.prod(shape)) * np.dtype(dt).itemsize
        ptr = cur[0]
        cur[0] += (nb + STAG + 4095) // 4096 * 4096
        cur[1] += 1
        return ptr
    t = torch.empty(shape, dtype={np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8,
                                  np.int32: torch.int32}[dt], device="cuda")
    keep.append(t)
    return t.data_ptr()


if POOL:
    dem_ptr = alloc((H, W), np.float32)

    class _D:
        def data_ptr(self):
            return dem_ptr
    dem = _D()
else:
    dem = torch.empty((H, W), dtype=torch.float32, device="cuda")
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, H, W, 0, 0, H, W, 0, dem.data_ptr()))
ch = chain.Chain(H, W, ctx=ctx, px=10.0, river_threshold=(H * W) // 512, alloc=alloc)
p, c = ch.p, ctx.h
full = _lib.Window(H, W, W, 0, 0, H, W, 0)
P = int(L.dt_perim_cells(H, W))
ring = [torch.empty(max(P, 1), dtype=d, device="cuda") for d in (torch.uint8, torch.int32, torch.int32, torch.int32,
                                                                  torch.float32, torch.int64)]
OPS = {
    "d8": lambda: L.dt_dev_slope_d8(c, dem.data_ptr(), H, W, 10.0, None, p("fdr"), None),
    "downslope": lambda: L.dt_dev_downslope(c, dem.data_ptr(), p("fdr"), H, W, 10.0, 5.0, 0, p("down")),
    "flowacc_river": lambda: L.dt_dev_flowacc_river(c, p("fdr"), dem.data_ptr(), H, W, ch.river_threshold, p("fac"),
                                                    p("river")),
    "fh_local": lambda: L.dt_dev_flowhand_local_w(c, C.byref(full), dem.data_ptr(), p("fdr"), p("river"), p("fac"),
                                                  *[t.data_ptr() for t in ring]),
    "fh_gfi_finish": lambda: L.dt_dev_flowhand_gfi_finish_w(c, C.byref(full), dem.data_ptr(), p("fdr"), p("river"),
                                                            p("fac"), 10.0, 0.4, 0.1, None, None, None, None, None, None,
                                                            p("fdist"), p("idx"), None, p("hand"), None, p("gfi"),
                                                            p("lnhlh")),
    "slope_twi": lambda: L.dt_dev_slope_twi(c, dem.data_ptr(), p("fac"), H, W, 10.0, 0.1, p("slope"), None, p("ti"),
                                            p("mti")),
}
ORDER = ["d8", "downslope", "flowacc_river", "fh_local", "fh_gfi_finish", "slope_twi"]


def in_chain(reps=5):
    for _ in range(2):
        for n in ORDER:
            _lib.check(OPS[n]())
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ORDER]
          for _ in range(reps)]
    for r in range(reps):
        for i, n in enumerate(ORDER):
            ev[r][i][0].record(st)
            _lib.check(OPS[n]())
            ev[r][i][1].record(st)
    torch.cuda.synchronize()
    return {n: float(np.mean([ev[r][i][0].elapsed_time(ev[r][i][1]) for r in range(reps)])) for i, n in enumerate(ORDER)}


def isolated(n, reps=5):
    for _ in range(2):
        _lib.check(OPS[n]())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        _lib.check(OPS[n]())
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def show(tag):
    a = in_chain()
    print("%-28s chain: %s  sum %.3f" % (tag, "  ".join("%s %.3f" % (n, a[n]) for n in ORDER), sum(a.values())), flush=True)
    print("%-28s alone: %s" % (tag, "  ".join("%s %.3f" % (n, isolated(n)) for n in ORDER)), flush=True)


show("default")
if "--knobs" in sys.argv:
    for key, name in ((1, "slope_twi plain stores"), (2, "fh_tile3 VH=2 occ4")):
        _lib.check(L.dt_debug_set(key, 1))
        show(name)
        _lib.check(L.dt_debug_set(key, 0))
    show("default again")

print("rasters:", {n: hex(p(n)) for n in ("slope", "fdr", "fac", "river", "fdist", "idx", "hand", "ti", "mti", "gfi", "lnhlh", "down")}, hex(dem.data_ptr()))
