"""Which two-stream schedule of the chain's five ops is fastest?  (GPU; not part of the suites.)
   python tools/schedule_probe.py [size] [steps]
Variants: where the side branch forks, what it carries, and the two streams' priorities.  Every variant runs the
same five calls on the same rasters; the rasters of the last variant are compared with the serial run's."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from descriptools_amd import _lib, chain
from descriptools_amd._lib import check
from descriptools_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = _lib.lib()


def build(main_prio, side_prio):
    ctx = Context(priority=main_prio)
    side = Context(priority=side_prio)
    ch = chain.Chain(n, n, ctx=ctx, side_ctx=side, overlap=True)
    return ctx, side, ch


def run_variant(ctx, side, ch, dem, plan):
    """plan: list of (op name, 'm' | 's'), in launch order, with 'fork' / 'join' markers"""
    ops = {name: call for name, _, call in ch.ops(dem, serial=True)}
    # ops(serial=True) binds everything to the main context: rebuild the side-bound calls by hand
    H = W = n
    p = ch.p
    side_calls = {
        "downslope": lambda: L.dt_dev_downslope(side.h, dem, p("fdr"), H, W, ch.px, ch.dz, 0, p("down")),
        "slope_twi": lambda: L.dt_dev_slope_twi(side.h, dem, p("fac"), H, W, ch.px, ch.n_top, p("slope"),
                                                p("slope_rad"), p("ti"), p("mti")),
    }
    for item in plan:
        if item == "fork":
            ctx.fork(side)
        elif item == "join":
            ctx.join(side)
        else:
            name, where = item
            check(ops[name]() if where == "m" else side_calls[name]())


PLANS = {
    "serial": [("d8", "m"), ("downslope", "m"), ("flowacc_flowhand_local", "m"), ("flowhand_gfi_finish", "m"),
               ("slope_twi", "m")],
    "A fork after d8: downslope": [("d8", "m"), "fork", ("downslope", "s"), ("flowacc_flowhand_local", "m"),
                                   ("flowhand_gfi_finish", "m"), ("slope_twi", "m"), "join"],
    "B fork after fa: downslope": [("d8", "m"), ("flowacc_flowhand_local", "m"), "fork", ("downslope", "s"),
                                   ("flowhand_gfi_finish", "m"), ("slope_twi", "m"), "join"],
    "C fork after fa: twi+downslope": [("d8", "m"), ("flowacc_flowhand_local", "m"), "fork", ("slope_twi", "s"),
                                       ("downslope", "s"), ("flowhand_gfi_finish", "m"), "join"],
    "D fork after d8: downslope; after fa: twi on side": [("d8", "m"), "fork", ("downslope", "s"),
                                                          ("flowacc_flowhand_local", "m"), "fork", ("slope_twi", "s"),
                                                          ("flowhand_gfi_finish", "m"), "join"],
    "E fork after fa: downslope+twi": [("d8", "m"), ("flowacc_flowhand_local", "m"), "fork", ("downslope", "s"),
                                       ("slope_twi", "s"), ("flowhand_gfi_finish", "m"), "join"],
}

for main_prio, side_prio in ((None, None), (-1, 1), (1, -1)):
    ctx, side, ch = build(main_prio, side_prio)
    dem = ctx.empty((n, n), np.float32)
    check(L.dt_dev_synth_dem(ctx.h, 1, n, n, 0, 0, n, n, 0, dem.ptr))
    ctx.sync()
    print("priorities main %s side %s   placement %s" % (main_prio, side_prio, ch.placement.get("classes")), flush=True)
    for name, plan in PLANS.items():
        for _ in range(3):
            run_variant(ctx, side, ch, dem.ptr, plan)
        ctx.sync(); side.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_variant(ctx, side, ch, dem.ptr, plan)
        ctx.sync(); side.sync()
        print("  %-52s %.3f ms" % (name, (time.perf_counter() - t0) / steps * 1e3), flush=True)
    ch.free()
    dem.free()
