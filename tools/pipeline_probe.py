"""Two (or three) chains in flight: consecutive DEMs processed by independent Chain objects on their own streams --
do the idle phases of one step (perimeter countdown, node jumps, launch gaps) get filled by the other's kernels?
   python tools/pipeline_probe.py [size] [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from descriptools_amd import _lib, chain
from descriptools_amd._lib import check
from descriptools_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
L = _lib.lib()


def make(seed, overlap):
    ctx = Context()
    ch = chain.Chain(n, n, ctx=ctx, overlap=overlap, want_slope_rad=False)
    dem = ctx.empty((n, n), np.float32)
    check(L.dt_dev_synth_dem(ctx.h, seed, n, n, 0, 0, n, n, 0, dem.ptr))
    ctx.sync()
    return ctx, ch, dem


def run(chains, total):
    for c, ch, dem in chains:
        ch.run(dem.ptr, want_a_river=False)
    for c, _, _ in chains:
        c.sync()
    t0 = time.perf_counter()
    for k in range(total):
        c, ch, dem = chains[k % len(chains)]
        ch.run(dem.ptr, want_a_river=False)
    for c, _, _ in chains:
        c.sync()
    return (time.perf_counter() - t0) / total * 1e3


for overlap in (True, False):
    chains = [make(1 + i, overlap) for i in range(3)]
    for depth in (1, 2, 3):
        ms = min(run(chains[:depth], steps), run(chains[:depth], steps))
        print("overlap=%s  %d chain(s) in flight: %.3f ms per step = %.1f Gcells/s" % (overlap, depth, ms, n * n / ms / 1e6), flush=True)
    for c, ch, dem in chains:
        ch.free()
        dem.free()
        c.close()
