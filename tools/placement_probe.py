"""Probes of the fused slope + TI + MTI stencil's sensitivity to WHERE its five rasters lie in physical memory
(DESIGN.md 6).  One script, several modes (the round-2 probes 2-5 are folded in as options):

  placement_probe.py gens [N]      N generations of the five rasters in ONE process (earlier generations kept alive, so
                                   every generation sits on different physical memory), each timed, the copy kernel
                                   beside it; PROBE_WX=1 adds the other tile geometries, PROBE_POL=1 the other cache
                                   policies on the same rasters (dt_debug_set keys 2 / 1)
  placement_probe.py pool [P] [heap_gb]
                                   a bench-sized heap (heap_gb of other allocations), then a pool of P candidate
                                   rasters: every PAIR as the two outputs of the slope + radians stencil (1 read + 2
                                   write streams) -> a conflict matrix; then output TRIPLES of the fused stencil drawn
                                   from the pool, fastest first -- can a fast triple be picked at set-up?
"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from descriptools_amd import _lib  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

L = _lib.lib()
S = 16384
n = S * S
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fused(dem, fac, slope, ti, mti):
    return lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), None,
                                      ti.data_ptr(), mti.data_ptr())


def raster():
    return torch.empty(n, dtype=torch.float32, device="cuda")


def mode_gens(gens):
    keep = []
    for g in range(gens):
        bufs = [raster() for _ in range(5)]
        keep.append(bufs)
        dem, fac, slope, ti, mti = bufs
        _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
        fac.view(torch.int32).random_(0, 5000)
        run = fused(dem, fac, slope, ti, mti)
        ms = timed(run)
        cp = timed(lambda: L.dt_dev_membench_copy(ctx.h, dem.data_ptr(), slope.data_ptr(), n, -1))
        extra = []
        if os.environ.get("PROBE_WX"):
            for wx in (2, 4):
                L.dt_debug_set(2, wx)
                extra.append("wx%d %.3f" % (wx, timed(run)))
            L.dt_debug_set(2, 0)
        if os.environ.get("PROBE_POL"):
            for pol in (1, 2, 3, 4, 5):
                L.dt_debug_set(1, pol)
                extra.append("pol%d %.3f" % (pol, timed(run)))
            L.dt_debug_set(1, 0)
        print("generation %d  slope+ti+mti %.3f ms (%.0f GB/s)   copy %.3f ms   dem at %#x  %s"
              % (g, ms, n * 20 / ms / 1e6, cp, dem.data_ptr(), "  ".join(extra)), flush=True)
    for a, b in ((0, gens - 1), (gens - 1, 0)):  # mixing generations: inputs of one, outputs of another
        ms = timed(fused(keep[a][0], keep[a][1], keep[b][2], keep[b][3], keep[b][4]))
        print("inputs of generation %d, outputs of %d: %.3f ms" % (a, b, ms), flush=True)


def mode_pool(P, heap_gb):
    heap = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(int(heap_gb))]
    dem, fac = raster(), raster()
    _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
    fac.view(torch.int32).random_(0, 5000)
    pool = [raster() for _ in range(P)]
    print("heap %d GiB, pool of %d rasters" % (len(heap), P), flush=True)
    # pairs: slope + radians stencil = dem read, two float32 rasters written
    print("pair matrix (ms, slope+radians stencil writing rasters i and j):")
    for i in range(P):
        row = []
        for j in range(P):
            if i == j:
                row.append("  -  ")
                continue
            a, b = pool[i], pool[j]
            row.append("%.3f" % timed(lambda: L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, a.data_ptr(), None,
                                                                b.data_ptr()), reps=5, warm=2))
        print("  %d: %s" % (i, " ".join(row)), flush=True)
    res = []
    for tri in itertools.combinations(range(P), 3):
        ms = timed(fused(dem, fac, pool[tri[0]], pool[tri[1]], pool[tri[2]]), reps=4, warm=1)
        res.append((ms, tri))
    res.sort()
    print("triples of the fused stencil, fastest first: " + "  ".join("%s %.3f" % (t, m) for m, t in res[:6]))
    print("slowest: " + "  ".join("%s %.3f" % (t, m) for m, t in res[-3:]))
    print("median %.3f  fast (< 0.90 ms): %d of %d" % (res[len(res) // 2][0], sum(m < 0.90 for m, _ in res), len(res)))
    # the same triple with the ROLES permuted: does it matter which raster is slope / TI / MTI?
    best = res[0][1]
    print("best triple permuted: " + "  ".join(
        "%s %.3f" % (p, timed(fused(dem, fac, pool[p[0]], pool[p[1]], pool[p[2]]), reps=4, warm=1))
        for p in itertools.permutations(best)))
    # inputs: other candidates as dem / fac
    for k in range(min(P, 4)):
        if k in best:
            continue
        pool[k].copy_(dem)
        print("dem moved into pool[%d]: %.3f ms" % (k, timed(fused(pool[k], fac, pool[best[0]], pool[best[1]], pool[best[2]]),
                                                             reps=4, warm=1)), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "gens"
    if mode.isdigit():  # the round-2 command line: placement_probe.py <generations>
        mode_gens(int(mode))
    elif mode == "gens":
        mode_gens(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    else:
        mode_pool(int(sys.argv[2]) if len(sys.argv) > 2 else 7, float(sys.argv[3]) if len(sys.argv) > 3 else 14)
