"""Probes of the fused slope + TI + MTI stencil's sensitivity to WHERE its five rasters lie in physical memory
(DESIGN.md 6).  One script, several modes (the round-2 probes 2-5 are folded in as options):

  placement_probe.py gens [N]      N generations of the five rasters in ONE process (earlier generations kept alive, so
                                   every generation sits on different physical memory), each timed, the copy kernel
                                   beside it; PROBE_WX=1 adds the other tile geometries, PROBE_POL=1 the other cache
                                   policies, PROBE_MAP=1 the experimental workgroup -> tile maps on the same rasters
                                   (dt_debug_set keys 2 / 1 / 3)
  placement_probe.py pick [K] [heap_gb]
                                   K candidate rasters allocated FIRST in the process: singles / pairs / triples as
                                   plain write streams, then the heap, then the fused stencil on the fastest and on
                                   the slowest write triple
  placement_probe.py classes [K] [plain|pad|bench]
                                   conflict classes of K consecutive allocations (pair tests against class representatives)
  placement_probe.py shift [K]     rasters allocated alike conflict as concurrent write streams: do they still when
                                   their bases differ by 64 B ... 8 KiB (sub-page shifts)?
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
        if os.environ.get("PROBE_MIX"):  # plain read / write mixes on the same five rasters (dt_dev_membench_mix)
            for nr, nw in ((2, 3), (1, 3), (0, 3), (2, 1), (1, 1), (2, 2)):
                for nt in (1, 0):
                    t = timed(lambda: L.dt_dev_membench_mix(ctx.h, dem.data_ptr(), fac.data_ptr(), slope.data_ptr(),
                                                            ti.data_ptr(), mti.data_ptr(), n, nr, nw, nt))
                    extra.append("%dr%dw%s %.3f=%.0f" % (nr, nw, "nt" if nt else "", t, n * 4 * (nr + nw) / t / 1e6))
            _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))  # the mixes overwrote nothing of dem, fac
        if os.environ.get("PROBE_MAP"):  # the experimental workgroup -> tile maps (DT_DBG_TWI_MAP = mode | param << 8)
            for mode, param in ((1, 0), (2, 0), (3, 2), (3, 4), (3, 8), (4, 37), (4, 101), (4, 1000), (5, 0)):
                L.dt_debug_set(3, mode | (param << 8))
                extra.append("m%d.%d %.3f" % (mode, param, timed(run)))
            L.dt_debug_set(3, 0)
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


def mode_pick(K, heap_gb):
    """at the START of a process: K candidate rasters, every single / pair / triple as the targets of the plain
    write-only kernel (dt_dev_membench_mix 0 reads); then the bench-sized heap, dem / fac, and the fused stencil on
    the fastest and on the slowest triple"""
    cand = [raster() for _ in range(K)]
    dummy = cand[0]

    def wr(idx):
        w = [cand[i] for i in idx] + [dummy] * (3 - len(idx))
        return timed(lambda: L.dt_dev_membench_mix(ctx.h, None, None, w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(),
                                                   n, 0, len(idx), 1), reps=3, warm=1)
    singles = [wr((i,)) for i in range(K)]
    print("single write streams (ms): " + " ".join("%.3f" % t for t in singles))
    print("pairs (ms):")
    pair = {}
    for i in range(K):
        row = []
        for j in range(K):
            if j <= i:
                row.append("  .  ")
                continue
            pair[(i, j)] = wr((i, j))
            row.append("%.3f" % pair[(i, j)])
        print("  %2d: %s" % (i, " ".join(row)), flush=True)
    tri = sorted((wr(t), t) for t in itertools.combinations(range(K), 3))
    print("triples fastest: " + "  ".join("%s %.3f" % (t, m) for m, t in tri[:8]))
    print("triples slowest: " + "  ".join("%s %.3f" % (t, m) for m, t in tri[-4:]))
    fast = [t for m, t in tri if m < 0.49]
    print("triples under 0.49 ms: %d of %d; median %.3f" % (len(fast), len(tri), tri[len(tri) // 2][0]))
    # is a triple fast exactly when its three pairs are?
    pm = sorted(pair.values())[len(pair) // 2]
    print("pair median %.3f; fastest triple's pairs: %s" % (pm, " ".join("%.3f" % pair[p] for p in itertools.combinations(tri[0][1], 2))))
    heap = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(int(heap_gb))]
    dem, fac = raster(), raster()
    _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
    fac.view(torch.int32).random_(0, 5000)
    for label, (m, t) in (("fastest", tri[0]), ("slowest", tri[-1])):
        ms = timed(fused(dem, fac, cand[t[0]], cand[t[1]], cand[t[2]]))
        print("fused stencil on the %s write triple %s (%.3f ms as plain writes): %.3f ms = %.1f %% of 8 TB/s"
              % (label, t, m, ms, n * 20 / ms / 1e6 / 80), flush=True)
    del heap


def mode_classes(K, pattern):
    """label K consecutively allocated rasters by CONFLICT CLASS: two rasters are in one class when writing them
    concurrently is slow (pair > 1.08 x the fastest pair).  pattern "bench": the chain's allocation order (1-byte
    rasters between the 4-byte ones); "pad": every raster 1 GiB + 64 MiB; default: plain 1-GiB rasters."""
    sizes = {"plain": [n * 4] * K, "pad": [n * 4 + (64 << 20)] * K,
             "bench": ([n * 4, n * 4, n, n * 4, n, n * 4, n * 4, n * 4, n * 4, n * 4, n * 4, n * 4, n * 4, n * 4, n * 4] * 4)[:K]}[pattern]
    bufs = [torch.empty(sz, dtype=torch.uint8, device="cuda") for sz in sizes]
    cand = [i for i, sz in enumerate(sizes) if sz >= n * 4]

    def pair(i, j):
        a, b = bufs[i], bufs[j]
        return timed(lambda: L.dt_dev_membench_mix(ctx.h, None, None, a.data_ptr(), b.data_ptr(), a.data_ptr(), n, 0, 2, 1),
                     reps=3, warm=1)
    reps, label, best = [], {}, 1e9
    times = {}
    for i in cand:
        found = None
        for c, r in enumerate(reps):
            t = pair(r, i)
            times[(r, i)] = t
            best = min(best, t)
        for c, r in enumerate(reps):
            if times[(r, i)] > 1.08 * best:
                found = c
                break
        if found is None:
            reps.append(i)
            found = len(reps) - 1
        label[i] = found
    print("pattern %s: %d classes; labels in allocation order: %s" % (pattern, len(reps), "".join(chr(65 + label[i]) if i in label else "." for i in range(K))))
    print("representatives %s; fastest pair %.3f ms; pair times vs representatives: %s"
          % (reps, best, "  ".join("%d-%d %.3f" % (a, b, t) for (a, b), t in sorted(times.items()))[:1500]), flush=True)
    print("addresses: " + " ".join("%#x" % b.data_ptr() for b in bufs[:12]))


def mode_shift(K):
    """do two / three write streams that CONFLICT (rasters allocated alike: mode pick) stop conflicting when their
    bases differ by less than a 4-KiB page?  Candidates are allocated with slack and used at byte offsets k * step."""
    big = os.environ.get("PROBE_BIGSHIFT", "0") == "1"  # shifts up to 128 MiB (bank / row bits) instead of < 16 KiB
    slack = (160 << 20) // 4 if big else 8192  # floats
    store = [torch.empty(n + slack, dtype=torch.float32, device="cuda") for _ in range(K)]

    def view(i, off_bytes):
        return store[i][off_bytes // 4: off_bytes // 4 + n]

    def wr(ws):
        w = list(ws) + [ws[0]] * (3 - len(ws))
        return timed(lambda: L.dt_dev_membench_mix(ctx.h, None, None, w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(),
                                                   n, 0, len(ws), 1), reps=3, warm=1)
    print("pairs at equal offsets (ms): " + "  ".join("%d-%d %.3f" % (i, j, wr([view(i, 0), view(j, 0)]))
                                                     for i, j in itertools.combinations(range(min(K, 5)), 2)))
    a, b, c = K - 3, K - 2, K - 1  # late allocations: the class that conflicts with itself
    if big:  # a triple that does conflict at equal offsets, wherever it was allocated
        slow = lambda i, j: wr([view(i, 0), view(j, 0)]) > 0.325 * (n / 2 ** 28)
        found = [(i, j, k) for i, j, k in itertools.combinations(range(K), 3) if slow(i, j) and slow(i, k) and slow(j, k)]
        if found:
            a, b, c = found[0]
        print("conflicting triple: blocks %d %d %d%s" % (a, b, c, "" if found else "  (NONE FOUND: the steps below say nothing)"))
    steps = (0, 64, 128, 256, 512, 1024, 2048, 4096, 8192)
    if big:
        steps = (0,) + tuple(1 << k for k in range(14, 27)) + (3 << 18, 3 << 20, 5 << 20, 3 << 22, 5 << 22, 3 << 24)
    for step in steps:
        p2 = wr([view(a, 0), view(b, step)])
        p3 = wr([view(a, 0), view(b, step), view(c, 2 * step)])
        print("second / third raster shifted by %5d / %5d bytes: pair %.3f ms   triple %.3f ms (%.0f GB/s)"
              % (step, 2 * step, p2, p3, n * 12 / p3 / 1e6), flush=True)
    dem, fac = raster(), raster()
    _lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
    fac.view(torch.int32).random_(0, 5000)
    for step in ((0, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 3 << 20) if big else (0, 256, 512, 1024, 2048)):
        ms = timed(fused(dem, fac, view(a, 0), view(b, step), view(c, 2 * step)))
        ms2 = timed(fused(dem, view(K - 4, step * 3)[:n] if K >= 4 else fac, view(a, 0), view(b, step), view(c, 2 * step)))
        print("fused stencil, outputs shifted by 0 / %d / %d bytes: %.3f ms = %.1f %% of 8 TB/s   (fac shifted by %d too: %.3f ms)"
              % (step, 2 * step, ms, n * 20 / ms / 1e6 / 80, 3 * step, ms2), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "gens"
    if mode.isdigit():  # the round-2 command line: placement_probe.py <generations>
        mode_gens(int(mode))
    elif mode == "gens":
        mode_gens(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    elif mode == "classes":
        mode_classes(int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else "plain")
    elif mode == "shift":
        mode_shift(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    elif mode == "pick":
        mode_pick(int(sys.argv[2]) if len(sys.argv) > 2 else 9, float(sys.argv[3]) if len(sys.argv) > 3 else 12)
    else:
        mode_pool(int(sys.argv[2]) if len(sys.argv) > 2 else 7, float(sys.argv[3]) if len(sys.argv) > 3 else 14)
