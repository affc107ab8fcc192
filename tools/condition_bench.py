"""Timing of the hydrological conditioning: the bundled Example DEM (2178 x 1534, an already filled
real DEM with 223,054 flat cells) and a rough synthetic 4096^2 DEM (pits, noise, integer plateaus); synchronous form
(iterates to the fixed point, one flag read per batch of rounds), asynchronous form (fixed budget, no host
synchronisation), and the chain with and without it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import ctypes as C  # noqa: E402
from descriptools_amd import _lib, chain  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

L = _lib.lib()


def rough(n, seed=3):
    ctx = Context()
    d = ctx.empty((n, n), np.float32)
    _lib.check(L.dt_dev_synth_dem(ctx.h, seed, n, n, 0, 0, n, n, 2, d.ptr))
    dem = d.to_host()
    d.free()
    ctx.close()
    rng = np.random.default_rng(seed)
    nod = dem == -100
    dem = np.floor(dem + rng.normal(0, 6.0, dem.shape).astype(np.float32)).astype(np.float32)
    dem[rng.random(dem.shape) < 0.02] -= 40
    dem[nod] = -100
    return dem


def example():
    from conftest import load_example
    return load_example()[0].astype(np.float32)


def bench(name, dem, px):
    H, W = dem.shape
    ctx = Context()
    d_dem, d_fill, d_fdr = ctx.to_device(dem), ctx.empty((H, W), np.float32), ctx.empty((H, W), np.uint8)
    info = (C.c_int32 * 3)()
    for _ in range(2):
        t0 = time.perf_counter()
        _lib.check(L.dt_dev_condition_d8(ctx.h, d_dem.ptr, H, W, px, d_fill.ptr, d_fdr.ptr, info))
        ctx.sync()
        t_sync = time.perf_counter() - t0
    budget = max(info[1], info[2]) + 4
    for _ in range(2):
        t0 = time.perf_counter()
        _lib.check(L.dt_dev_condition_d8_async(ctx.h, d_dem.ptr, H, W, px, d_fill.ptr, d_fdr.ptr, budget))
        ctx.sync()
        t_async = time.perf_counter() - t0
    assert ctx.status() == 0
    times = {}
    for cond in (False, True):
        ch = chain.Chain(H, W, ctx=ctx, px=px, condition=cond, condition_rounds=budget, tune_placement=False,
                         long_walks=cond)  # conditioned terrain has flats: downslope with the long-walk workspace
        for _ in range(3):
            ch.run(d_dem.ptr)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            ch.run(d_dem.ptr)
        ctx.sync()
        times[cond] = (time.perf_counter() - t0) / 5
        ch.check_status()
        ch.free()
    n = H * W
    print("%s %dx%d: %d fill rounds, %d flat rounds | synchronous %.2f ms (%.2f ns/cell) | asynchronous, budget %d rounds: "
          "%.2f ms (%.2f ns/cell) | chain %.2f ms, chain with conditioning %.2f ms"
          % (name, H, W, info[1], info[2], t_sync * 1e3, t_sync / n * 1e9, budget, t_async * 1e3, t_async / n * 1e9,
             times[False] * 1e3, times[True] * 1e3), flush=True)
    for b in (d_dem, d_fill, d_fdr):
        b.free()
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "sweeps":  # N sweeps: rounds of sweeps per visit, fill x flat
        n = int(sys.argv[1])
        dem_r, dem_e = rough(n), example()
        for fs in (1, 2, 3):
            for ls in (1, 2, 3, 6):
                L.dt_debug_set(6, fs)
                L.dt_debug_set(7, ls)
                print("fill sweeps %d, flat sweeps %d:" % (fs, ls), flush=True)
                bench("   rough synthetic DEM", dem_r, 10.0)
                bench("   Example DEM", dem_e, 12.5)
        L.dt_debug_set(6, 0)
        L.dt_debug_set(7, 0)
        sys.exit(0)
    if len(sys.argv) > 1:  # python tools/condition_bench.py N: the rough DEM at N x N only
        n = int(sys.argv[1])
        bench("rough synthetic DEM", rough(n), 10.0)
        sys.exit(0)
    bench("Example DEM", example(), 12.5)
    bench("rough synthetic DEM", rough(4096), 10.0)
