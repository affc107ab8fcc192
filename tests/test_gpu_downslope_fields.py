"""GPU (-m gpu): the windowed downslope kernel on rasters of several windows against the oracle's literal walk
(downslope.py:161-314 + :435-532): arbitrary direction fields on integer / fixed-point heights, terrain with NaN,
heights far below the nodata value and 1000 m steps, DEMs off any quantum, an int16-like DEM with the raw -50 marks.
(Written for a trial of an integer window walk -- profiles/r3/downslope_quantised_trial.txt -- and kept as parity
cases for the float kernel.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import oracle
    import descriptools_amd.downslope as downslope
    from descriptools_amd import _lib
    return oracle, downslope, _lib.lib()


def _same(got, want):
    assert np.array_equal(got, want, equal_nan=True), int((got != want).sum())


@pytest.mark.parametrize("quantum", [1.0, 1.0 / 256, 8.0])
def test_arbitrary_direction_field_on_a_quantum(env, quantum):
    """every exit of the window walk: drop reached, non-D8 codes, window ring, the 256-move limit, stepping onto
    nodata, start cells at / below the nodata value, the unsafe-rounding recheck"""
    oracle, downslope, L = env
    rng = np.random.default_rng(int(quantum * 256))
    H, W = 420, 900
    codes = np.array([1, 2, 4, 8, 16, 32, 64, 128, 0, 3, 255], np.uint8)
    fdr = codes[rng.integers(0, len(codes), size=(H, W))]
    fdr[rng.random((H, W)) < 0.5] = 4
    dem = (rng.integers(0, int(48 / quantum) + 2, size=(H, W)) * quantum).astype(np.float32)
    dem[:, 500:] = np.float32(8.0)               # flat part: walks run until a cycle, the edge or the cap
    dem[rng.random((H, W)) < 0.02] = -100
    dem[200:203, 300:303] = np.float32(-104.0)   # below the nodata value but not equal to it (on every quantum here)
    for dz in (5.0, 0.3, 0.0):
        _same(downslope.downsloper(dem, fdr, 10.0, dz), oracle.downslope(dem, fdr, 10.0, dz))


def test_terrain_with_windows_handed_back(env):
    """2^-8 m terrain with nodata blobs, one height off the quantum, a plateau lifted by 1000 m, a NaN, a huge negative
    height"""
    oracle, downslope, _ = env
    H, W = 1100, 1500
    dem = oracle.synth_dem(7, 4096, 4096, 300, 500, H, W, 3)
    dem[400, 700] += np.float32(2.0 ** -12)
    dem[600:700, 200:330] += np.float32(1000.0)
    dem[900, 1200] = np.nan
    dem[150, 1300] = np.float32(-9999.0)
    _, fdr = oracle.slope_d8(np.nan_to_num(dem, nan=0.0), 10.0)
    for dz in (5.0, 1.0):
        _same(downslope.downsloper(dem, fdr, 10.0, dz), oracle.downslope(dem, fdr, 10.0, dz))


def test_dem_off_any_quantum(env):
    oracle, downslope, _ = env
    rng = np.random.default_rng(3)
    H, W = 300, 700
    dem = (oracle.synth_dem(5, 2048, 2048, 0, 0, H, W, 2) + rng.random((H, W)).astype(np.float32) * np.float32(0.01))
    dem[oracle.synth_dem(5, 2048, 2048, 0, 0, H, W, 2) == -100] = -100
    _, fdr = oracle.slope_d8(dem, 12.5)
    _same(downslope.downsloper(dem, fdr, 12.5, 5.0), oracle.downslope(dem, fdr, 12.5, 5.0))


def test_int16_example_like_dem(env):
    """integer metres (qe >= 0), the Example's dtype: D8 from the DEM itself, raw -50 marks included"""
    oracle, downslope, _ = env
    H, W = 700, 1300
    dem = np.floor(oracle.synth_dem(11, 2048, 2048, 100, 200, H, W, 2) * 0.5).astype(np.float32)
    dem[oracle.synth_dem(11, 2048, 2048, 100, 200, H, W, 2) == -100] = -100
    _, fdr = oracle.slope_d8(dem, 30.0)
    want = oracle.downslope(dem, fdr, 30.0, 5.0)
    _same(downslope.downsloper(dem, fdr, 30.0, 5.0), want)
    raw = downslope.downslope_cpu(dem, fdr, 30.0, 5.0)
    keep = raw != -50
    _same(raw[keep], want[keep])


def test_long_walks_with_the_skip_table(env):
    """the long-walk workspace (dt_dev_downslope_lift; the host-tier downsloper uses it): terrain whose walks run for
    thousands of moves -- a tilted plane of 1e-3 per cell drained by a serpentine channel, flats, a lake of NaN-free
    equal heights, walks that hit the 5000-move cap exactly, walks that fail on a non-D8 code far from their start --
    against the oracle's literal walk, and the device entry with and without the workspace bit for bit"""
    oracle, downslope, L = env
    import torch
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    rng = np.random.default_rng(11)
    H, W = 400, 600
    yy, xx = np.mgrid[0:H, 0:W]
    # gentle plane: 5 m of drop need ~5000 cardinal moves -- walks of every length up to and beyond the cap
    dem = (200.0 - 0.001 * xx - 0.0002 * yy).astype(np.float32)
    fdr = np.full((H, W), 1, np.uint8)            # everything flows east ...
    fdr[:, W - 1] = 4                             # ... then south along the last column
    fdr[H - 1, :] = 16                            # ... and back west along the last row (a long way round)
    fdr[H - 1, 0] = 0
    dem[150:180, 100:500] = np.float32(150.0)     # a flat well below its surroundings: walks end at its rim
    fdr[rng.random((H, W)) < 0.0005] = 3          # non-D8 codes: walks fail there, far from where they started
    dem[rng.random((H, W)) < 0.0005] = -100       # nodata ahead stops a walk too
    for dz in (5.0, 0.25, 4.99):
        _same(downslope.downsloper(dem, fdr, 1.0, dz), oracle.downslope(dem, fdr, 1.0, dz))
    raw = downslope.downslope_cpu(dem, fdr, 1.0, 5.0)
    want = oracle.downslope(dem, fdr, 1.0, 5.0)
    keep = raw != -50
    _same(raw[keep], want[keep])
    # device entries: with / without the workspace
    ctx = Context()
    d, f = ctx.to_device(dem), ctx.to_device(fdr)
    a, b = ctx.empty((H, W), np.float32), ctx.empty((H, W), np.float32)
    nb = int(L.dt_downslope_lift_workspace(H, W))
    work = ctx.empty((nb,), np.uint8)
    _lib.check(L.dt_dev_downslope(ctx.h, d.ptr, f.ptr, H, W, 1.0, 5.0, 0, a.ptr))
    _lib.check(L.dt_dev_downslope_lift(ctx.h, d.ptr, f.ptr, H, W, 1.0, 5.0, 0, b.ptr, work.ptr, nb))
    ctx.sync()
    ra, rb = a.to_host(), b.to_host()
    assert np.array_equal(ra.view(np.int32), rb.view(np.int32)) and np.array_equal(ra, want, equal_nan=True)
    # nearly every walk of this plane is a long one: more than the queue holds (half the cells), so this case also
    # covers the walks a full queue leaves to the window kernel
    queued = int(work.to_host()[:4].view(np.uint32)[0])
    assert queued > (H * W + 1) // 2, queued
    for x in (d, f, a, b, work):
        x.free()
    ctx.close()
    torch.cuda.empty_cache()


def test_walks_that_end_with_a_drop_of_exactly_zero(env):
    """flats that drain into nodata, off the raster's edge or onto a non-D8 code: the walk fails with a drop of exactly
    zero and the reference stores 0 / distance = 0.  (Until round 4 the count form's rounding test sent every such
    walk to the move-by-move form -- correct, and 6 of the 6.6 ms of the long-walk kernel on the tiled Example; the
    exemption is ds_quotient's.)  Short walks (window kernel) and long ones (queue + skip tables), integer metres like
    the Example and a height of -0.0, against the oracle's literal walk."""
    oracle, downslope, L = env
    rng = np.random.default_rng(5)
    H, W = 330, 1400
    dem = np.full((H, W), 37.0, np.float32)
    fdr = np.full((H, W), 1, np.uint8)             # east, row by row: walks of 1 .. 1399 moves, all on one height
    fdr[::3, :] = 16                               # every third row west (off the raster's edge at column 0)
    dem[:, 900] = -100                             # a nodata column ahead of the walks from both sides
    fdr[100:130, 400] = 7                          # a non-D8 code: walks spin there (failed, drop 0)
    dem[200:260, :] = np.float32(-0.0)             # 0 - (-0) and (-0) - 0: the sign of the zero is the reference's
    dem[230:260, 300:600] = np.float32(0.0)
    dem[rng.random((H, W)) < 0.0003] = -100
    dem[300:, :] = np.floor(rng.random((H - 300, W)) * 3).astype(np.float32) + 50   # and a strip that does drop
    for dz in (5.0, 0.5):
        got, want = downslope.downsloper(dem, fdr, 12.5, dz), oracle.downslope(dem, fdr, 12.5, dz)
        _same(got, want)
        assert np.array_equal(np.signbit(got), np.signbit(want))
    assert (oracle.downslope(dem, fdr, 12.5, 5.0)[:200] == 0).mean() > 0.9


def test_walks_that_end_at_nodata_or_at_an_infinite_height(env):
    """the walk never moves onto nodata: the reference stops one move earlier and keeps the walk so far
    (downslope.py:231-281).  The kernels with the queue mark the cells whose successor is nodata while they stage a
    window (round 4: no second walk on global memory); a height of -inf is staged like nodata but is NOT nodata -- the
    walk steps onto it and ends with an infinite drop.  Streams that drain into nodata blobs after 1 .. 60 moves, a
    nodata start cell's neighbours, -inf pits; host tier (queue + tables) and the plain device entry."""
    oracle, downslope, L = env
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    rng = np.random.default_rng(21)
    H, W = 500, 900
    yy, xx = np.mgrid[0:H, 0:W]
    dem = (300.0 - 0.05 * xx + 0.01 * yy + rng.random((H, W)) * 0.02).astype(np.float32)   # gentle: walks of ~100 moves
    fdr = np.full((H, W), 1, np.uint8)
    fdr[rng.random((H, W)) < 0.2] = 2                 # some diagonal moves
    for _ in range(150):                              # nodata blobs the streams run into
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, 6)
        dem[max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = -100
    for _ in range(40):                               # and heights of -inf (not nodata)
        dem[rng.integers(0, H), rng.integers(0, W)] = -np.inf
    for dz in (5.0, 1.0):
        want = oracle.downslope(dem, fdr, 12.5, dz)
        with np.errstate(invalid="ignore"):
            got = downslope.downsloper(dem, fdr, 12.5, dz)
        _same(got, want)
    ctx = Context()
    d, f, a = ctx.to_device(dem), ctx.to_device(fdr), ctx.empty((H, W), np.float32)
    _lib.check(L.dt_dev_downslope(ctx.h, d.ptr, f.ptr, H, W, 12.5, 5.0, 0, a.ptr))
    ctx.sync()
    _same(a.to_host(), oracle.downslope(dem, fdr, 12.5, 5.0))
    for x in (d, f, a):
        x.free()
    ctx.close()


def test_chain_with_long_walks_on_the_example():
    """Chain(long_walks=True) on the bundled Example (its GIS D8 raster has the flats): same downslope raster"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_example
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    L = _lib.lib()
    ex = load_example()
    dem, fdr = np.asarray(ex[0], np.float32), np.ascontiguousarray(ex[1], np.uint8)
    H, W = dem.shape
    ctx = Context()
    d, f = ctx.to_device(dem), ctx.to_device(fdr)
    a, b = ctx.empty((H, W), np.float32), ctx.empty((H, W), np.float32)
    nb = int(L.dt_downslope_lift_workspace(H, W))
    work = ctx.empty((nb,), np.uint8)
    _lib.check(L.dt_dev_downslope(ctx.h, d.ptr, f.ptr, H, W, 12.5, 5.0, 0, a.ptr))
    _lib.check(L.dt_dev_downslope_lift(ctx.h, d.ptr, f.ptr, H, W, 12.5, 5.0, 0, b.ptr, work.ptr, nb))
    ctx.sync()
    assert np.array_equal(a.to_host().view(np.int32), b.to_host().view(np.int32))
    for x in (d, f, a, b, work):
        x.free()
    ctx.close()


def test_run_host_finishes_long_walks_by_itself():
    """chain.run_host (long_walks="auto"): the walks are queued by the step and finished before the rasters come back
    -- with tables on terrain full of long walks, without on terrain that has none; same downslope raster as the
    host-tier function in both cases"""
    import oracle
    import descriptools_amd.downslope as downslope
    from descriptools_amd import chain
    H, W = 400, 600
    yy, xx = np.mgrid[0:H, 0:W]
    gentle = (200.0 - 0.001 * xx - 0.0002 * yy).astype(np.float32)   # every walk is long
    rough = oracle.synth_dem(3, 2048, 2048, 100, 100, H, W, 2)        # none is
    for dem in (gentle, rough):
        out = chain.run_host(dem, 1.0)
        _same(out["down"], downslope.downsloper(dem, out["fdr"], 1.0, 5.0))
        _same(out["down"], oracle.downslope(dem, out["fdr"], 1.0, 5.0))
