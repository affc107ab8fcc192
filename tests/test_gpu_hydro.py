"""GPU (-m gpu): hydrological conditioning (dt_hydro.hip) -- depression filling + flat resolution so
that the net-new D8 -> flow accumulation works on DEMs with pits and flats.  No reference counterpart exists (the
reference reads `fdr` from a GIS tool): the definition is the oracle's sequential priority flood + breadth-first flat
routing, which the kernels must reproduce bit for bit; the bundled Example rasters (produced by that GIS tool) anchor
how close the definition is to common practice."""
import numpy as np
import pytest

import oracle
from conftest import load_example

pytestmark = pytest.mark.gpu


def _check(dem, px):
    from descriptools_amd import flowacc, flowdir
    fdr, filled = flowdir.d8_conditioned(dem, px, return_filled=True)
    fdr_o, filled_o = oracle.condition_d8(dem, px)
    assert np.array_equal(filled, filled_o), "filled surface: exact (%d cells differ)" % int((filled != filled_o).sum())
    assert np.array_equal(fdr, fdr_o), "conditioned D8: exact (%d cells differ)" % int((fdr != fdr_o).sum())
    valid = dem != -100
    assert (fdr[valid] != 0).all(), "every valid cell has a code"
    acc = flowacc.accumulate(fdr, dem)
    assert (acc[valid] >= 0).all(), "no D8 cycle"
    return fdr, filled, acc


@pytest.mark.parametrize("H,W,seed", [(300, 417, 1), (64, 64, 2), (129, 1000, 3), (1, 77, 4), (50, 1, 5)])
def test_conditioning_on_rough_synthetic_terrain(H, W, seed):
    """tilted fBm with random pits, noise that breaks the monotone tilt, integer plateaus and nodata blobs"""
    rng = np.random.default_rng(seed)
    dem = oracle.synth_dem(seed, 2048, 2048, 11, 17, H, W, 3)
    nod = dem == -100
    dem = np.floor(dem + rng.normal(0, 6.0, dem.shape)).astype(np.float32)   # integer heights: large flats
    dem[rng.random(dem.shape) < 0.02] -= 40                                   # pits
    if H > 40 and W > 40:
        dem[10:30, 5:35] = dem[10:30, 5:35].min()                             # a lake-sized plateau
    dem[nod] = -100
    fdr, filled, acc = _check(dem, 10.0)
    assert (filled[~nod] >= dem[~nod]).all()
    assert min(H, W) < 3 or (filled[~nod] > dem[~nod]).sum() > 0, "pits were filled (every cell of a 1-D raster is an outlet)"
    if H > 40 and W > 40:
        sl, plain = oracle.slope_d8(filled, 10.0)
        assert ((plain == 0) & ~nod).sum() > 500, "the case has flats to resolve"


def test_conditioning_reproduces_the_bundled_rasters_in_the_main():
    """Example/input: 12_dem.tif is an (already filled) real DEM with 223,054 flat cells; 12_fdr.tif / 12_fac.tif come
    from the GIS tool the reference relies on.  Our conditioned D8 must give every valid cell a code without a
    cycle, and lands close to that tool's rasters (exact equality of the accumulation is not attainable: routing
    differs on 12 % of the flat cells, and the bundled accumulation includes inflow from outside the clipped basin)."""
    dem, fdr_ref, fac_ref, river, flood, klass = load_example()
    d = dem.astype(np.float32)
    fdr, filled, acc = _check(d, 12.5)
    valid = dem != -100
    assert np.array_equal(filled, d), "the bundled DEM has no depressions left"
    sl, plain = oracle.slope_d8(d, 12.5)
    flat = valid & (plain == 0)
    a_all = float((fdr[valid] == fdr_ref[valid]).mean())
    a_flat = float((fdr[flat] == fdr_ref[flat]).mean())
    both = (acc > 100) & (fac_ref > 100)
    med = float(np.median(np.abs(np.log(acc[both] / fac_ref[both]))))
    eq = float((acc[valid] == fac_ref[valid]).mean())
    print("conditioned D8 vs 12_fdr.tif: %.4f of valid cells, %.4f of the %d flat cells; accumulation equal on %.4f, "
          "median |ln ratio| where both > 100 cells: %.4f" % (a_all, a_flat, int(flat.sum()), eq, med))
    assert flat.sum() == 223054
    assert a_all > 0.96 and a_flat > 0.87 and eq > 0.74 and med < 0.02


def _serpentine(H, W, walls=1):
    """a channel that meanders through the whole raster between 1-cell walls: rows of channel alternate with rows of
    wall, joined at alternating ends.  Returns the boolean channel mask and the order of its cells along the path."""
    chan = np.zeros((H, W), bool)
    order = []
    left = True
    for y in range(1, H - 1, 1 + walls):
        xs = list(range(1, W - 1))
        if not left:
            xs.reverse()
        for x in xs:
            chan[y, x] = True
            order.append((y, x))
        left = not left
        if y + 1 + walls < H - 1:  # the connector through the wall rows, at the end the row finished on
            xe = xs[-1]
            for k in range(1, 1 + walls):
                chan[y + k, xe] = True
                order.append((y + k, xe))
    return chan, order


@pytest.fixture
def coloured_rounds():
    """the coloured form of the relaxation rounds (four launches per round, a 2 x 2 colouring of the tiles, activity
    flags in place: dt_hydro.hip hy_tile_of_block) is the default from 16384 tiles on; debug key 8 lowers the bar so that
    the small rasters of these tests run it"""
    from descriptools_amd import _lib
    L = _lib.lib()
    _lib.check(L.dt_debug_set(8, 1))
    yield
    _lib.check(L.dt_debug_set(8, 0))


@pytest.mark.parametrize("H,W,seed", [(300, 417, 1), (64, 64, 2), (129, 1000, 3), (1, 77, 4), (50, 1, 5), (700, 900, 6)])
def test_coloured_rounds_on_rough_terrain(coloured_rounds, H, W, seed):
    """the coloured rounds reach the same fixed points: rasters of one tile, one row / column of tiles (two of the four
    colours have no tile), odd and even tile counts; synchronous and asynchronous form"""
    import ctypes as C
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    rng = np.random.default_rng(seed)
    dem = oracle.synth_dem(seed, 2048, 2048, 11, 17, H, W, 3)
    nod = dem == -100
    dem = np.floor(dem + rng.normal(0, 6.0, dem.shape)).astype(np.float32)
    dem[rng.random(dem.shape) < 0.02] -= 40
    if H > 40 and W > 40:
        dem[10:30, 5:35] = dem[10:30, 5:35].min()
    dem[nod] = -100
    fdr, filled, _ = _check(dem, 10.0)
    L = _lib.lib()
    ctx = Context()
    d, f, c = ctx.to_device(dem), ctx.empty((H, W), np.float32), ctx.empty((H, W), np.uint8)
    _lib.check(L.dt_dev_condition_d8_async(ctx.h, d.ptr, H, W, 10.0, f.ptr, c.ptr, 60))
    assert ctx.status() == 0
    assert np.array_equal(f.to_host(), filled) and np.array_equal(c.to_host(), fdr)
    for b in (d, f, c):
        b.free()
    ctx.close()


def test_coloured_rounds_on_the_serpentine_and_the_spiral(coloured_rounds):
    test_conditioning_serpentine_depression_across_many_tiles()
    test_conditioning_spiral_flat_across_many_tiles()


@pytest.mark.parametrize("heights,widths", [([192, 130], [128, 250]), ([320], [128, 64, 130])])
def test_coloured_rounds_over_ranks(coloured_rounds, heights, widths):
    test_conditioning_tiled_over_ranks_equals_untiled(heights, widths)


def test_conditioning_serpentine_depression_across_many_tiles():
    """the spill path of a depression may cross tile borders far more often than the raster has tiles.
    A closed basin whose floor is a serpentine channel between high 1-cell walls (192 x 192 = 9 tiles; the channel
    crosses a 64-cell tile border ~190 times), rising towards its single outlet: filling it is a chain of thousands
    of dependent relaxations along the channel.  Must converge to the oracle's priority flood, bit for bit."""
    H = W = 192
    dem = np.full((H, W), 500.0, np.float32)  # walls and rim
    chan, order = _serpentine(H, W)
    n = len(order)
    # the channel floor RISES along the path towards the outlet at its far end, with pits every 7 cells: every pit
    # spills over the next sill, information travels the whole channel
    for k, (y, x) in enumerate(order):
        dem[y, x] = 10.0 + 0.01 * k - (3.0 if k % 7 == 3 else 0.0)
    ye, xe = order[-1]
    dem[ye:, xe] = np.minimum(dem[ye:, xe], 10.0 + 0.01 * n)  # cut the rim below the channel's end: the outlet
    from descriptools_amd import _lib
    import ctypes as C
    fdr, filled, acc = _check(dem, 10.0)
    assert (filled[chan] >= dem[chan]).all() and (filled[chan] > dem[chan]).sum() > n // 8
    info = (C.c_int32 * 3)()
    f2 = np.empty((H, W), np.uint8)
    _lib.check(_lib.lib().dt_d8_conditioned_f32(_lib.ptr(np.ascontiguousarray(dem), _lib.c_f32p), H, W, 10.0,
                                                _lib.ptr(f2, _lib.c_u8p), None, info))
    print("serpentine depression: %d fill rounds, %d flat rounds for %d tiles" % (info[1], info[2], 9))
    assert info[0] == 0 and np.array_equal(f2, fdr)


def test_conditioning_spiral_flat_across_many_tiles():
    """the same for the flat router: one perfectly flat serpentine channel (a single plateau 18,000 cells long, walls
    higher) with its only lower neighbour at one end -- the hop distances grow along the whole channel, through every
    tile many times."""
    H = W = 192
    dem = np.full((H, W), 500.0, np.float32)
    chan, order = _serpentine(H, W)
    dem[chan] = 100.0
    ye, xe = order[-1]
    dem[ye + 1:, xe] = 50.0  # below the channel's last cell: the plateau's only way out
    fdr, filled, acc = _check(dem, 10.0)
    assert np.array_equal(filled, dem)
    assert acc[ye + 1, xe] >= len(order), "the whole plateau drains through the gap below its far end"


def _rough(H, W, seed):
    rng = np.random.default_rng(seed)
    dem = oracle.synth_dem(seed, 2048, 2048, 11, 17, H, W, 3)
    nod = dem == -100
    dem = np.floor(dem + rng.normal(0, 6.0, dem.shape)).astype(np.float32)
    dem[rng.random(dem.shape) < 0.02] -= 40
    dem[10:30, 5:35] = dem[10:30, 5:35].min()
    dem[nod] = -100
    return dem


def test_async_conditioning_equals_the_synchronous_form_and_reports_an_exhausted_budget():
    """dt_dev_condition_d8_async: a fixed budget of rounds enqueued without a host synchronisation; same rasters as
    the synchronous form when the budget suffices, DT_STATUS_NOT_CONVERGED on the context when it does not"""
    from descriptools_amd import _lib, flowdir
    from descriptools_amd.device import Context
    L = _lib.lib()
    H, W, px = 300, 417, 10.0
    dem = _rough(H, W, 7)
    fdr_s, filled_s = flowdir.d8_conditioned(dem, px, return_filled=True)
    ctx = Context()
    d_dem, d_fill, d_fdr = ctx.to_device(dem), ctx.empty((H, W), np.float32), ctx.empty((H, W), np.uint8)
    _lib.check(L.dt_dev_condition_d8_async(ctx.h, d_dem.ptr, H, W, px, d_fill.ptr, d_fdr.ptr, 64))
    assert ctx.status() == 0
    assert np.array_equal(d_fill.to_host(), filled_s) and np.array_equal(d_fdr.to_host(), fdr_s)
    _lib.check(L.dt_dev_condition_d8_async(ctx.h, d_dem.ptr, H, W, px, d_fill.ptr, d_fdr.ptr, 1))  # one round: not enough
    with pytest.raises(RuntimeError, match="NOT_CONVERGED"):
        ctx.raise_on_status()
    assert ctx.status() == 0  # reading clears it
    for b in (d_dem, d_fill, d_fdr):
        b.free()
    ctx.close()


def test_chain_with_conditioning():
    """Chain(condition=True): the step's D8 codes are the conditioned ones (oracle.condition_d8), the accumulation has
    no cycle and no undrained pit, and every downstream raster is what the stand-alone API gives with that D8 raster."""
    from descriptools_amd import chain, downslope, flowhand
    from descriptools_amd.device import Context
    H, W, px = 384, 512, 10.0
    dem = _rough(H, W, 9)
    thr = H * W // 512
    ctx = Context()
    ch = chain.Chain(H, W, ctx=ctx, px=px, river_threshold=thr, condition=True)
    d_dem = ctx.to_device(dem)
    ch.run(d_dem.ptr)
    ch.check_status()
    out = {k: ch.buf[k].to_host() for k in ("fdr", "fac", "river", "fdist", "idx", "hand", "down", "filled", "slope")}
    fdr_o, filled_o = oracle.condition_d8(dem, px)
    assert np.array_equal(out["fdr"], fdr_o) and np.array_equal(out["filled"], filled_o)
    acc_o = oracle.flowacc(fdr_o, dem)
    valid = dem != -100
    assert np.array_equal(out["fac"], acc_o) and (acc_o[valid] >= 0).all()
    assert np.array_equal(out["river"], (acc_o > thr).astype(np.int8))
    fd, idx, hand = flowhand.flow_hand_index(dem, fdr_o, out["river"], px)
    assert np.array_equal(out["idx"], idx) and np.array_equal(out["hand"], hand) and np.array_equal(out["fdist"], fd)
    assert np.array_equal(out["down"], downslope.downsloper(dem, fdr_o, px, 5.0))
    assert np.array_equal(out["slope"], oracle.slope_d8(dem, px)[0])  # the slope is the raw DEM's
    # a budget that is too small is reported, not silently wrong
    ch2 = chain.Chain(H, W, ctx=ctx, px=px, river_threshold=thr, condition=True, condition_rounds=1)
    ch2.run(d_dem.ptr)
    with pytest.raises(RuntimeError, match="NOT_CONVERGED"):
        ch2.check_status()
    ch.free()
    ch2.free()
    d_dem.free()
    ctx.close()


@pytest.mark.parametrize("heights,widths", [([192, 130], [256, 200]), ([128, 128, 64], [192, 192])])
def test_conditioning_tiled_over_ranks_equals_untiled(heights, widths):
    """SURVEY.md 8f-4 tiled: N logical ranks iterate the two fixed points on their own windows with halo exchanges in
    between (tiling.condition_ranks); filled surface and conditioned D8 codes must be the single raster's, bit for bit
    -- including depressions and flats that span rank borders (the lake plateau and the pits straddle them)"""
    import torch
    from descriptools_amd import flowdir, tiling
    layout = tiling.Layout(heights, widths)
    Hg, Wg = layout.Hg, layout.Wg
    px = 10.0
    dem = _rough(Hg, Wg, 11)
    dem[heights[0] - 12:heights[0] + 12, widths[0] - 20:widths[0] + 20] = dem[heights[0] - 12:heights[0] + 12,
                                                                               widths[0] - 20:widths[0] + 20].min()
    dem[dem == -100] = -100
    fdr_u, filled_u = flowdir.d8_conditioned(dem, px, return_filled=True)
    h = tiling.HALO
    pad = np.zeros((Hg + 2 * h, Wg + 2 * h), np.float32)
    pad[h:h + Hg, h:h + Wg] = dem
    tiles = []
    for r in range(layout.size):
        t = tiling.RankTile(layout, r, device=0, px=px, river_threshold=10)
        y0, x0 = layout.origin(r)
        t.set_dem_ext(pad[y0:y0 + t.He, x0:x0 + t.We])
        tiles.append(t)
    left, it_fill, it_flat = tiling.condition_local(tiles, layout)
    assert left == 0 and it_fill >= 2 and it_flat >= 1
    for t in tiles:
        y0, x0 = layout.origin(t.rank)
        sl = (slice(y0, y0 + t.H), slice(x0, x0 + t.W))
        assert np.array_equal(t.host("filled"), filled_u[sl]), "rank %d filled surface" % t.rank
        assert np.array_equal(t.host("fdr"), fdr_u[sl]), "rank %d conditioned D8" % t.rank
    # the halos hold the neighbours' conditioned codes: the step that follows needs them (downslope's margin)
    t0 = tiles[0]
    ext = t0.t["fdr"].cpu().numpy()
    assert np.array_equal(ext[h:h + t0.H, h + t0.W:h + t0.W + 8], fdr_u[0:t0.H, t0.W:t0.W + 8])
    for t in tiles:
        t.free()
    torch.cuda.empty_cache()


def test_example_conditioned_chain_in_ranks_equals_untiled():
    """REAL terrain end to end over ranks: the bundled Example DEM, conditioned over 2 x 2 logical ranks
    (tiling.condition_local), the rank step on the conditioned codes (flats and valley floors: flow paths and downslope
    walks of thousands of cells across the rank borders), the walks that leave a rank finished as walkers
    (tiling.finish_downslope) -- every raster equals the untiled chain's (Chain(condition=True, long_walks=True))"""
    import threading
    import torch
    from conftest import load_example
    from descriptools_amd import chain, tiling
    from descriptools_amd.device import Context
    dem = np.asarray(load_example()[0], np.float32)
    Hg, Wg = dem.shape
    px = 12.5
    ctx = Context()
    d = ctx.to_device(dem)
    ch = chain.Chain(Hg, Wg, ctx=ctx, px=px, condition=True, condition_rounds=96, long_walks=True, tune_placement=False)
    ch.run(d.ptr)
    ch.check_status()
    names = ["fdr", "fac", "river", "fdist", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"]
    ref = {k: ch.buf[k].to_host() for k in names}
    ch.free()
    d.free()
    ctx.close()
    layout = tiling.Layout([1088, Hg - 1088], [768, Wg - 768])
    h = tiling.HALO
    pad = np.zeros((Hg + 2 * h, Wg + 2 * h), np.float32)
    pad[h:h + Hg, h:h + Wg] = dem
    tiles = []
    for r in range(layout.size):
        t = tiling.RankTile(layout, r, device=0, px=px)
        y0, x0 = layout.origin(r)
        t.set_dem_ext(pad[y0:y0 + t.He, x0:x0 + t.We])
        tiles.append(t)
    left, _, _ = tiling.condition_local(tiles, layout)
    assert left == 0
    tiling.simulate_dev(tiles, layout, d8=False)
    marked = sum(t.unresolved_downslope() for t in tiles)
    assert marked > 0, "the Example's valley floors send walks across the rank borders"
    comms = tiling.LocalComm.create(layout.size)
    errors = []

    def work(r):
        try:
            tiling.finish_downslope(tiles[r], comms[r])
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append(e)
            comms[r].sh.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(layout.size)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in tiles:
        t.check_status()
        y0, x0 = layout.origin(t.rank)
        sl = (slice(y0, y0 + t.H), slice(x0, x0 + t.W))
        for k in names:
            got, want = t.host(k), ref[k][sl]
            assert np.array_equal(got, want.astype(got.dtype), equal_nan=True), (t.rank, k, int((got != want).sum()))
        t.free()
    torch.cuda.empty_cache()


def test_run_host_with_conditioning_on_the_example():
    """chain.run_host(dem, px, condition=True): conditioning, the chain on the conditioned codes and the long downslope
    walks (queued, finished with skip tables) in one host call -- the rasters of Chain(condition=True, long_walks=True);
    a budget of rounds that is too small raises instead of returning unconditioned rasters"""
    from conftest import load_example
    from descriptools_amd import chain
    from descriptools_amd.device import Context
    dem = np.asarray(load_example()[0], np.float32)
    H, W = dem.shape
    out = chain.run_host(dem, 12.5, condition=True, condition_rounds=96)
    ctx = Context()
    d = ctx.to_device(dem)
    ch = chain.Chain(H, W, ctx=ctx, px=12.5, condition=True, condition_rounds=96, long_walks=True, tune_placement=False)
    ch.run(d.ptr)
    ch.check_status()
    for k in ("fdr", "fac", "hand", "down", "gfi"):
        ref = ch.buf[k].to_host()
        assert np.array_equal(out[k], ref.astype(out[k].dtype), equal_nan=True), k
    ch.free()
    d.free()
    ctx.close()
    with pytest.raises(RuntimeError, match="NOT_CONVERGED"):
        chain.run_host(dem, 12.5, condition=True, condition_rounds=3)
