"""GPU (-m gpu): the multi-rank tiling run as N logical ranks on ONE device (the all-gathers replaced by
list collection, everything else identical to the torch.distributed path) must reproduce the untiled
single-GPU chain bit for bit -- the contract "tiled == untiled" (SURVEY.md 5 / 8e)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

NAMES = ["fdr", "fac", "river", "fdist", "hand", "a_river", "slope", "ti", "mti", "gfi", "lnhlh", "down"]


def _global_dem(seed, Hg, Wg, nod):
    return oracle.synth_dem(seed, 2048, 2048, 300, 200, Hg, Wg, nod)


@pytest.mark.parametrize("solver", ["device", "numpy"])
@pytest.mark.parametrize("heights,widths,nod,seed", [
    ([192, 192], [256, 256], 0, 1),        # 2 x 2
    ([384], [192, 320], 3, 2),             # 1 x 2, uneven, nodata blobs
    ([128, 256], [512], 0, 3),             # 2 x 1
    ([192, 130], [256, 200], 2, 6),        # 2 x 2, ragged last row / column
    ([128, 128, 128], [128, 256, 128], 2, 4),  # 3 x 3
    ([384], [512], 0, 5),                  # 1 x 1 through the windowed entry points
])
def test_tiled_equals_untiled(heights, widths, nod, seed, solver, acc64=None):
    """solver "device" = the product's rank-level solves (dt_dev_rank_solve_*), "numpy" = their host
    restatement in tiling.py (test infrastructure for the gloo rehearsals).  acc64: the ranks keep the flow
    accumulation as int64 rasters (what a global raster of >= 2^31 cells gets), forced here at small sizes."""
    import torch
    from descriptools_amd import chain, tiling
    layout = tiling.Layout(heights, widths)
    Hg, Wg = layout.Hg, layout.Wg
    dem = _global_dem(seed, Hg, Wg, nod)
    px, thr = 10.0, (Hg * Wg) // 512
    ref = chain.run_host(dem, px, river_threshold=thr)
    h = tiling.HALO
    pad = np.full((Hg + 2 * h, Wg + 2 * h), np.nan, np.float32)  # NaN = must never be read
    pad[h:h + Hg, h:h + Wg] = dem
    tiles = []
    for r in range(layout.size):
        # the river index raster in both widths (int64 is what a global raster beyond 2^31 cells gets)
        t = tiling.RankTile(layout, r, device=0, px=px, river_threshold=thr, idx64=(seed % 2 == 0), acc64=acc64)
        assert t.t["fac"].dtype == (torch.int64 if acc64 else torch.int32)
        y0, x0 = layout.origin(r)
        t.set_dem_ext(pad[y0:y0 + t.He, x0:x0 + t.We])
        tiles.append(t)
    (tiling.simulate_dev if solver == "device" else tiling.simulate)(tiles, layout)
    for t in tiles:
        assert t.unresolved_downslope() == 0
        y0, x0 = layout.origin(t.rank)
        sl = (slice(y0, y0 + t.H), slice(x0, x0 + t.W))
        for name in NAMES:
            got, want = t.host(name), ref[name][sl]
            assert np.array_equal(got, want.astype(got.dtype), equal_nan=True), \
                "rank %d %s: %d cells differ" % (t.rank, name, int((got != want).sum()))
        # river index: global flat index of the untiled run
        assert t.host("idx").dtype == (np.int64 if seed % 2 == 0 else np.int32)
        assert np.array_equal(t.host("idx"), ref["idx"][sl])


def test_tiled_flowacc_cycles_across_ranks():
    """arbitrary direction field (cycles inside tiles, across tiles and across ranks): the tiled
    flow accumulation equals the oracle's."""
    from descriptools_amd import tiling
    import torch
    rng = np.random.default_rng(11)
    layout = tiling.Layout([128, 128], [128, 64, 64])
    Hg, Wg = layout.Hg, layout.Wg
    codes = np.array([1, 2, 4, 8, 16, 32, 64, 128], np.uint8)
    fdr = codes[rng.integers(0, 8, size=(Hg, Wg))]
    fdr[rng.random((Hg, Wg)) < 0.3] = 4
    ref = oracle.flowacc(fdr)
    h = tiling.HALO
    pad = np.zeros((Hg + 2 * h, Wg + 2 * h), np.uint8)
    pad[h:h + Hg, h:h + Wg] = fdr
    tiles = []
    for r in range(layout.size):
        t = tiling.RankTile(layout, r, device=0, river_threshold=10)
        y0, x0 = layout.origin(r)
        t.t["fdr"].copy_(torch.as_tensor(pad[y0:y0 + t.He, x0:x0 + t.We]))
        t.t["dem"].fill_(1.0)
        tiles.append(t)
    torch.cuda.synchronize()
    fa = [tuple(a.cpu().numpy() for a in t.fa_local()) for t in tiles]
    ext = tiling.solve_flowacc(layout, fa)
    for t in tiles:
        t.fill_ring_codes()
    rows = torch.cat([t.fa_row for t in tiles])
    torch.cuda.synchronize()
    for t in tiles:
        y0, x0 = layout.origin(t.rank)
        want = ref[y0:y0 + t.H, x0:x0 + t.W]
        t.fa_finish(ext[t.rank])
        got = t.host("fac")
        assert np.array_equal(got, want), "rank %d: %d cells differ" % (t.rank, int((got != want).sum()))
        t.t["fac"].zero_()
        torch.cuda.synchronize()
        t.fa_local()             # the inflow is added on top of passes 1-2: redo them
        t.fa_solve_finish(rows)  # the rank-level solve on the GPU
        got = t.host("fac")
        assert np.array_equal(got, want), "rank %d (device solve): %d cells differ" % (
            t.rank, int((got != want).sum()))


def test_tiled_evaluation_matches_untiled():
    """config #5's classifier step on a tiled HAND: per-rank np.unique extremes combined as the
    all-gather would (combine_extremes) and per-rank confusion counts summed as the all-reduce would equal
    the untiled device evaluation."""
    import ctypes as C
    from descriptools_amd import _lib, chain, evaluation, tiling
    from descriptools_amd.device import Context
    L = _lib.lib()
    layout = tiling.Layout([192, 192], [256, 256])
    Hg, Wg = layout.Hg, layout.Wg
    dem = _global_dem(7, Hg, Wg, 2)
    ref = chain.run_host(dem, 10.0, river_threshold=(Hg * Wg) // 512)
    hand = ref["hand"]
    rng = np.random.default_rng(3)
    flood = ((hand >= 0) & (hand < 2.0) & (rng.random(hand.shape) < 0.9)).astype(np.int8)
    ctx = Context()
    d_h, d_f = ctx.to_device(hand), ctx.to_device(flood)
    want = evaluation.evaluate_resident(ctx, d_h.ptr, d_f.ptr, hand.size, 'under')
    d_h.free()
    d_f.free()
    bufs = []
    for r in range(layout.size):
        y0, x0 = layout.origin(r)
        H, W = layout.shape(r)
        bufs.append((ctx.to_device(np.ascontiguousarray(hand[y0:y0 + H, x0:x0 + W])),
                     ctx.to_device(np.ascontiguousarray(flood[y0:y0 + H, x0:x0 + W]))))
    ext = []
    for dh, _ in bufs:
        e = ctx.empty(3, np.float32)
        _lib.check(L.dt_dev_unique_extremes_f32(ctx.h, dh.ptr, dh.size, e.ptr))
        ext.append(e.to_host())
        e.free()
    glob = evaluation.combine_extremes(ext)
    assert float(glob[1]) == want["mn"] and float(glob[2]) == want["mx"]
    # value of the scaled raster at global [0, 0] = binary_map's nodata (evaluation.py:111)
    first = np.nan if hand[0, 0] == -100 else float((np.float32(hand[0, 0]) - glob[1]) / (glob[2] - glob[1]))
    # every rank runs the same calibration with the counts summed over ranks at each stage
    results = []
    for i in range(layout.size):
        def allreduce(c_local, i=i):
            tot = c_local.copy()
            for j, (dh, df) in enumerate(bufs):
                if j == i:
                    continue
                dd = ctx.empty(dh.size, np.float64)
                _lib.check(L.dt_dev_minmax_scale_f32(ctx.h, dh.ptr, dh.size, glob[1], glob[2], np.float32(-100), dd.ptr))
                cc = ctx.empty(4 * len(c_local), np.int64)
                th = np.ascontiguousarray(allreduce.th)
                _lib.check(L.dt_dev_confusion_multi(ctx.h, dd.ptr, df.ptr, dh.size, first,
                                                    th.ctypes.data_as(C.POINTER(C.c_double)), len(th), 1, cc.ptr))
                tot += cc.to_host().reshape(len(th), 4)
                dd.free()
                cc.free()
            return tot
        # the thresholds of the current stage are needed by the emulated peers: wrap dt_dev_confusion_multi
        orig = L.dt_dev_confusion_multi

        def spy(ctxh, d, f, n, nod, thp, nth, under, cnt, _orig=orig, _ar=allreduce):
            _ar.th = np.ctypeslib.as_array(thp, (nth,)).copy()
            return _orig(ctxh, d, f, n, nod, thp, nth, under, cnt)
        dh, df = bufs[i]
        L_spy = type("LibSpy", (), {"__getattr__": lambda self, k: spy if k == "dt_dev_confusion_multi" else getattr(L, k)})()
        old = _lib._lib
        _lib._lib = L_spy
        try:
            results.append(evaluation.evaluate_resident(ctx, dh.ptr, df.ptr, dh.size, 'under',
                                                        reduce_extremes=lambda e: glob, reduce_counts=allreduce,
                                                        nodata_first=first))
        finally:
            _lib._lib = old
    for res in results:
        assert res["threshold"] == want["threshold"]
        assert np.array_equal(res["counts"], want["counts"])
        assert res["fit"] == want["fit"] and res["correctness"] == want["correctness"]
    for dh, df in bufs:
        dh.free()
        df.free()
    ctx.close()


def _boustrophedon(H, W, period):
    """east on even rows, west on odd rows, south at the row ends: ONE path through the whole raster; a river cell
    at the end of every `period`-th row cuts it into stretches of period * W moves."""
    fdr = np.zeros((H, W), np.uint8)
    fdr[0::2, :] = 1
    fdr[1::2, :] = 16
    fdr[0::2, W - 1] = 4
    fdr[1::2, 0] = 4
    river = np.zeros((H, W), np.int8)
    for y in range(period - 1, H, period):
        river[y, W - 1 if y % 2 == 0 else 0] = 1
    return fdr, river


def test_move_cap_through_tiles_and_ranks():
    """The 20000-move cap (flowhand.py:834-837) on paths that cross hundreds of tiles and several ranks: a
    2048-wide boustrophedon whose river cells sit 24576 moves apart, so the first 4576 cells of every stretch exceed
    the cap and must come out -100, the others with their exact move counts -- untiled (tile -> perimeter
    hierarchy) and as 2 x 2 logical ranks (tile -> perimeter -> rank), against the oracle's O(N) solve."""
    import torch
    from descriptools_amd import flowhand, tiling
    H = W = 2048
    px = 10.0
    fdr, river = _boustrophedon(H, W, 12)
    idx_o, nc, nd = oracle.flowhand_fast(fdr, river)
    ok = idx_o != -100
    assert (~ok).sum() > 0.15 * H * W and ok.sum() > 0.7 * H * W, "both sides of the cap are exercised"
    d_o = np.where(ok, px * nc + (px * np.sqrt(2.0)) * nd, -100.0).astype(np.float32)
    assert nc[ok].max() == 20000
    dem = np.ones((H, W), np.float32)
    fd, idx, _ = flowhand.flow_hand_index(dem, fdr, river, px)
    assert np.array_equal(idx, idx_o) and np.array_equal(fd, d_o)
    layout = tiling.Layout([1024, 1024], [1024, 1024])
    h = tiling.HALO
    pf = np.zeros((H + 2 * h, W + 2 * h), np.uint8)
    pr = np.zeros((H + 2 * h, W + 2 * h), np.int8)
    pf[h:h + H, h:h + W], pr[h:h + H, h:h + W] = fdr, river
    tiles = []
    for r in range(layout.size):
        t = tiling.RankTile(layout, r, device=0, px=px, river_threshold=10)
        y0, x0 = layout.origin(r)
        t.t["fdr"].copy_(torch.as_tensor(pf[y0:y0 + t.He, x0:x0 + t.We]))
        t.t["river"].copy_(torch.as_tensor(pr[y0:y0 + t.He, x0:x0 + t.We]))
        t.t["dem"].fill_(1.0)
        tiles.append(t)
    torch.cuda.synchronize()
    for t in tiles:
        t.fill_ring_codes()
        t.fh_local(sync=False)
    for t in tiles:
        t.ctx.sync()
    rows = torch.cat([t.fh_row for t in tiles])
    torch.cuda.synchronize()
    for t in tiles:
        t.fh_solve_finish(rows)
    for t in tiles:
        y0, x0 = layout.origin(t.rank)
        sl = (slice(y0, y0 + t.H), slice(x0, x0 + t.W))
        assert np.array_equal(t.host("idx"), idx_o[sl]), "rank %d river index" % t.rank
        assert np.array_equal(t.host("fdist"), d_o[sl]), "rank %d flow distance" % t.rank


@pytest.mark.parametrize("heights,widths,nod,seed", [
    ([192, 192], [128, 192, 128, 192], 0, 8),       # config #5's 2 x 4 rank grid
    ([256, 150], [128, 128, 192, 100], 3, 9),       # 2 x 4, ragged last row / column, nodata blobs
])
def test_tiled_2x4_equals_untiled(heights, widths, nod, seed):
    test_tiled_equals_untiled(heights, widths, nod, seed, "device")


@pytest.mark.parametrize("solver", ["device", "numpy"])
@pytest.mark.parametrize("heights,widths,nod,seed", [
    ([192, 192], [256, 256], 0, 1),
    ([192, 130], [256, 200], 2, 6),                 # ragged, nodata
    ([256, 150], [128, 128, 192, 100], 3, 9),       # 2 x 4
    ([384], [512], 0, 5),                           # 1 x 1
])
def test_tiled_int64_accumulation_equals_untiled(heights, widths, nod, seed, solver):
    """the int64 accumulation path (every `_a64` entry point) against the untiled chain, raster for raster"""
    test_tiled_equals_untiled(heights, widths, nod, seed, solver, acc64=True)


def test_tiled_2x4_flowacc_cycles_across_ranks():
    """arbitrary direction field on the 2 x 4 grid: cycles inside tiles, across tiles and across ranks"""
    from descriptools_amd import tiling
    import torch
    rng = np.random.default_rng(21)
    layout = tiling.Layout([128, 128], [64, 128, 64, 128])
    Hg, Wg = layout.Hg, layout.Wg
    codes = np.array([1, 2, 4, 8, 16, 32, 64, 128], np.uint8)
    fdr = codes[rng.integers(0, 8, size=(Hg, Wg))]
    fdr[rng.random((Hg, Wg)) < 0.3] = 1
    ref = oracle.flowacc(fdr)
    h = tiling.HALO
    pad = np.zeros((Hg + 2 * h, Wg + 2 * h), np.uint8)
    pad[h:h + Hg, h:h + Wg] = fdr
    tiles = []
    for r in range(layout.size):
        t = tiling.RankTile(layout, r, device=0, river_threshold=10)
        y0, x0 = layout.origin(r)
        t.t["fdr"].copy_(torch.as_tensor(pad[y0:y0 + t.He, x0:x0 + t.We]))
        t.t["dem"].fill_(1.0)
        tiles.append(t)
    torch.cuda.synchronize()
    for t in tiles:
        t.fa_local(sync=False)
        t.fill_ring_codes()
    for t in tiles:
        t.ctx.sync()
    rows = torch.cat([t.fa_row for t in tiles])
    torch.cuda.synchronize()
    for t in tiles:
        t.fa_solve_finish(rows)
        y0, x0 = layout.origin(t.rank)
        got, want = t.host("fac"), ref[y0:y0 + t.H, x0:x0 + t.W]
        assert np.array_equal(got, want), "rank %d: %d cells differ" % (t.rank, int((got != want).sum()))


def test_accumulation_overflow_is_reported():
    """int32 accumulation rasters: an inflow that takes a value to 2^31 raises the context's status instead of
    wrapping silently (DT_STATUS_ACC_OVERFLOW); just below the limit it does not."""
    import torch
    from descriptools_amd import tiling
    layout = tiling.Layout([128], [128, 128])
    for inflow, expect in ((2 ** 31 - 20000, False), (2 ** 31 - 100, True), (2 ** 33, True)):
        t = tiling.RankTile(layout, 1, device=0, river_threshold=10)
        t.t["fdr"].fill_(1)            # everything flows east
        t.t["dem"].fill_(1.0)
        torch.cuda.synchronize()
        t.fa_local()
        ext = np.zeros(t.P, np.uint64)
        ys, xs = tiling.ring_coords(t.H, t.W)
        ext[(ys == 5) & (xs == 0)] = inflow  # enters at the west border of row 5
        t.fa_finish(ext)
        if expect:
            with pytest.raises(OverflowError):
                t.check_status()
        else:
            t.check_status()
            fac = t.host("fac")
            assert fac[5, 0] == inflow and fac[5, 127] == inflow + 127 and fac[6, 127] == 127


def test_two_phase_state_is_guarded():
    """dt_dev_*_finish_w needs the state its *_local_w left in the context's scratch: a scratch-using call in
    between must make the finish fail loudly instead of producing wrong rasters with rc 0."""
    import ctypes as C
    import torch
    from descriptools_amd import _lib, tiling
    layout = tiling.Layout([128], [128])
    t = tiling.RankTile(layout, 0, device=0, river_threshold=10)
    t.synth_dem(3)
    t.d8()
    t.fa_local()
    out3 = torch.zeros(3, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().dt_dev_unique_extremes_f32(t.ctx.h, t.p("dem"), 16, out3.data_ptr()))  # takes the scratch
    with pytest.raises(RuntimeError, match="without a matching"):
        t.fa_finish(None)
    t.fa_local()
    t.fa_finish(None)          # fine again
    t.fh_local()
    with pytest.raises(RuntimeError, match="without a matching"):
        t.fa_finish(None)      # the HAND state replaced the flow-accumulation state
    t.fh_finish(None)
    t.ctx.sync()


def test_walker_records_are_routed_on_the_device_for_a_thousand_ranks():
    """dt_dev_downslope_walk_route_w's classification and grouping alone, on a 32 x 31 layout (992 ranks; the scatter's
    offsets are a scan over the counts): finished records whose start cell is this rank's are written into the raster
    and dropped, finished ones of other ranks go to the owner of their start cell, walkers standing elsewhere to the
    owner of the cell they stand on; counts and groups against numpy"""
    import ctypes as C
    import torch
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    L = _lib.lib()
    ty, tx, T = 32, 31, 64
    Hg, Wg = ty * T, tx * T
    rng = np.random.default_rng(8)
    n = 60000
    rec = np.zeros((n, 12), np.int32)
    rec[:, 0], rec[:, 1] = rng.integers(T, Hg, n), rng.integers(0, Wg, n)        # start cells (not rank 0's)
    rec[:, 2], rec[:, 3] = rng.integers(2 * T, Hg, n), rng.integers(0, Wg, n)    # standing far from rank 0's memory
    done = rng.random(n) < 0.5
    rec[:, 7] = np.where(done, 2, 0)
    rec[:, 10] = rng.random(n).astype(np.float32).view(np.int32)
    # a few finished records whose start cell is rank 0's own (distinct cells)
    mine = rng.choice(T * T, 300, replace=False)
    rec[:300, 0], rec[:300, 1], rec[:300, 7] = mine // T, mine % T, 2
    done[:300] = True
    ctx = Context()
    dev = torch.device("cuda", 0)
    ld = T + 2                                     # rank 0's window with a one-cell halo
    win = _lib.Window(T, T, ld, 0, 0, Hg, Wg, 1)
    dem = torch.zeros((ld, ld), dtype=torch.float32, device=dev)
    fdr = torch.zeros((ld, ld), dtype=torch.uint8, device=dev)
    out_ext = torch.full((ld, ld), -7.0, dtype=torch.float32, device=dev)
    core = lambda t: t.data_ptr() + (ld + 1) * t.element_size()
    out = out_ext[1:1 + T, 1:1 + T]
    d_rec = torch.as_tensor(rec, device=dev).contiguous()
    send = torch.zeros_like(d_rec)
    rows = torch.as_tensor(np.arange(ty + 1, dtype=np.int32) * T, device=dev)
    cols = torch.as_tensor(np.arange(tx + 1, dtype=np.int32) * T, device=dev)
    counts = torch.zeros(ty * tx + 1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(n + ty * tx, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    _lib.check(L.dt_dev_downslope_walk_route_w(ctx.h, C.byref(win), core(dem), core(fdr), 10.0, 5.0, n,
                                               d_rec.data_ptr(), None, 0, core(out_ext), rows.data_ptr(), ty,
                                               cols.data_ptr(), tx, send.data_ptr(), counts.data_ptr(),
                                               scratch.data_ptr()))
    ctx.sync()
    home = done & (rec[:, 0] < T) & (rec[:, 1] < T)
    dest = np.where(done, (rec[:, 0] // T) * tx + rec[:, 1] // T, (rec[:, 2] // T) * tx + rec[:, 3] // T)
    live = ~home
    want_counts = np.bincount(dest[live], minlength=ty * tx)
    got_counts = counts.cpu().numpy()
    assert np.array_equal(got_counts[:-1], want_counts) and got_counts[-1] == int((~done).sum())
    got_send = send.cpu().numpy()[:int(want_counts.sum())]
    offs = np.concatenate([[0], np.cumsum(want_counts)])
    key = lambda a: a[np.lexsort(a.T[::-1])]
    for d in np.flatnonzero(want_counts)[::37]:  # (every 37th group, row for row)
        seg = got_send[offs[d]:offs[d + 1]]
        assert np.array_equal(key(seg), key(rec[live & (dest == d)])), d
    g = got_send
    gd = np.where((g[:, 7] & 2) != 0, (g[:, 0] // T) * tx + g[:, 1] // T, (g[:, 2] // T) * tx + g[:, 3] // T)
    assert np.array_equal(gd, np.repeat(np.arange(ty * tx), want_counts)), "every row sits in its destination's group"
    o = out.cpu().numpy()
    assert np.array_equal(o[rec[home, 0], rec[home, 1]].view(np.int32), rec[home, 10])
    assert int((o != -7.0).sum()) == len(np.unique(rec[home, 0] * T + rec[home, 1]))
    ctx.close()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("long_walks", [False, True])
@pytest.mark.parametrize("heights,widths", [([256, 256], [384, 384]), ([512], [192, 320, 256])])
def test_downslope_walks_across_rank_borders(heights, widths, long_walks):
    """terrain whose walks run for thousands of moves (a 1 per mille plane, a flat, nodata): on every rank border the
    window kernel marks the cells whose walk leaves the rank's memory; tiling.finish_downslope sends them on as
    walkers from rank to rank (LocalComm: one thread per logical rank) -- the result equals the oracle's walk on the
    whole raster, cell for cell.  long_walks: the ranks run dt_dev_downslope_lift_w (queue + skip tables over core +
    halo; on this plane more walks are long than the queue holds, so the walks a full queue leaves to the window kernel
    are covered too)"""
    import threading
    import torch
    import oracle
    from descriptools_amd import tiling
    layout = tiling.Layout(heights, widths)
    Hg, Wg = layout.Hg, layout.Wg
    yy, xx = np.mgrid[0:Hg, 0:Wg]
    dem = (200.0 - 0.001 * xx - 0.0002 * yy).astype(np.float32)
    dem[150:180, 100:500] = np.float32(150.0)
    rng = np.random.default_rng(4)
    dem[rng.random((Hg, Wg)) < 0.0005] = -100
    _, fdr = oracle.slope_d8(dem, 1.0)
    want = oracle.downslope(dem, fdr, 1.0, 5.0)
    h = tiling.HALO
    pad = np.full((Hg + 2 * h, Wg + 2 * h), -100.0, np.float32)
    pad[h:h + Hg, h:h + Wg] = dem
    tiles = []
    for r in range(layout.size):
        tl = tiling.RankTile(layout, r, device=0, px=1.0, dz=5.0, river_threshold=Hg * Wg // 64,
                             long_walks=long_walks)
        y0, x0 = layout.origin(r)
        tl.set_dem_ext(pad[y0:y0 + tl.He, x0:x0 + tl.We])
        tiles.append(tl)
    tiling.simulate_dev(tiles, layout)
    marked = sum(tl.unresolved_downslope() for tl in tiles)
    assert marked > 1000, "the terrain is meant to send many walks across the borders"
    comms = tiling.LocalComm.create(layout.size)
    done, errors = [None] * layout.size, []

    def work(r):
        try:
            done[r] = tiling.finish_downslope(tiles[r], comms[r])
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append(e)
            comms[r].sh.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(layout.size)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert all(d == marked for d in done)
    for tl in tiles:
        y0, x0 = layout.origin(tl.rank)
        got = tl.host("down")
        ref = want[y0:y0 + tl.H, x0:x0 + tl.W]
        assert tl.unresolved_downslope() == 0
        assert np.array_equal(got, ref, equal_nan=True), (tl.rank, int((got != ref).sum()))
        tl.free()
    torch.cuda.empty_cache()


def test_ranks_on_real_terrain_with_the_long_walk_workspace():
    """the bundled Example with its GIS D8 raster (flats, valley floors: walks of thousands of moves), cropped to
    multiples of 64 and split 2 x 2 through the middle of the data: with and without RankTile(long_walks=True) every
    rank marks the same cells as leaving its memory, and the cells it resolves itself equal the untiled raster's"""
    import torch
    from conftest import load_example
    from descriptools_amd import tiling, downslope
    ex = load_example()
    dem = np.ascontiguousarray(np.asarray(ex[0], np.float32)[:2176, :1472])
    fdr = np.ascontiguousarray(np.asarray(ex[1], np.uint8)[:2176, :1472])
    want = downslope.downsloper(dem, fdr, 12.5, 5.0)
    layout = tiling.Layout([1088, 1088], [704, 768])
    h = tiling.HALO
    dem_p, fdr_p = np.pad(dem, h, constant_values=-100.0), np.pad(fdr, h, constant_values=0)
    left = {}
    for long_walks in (False, True):
        for r in range(layout.size):
            tl = tiling.RankTile(layout, r, device=0, px=12.5, dz=5.0, rasters=("dem", "fdr", "down"),
                                 tune_placement=False, long_walks=long_walks)
            y0, x0 = layout.origin(r)
            tl.set_dem_ext(dem_p[y0:y0 + tl.He, x0:x0 + tl.We])
            with tl.on_stream():
                tl.t["fdr"].copy_(torch.as_tensor(np.ascontiguousarray(fdr_p[y0:y0 + tl.He, x0:x0 + tl.We])))
            tl.ctx.sync()
            tl.downslope()
            un = tl.unresolved_downslope()
            got = tl.host("down")
            own = got != -50
            assert int((~own).sum()) == un
            ref = want[y0:y0 + tl.H, x0:x0 + tl.W]
            assert np.array_equal(got[own].view(np.int32), ref[own].view(np.int32)), (long_walks, r)
            left.setdefault(r, []).append((un, np.flatnonzero(~own)))
            tl.free()
    assert sum(v[0][0] for v in left.values()) > 1000, "the split is meant to cut through long walks"
    for r, (a, b) in left.items():
        assert a[0] == b[0] and np.array_equal(a[1], b[1]), r
    torch.cuda.empty_cache()
