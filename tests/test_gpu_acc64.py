"""GPU (-m gpu): 64-bit flow accumulation.  The reference's `fac` is int64 end to end (Example/example.py:39,
topoindexes.py:252-261, gfi.py:141-143, :432-440); on the device a raster below 2^31 cells keeps it as int32, a raster of
>= 2^31 cells split over ranks as int64 (the `_a64` entry points, RankTile(acc64=True)).  Here: values beyond 2^31
really flowing through the tile hierarchy, the rank-level solves and every consumer, at small raster sizes (huge
inflows injected where another rank would deliver them) and on a real raster of more than 2^31 cells."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gfi_expected(hand, a_river, fac, n=0.4, b=0.1, px=10.0):
    """gfi.py:268-294 and :404-440 in float64 (the arithmetic Numba types the kernels with)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        g = np.log(b * np.power(a_river.astype(np.float64) * px * px, n) / (hand.astype(np.float64) + 0.01))
        f = np.where(fac == 0, 1, fac).astype(np.float64)
        l = np.log(b * np.power(f * px * px, n) / (hand.astype(np.float64) + 0.01))
    return g, l


def test_inflow_beyond_2_31_through_tiles():
    """One rank, everything flows east; inflows of 2^31 - 100, 2^33 + 7 and 2^40 + 12345 cells enter at the west border
    (where the rank-level solve would deliver another rank's drainage) and a second, merging one of 2^31 + 5: the tile
    pass carries them in two limbs through its 32-bit LDS raster, across the tile border at x = 64 and through a
    confluence.  With int32 rasters the same inflow raises DT_STATUS_ACC_OVERFLOW (test_gpu_tiling); here the values
    come out exactly and the status stays clear."""
    import torch
    from descriptools_amd import tiling
    layout = tiling.Layout([128], [128, 128])
    B = 2 ** 31 + 5
    for A in (2 ** 31 - 100, 2 ** 33 + 7, 2 ** 40 + 12345):
        t = tiling.RankTile(layout, 1, device=0, river_threshold=2 ** 32, acc64=True)
        t.t["fdr"].fill_(1)            # everything flows east ...
        h = t.halo
        t.t["fdr"][h + 6, h + 64] = 128  # ... except (6, 64): north-east, into row 5
        t.t["dem"].fill_(1.0)
        torch.cuda.synchronize()
        t.fa_local()
        ext = np.zeros(t.P, np.uint64)
        ys, xs = tiling.ring_coords(t.H, t.W)
        ext[(ys == 5) & (xs == 0)] = A
        ext[(ys == 6) & (xs == 0)] = B
        ext[(ys == 9) & (xs == 0)] = 3  # a small one beside them
        t.fa_finish(ext)
        t.check_status()
        fac = t.host("fac")
        assert fac.dtype == np.int64
        x = np.arange(128, dtype=np.int64)
        want = np.tile(x, (128, 1))
        want[5] += A
        want[5, 65:] += B + 65          # row 6's first 65 cells and its inflow join at (5, 65)
        want[6, :65] += B
        want[6, 65:] = x[65:] - 65      # cut off from its upstream cells
        want[9] += 3
        assert np.array_equal(fac, want), int((fac != want).sum())
        river = t.host("river")
        assert np.array_equal(river, (want > 2 ** 32).astype(np.int8))
        t.free()


def test_int64_accumulation_across_ranks_small():
    """1 x 2 ranks, everything flows east; rank 0's summary row is raised by 2^33 cells per exit before the
    all-gather (as if it had that much more upstream): rank 1's accumulation, river mask, the river accumulation that
    HAND carries back into rank 0 as payload (a REMOTE river cell beyond 2^31) and GFI / ln(hl/H) from it, through the
    product's device-side rank-level solves."""
    import torch
    from descriptools_amd import tiling
    layout = tiling.Layout([128], [128, 128])
    px, BIG = 10.0, 2 ** 33
    tiles = []
    for r in range(2):
        t = tiling.RankTile(layout, r, device=0, px=px, river_threshold=2 ** 32, acc64=True)
        t.t["fdr"].fill_(1)
        t.t["dem"].fill_(7.0)
        tiles.append(t)
    torch.cuda.synchronize()
    for t in tiles:
        t.fa_local(sync=False)
        t.fill_ring_codes()
    for t in tiles:
        t.ctx.sync()
    A0 = tiles[0]._fa_v["A"]
    A0[A0 > 0] += BIG
    torch.cuda.synchronize()
    rows = torch.cat([t.fa_row for t in tiles])
    torch.cuda.synchronize()
    for t in tiles:
        t.fa_solve_finish(rows)
    for t in tiles:
        t.fh_local(sync=False)
    for t in tiles:
        t.ctx.sync()
    rows = torch.cat([t.fh_row for t in tiles])
    torch.cuda.synchronize()
    for t in tiles:
        t.fh_solve_finish(rows, fuse_gfi=True)
        t.check_status()
    x = np.arange(128, dtype=np.int64)
    f0, f1 = tiles[0].host("fac"), tiles[1].host("fac")
    assert np.array_equal(f0, np.tile(x, (128, 1)))
    assert np.array_equal(f1, np.tile(BIG + 128 + x, (128, 1)))
    assert tiles[0].host("river").sum() == 0 and tiles[1].host("river").all()
    # rank 0 drains into the river cell (y, 128) of rank 1: index, distance, payload
    ar0 = tiles[0].host("a_river")
    assert ar0.dtype == np.int64 and np.array_equal(ar0, np.full((128, 128), BIG + 128))
    assert np.array_equal(tiles[0].host("idx"), np.tile(np.arange(128)[:, None] * 256 + 128, (1, 128)))
    assert np.array_equal(tiles[0].host("fdist"), np.tile(((128 - x) * px).astype(np.float32), (128, 1)))
    assert (tiles[0].host("hand") == 0).all() and (tiles[1].host("hand") == 0).all()
    for t, fac in ((tiles[0], f0), (tiles[1], f1)):
        g, l = _gfi_expected(t.host("hand"), t.host("a_river"), fac)
        assert np.allclose(t.host("gfi"), g, rtol=1e-6, atol=0) and np.allclose(t.host("lnhlh"), l, rtol=1e-6, atol=0)
    # fused slope + TI + MTI on the int64 raster: TI = ln(max(fac, 1) px^2 / tan(0 + 0.01)) on this flat DEM
    for t, fac in ((tiles[0], f0), (tiles[1], f1)):
        t.slope_twi()
        ti = t.host("ti")
        want = np.log(np.maximum(fac, 1).astype(np.float64) * px * px / np.tan(0.01))
        assert np.allclose(ti, want, rtol=1e-5, atol=0), float(np.abs(ti - want).max())
        mti = t.host("mti")
        want = np.log(np.power(np.maximum(fac, 1).astype(np.float64) * px * px, 0.1) / np.tan(0.01))
        assert np.allclose(mti, want, rtol=1e-5, atol=1e-6)
        t.free()


def _comb_tile(t, torch):
    """rows flow west into column 0 of the GLOBAL raster, column 0 flows south: one basin holding every cell"""
    gx = torch.arange(t.gx0 - t.halo, t.gx0 - t.halo + t.We, device=t.dev)
    gy = torch.arange(t.gy0 - t.halo, t.gy0 - t.halo + t.He, device=t.dev)
    code = torch.where(gx == 0, 4, 16).to(torch.uint8).view(1, -1).expand(t.He, t.We).clone()
    inside = ((gy >= 0) & (gy < t.layout.Hg)).view(-1, 1) & ((gx >= 0) & (gx < t.layout.Wg)).view(1, -1)
    code[~inside] = 0
    t.t["fdr"].copy_(code)


def test_comb_basin_of_more_than_2_31_cells():
    """A real raster of 32768 x 65600 = 2.15e9 cells (> 2^31) as 2 x 4 logical ranks on one GPU, all of it ONE basin
    (rows flow west, column 0 flows south): accumulation values up to Hg * Wg - 1 > 2^31 against the closed form,
    conservation, the river mask at a 64-bit threshold, the global river index as int64, HAND's river payload and GFI
    beyond 32 bits, the 20000-move cap on the walks down column 0; the overflow status never fires."""
    import torch
    from descriptools_amd import tiling
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if total < 200 * 2 ** 30:
        pytest.skip("needs an MI355X-sized HBM (288 GB)")
    layout = tiling.Layout([16384, 16384], [16384, 16384, 16384, 16448])
    Hg, Wg = layout.Hg, layout.Wg
    assert Hg * Wg > 2 ** 31
    px = 10.0
    thr = 2 ** 31 + 12345  # river: column 0 below row thr / Wg
    names = ("fdr", "fac", "river", "dem", "fdist", "idx", "hand", "a_river", "gfi", "lnhlh")
    tiles = []
    for r in range(layout.size):
        t = tiling.RankTile(layout, r, device=0, px=px, river_threshold=thr, rasters=names)
        assert t.acc64 and t.idx_dtype == torch.int64
        with t.on_stream():
            _comb_tile(t, torch)
            t.t["dem"].fill_(3.0)
        tiles.append(t)
    torch.cuda.synchronize()

    def gather(rows):
        for t in tiles:
            t.ctx.sync()
        out = torch.cat(rows)
        torch.cuda.synchronize()
        return out
    for t in tiles:
        t.fa_local(sync=False)
        t.fill_ring_codes()
    rows = gather([t.fa_row for t in tiles])
    for t in tiles:
        t.fa_solve_finish(rows)
    for t in tiles:
        t.fh_local(sync=False)
    rows = gather([t.fh_row for t in tiles])
    for t in tiles:
        t.fh_solve_finish(rows, fuse_gfi=True)
    yr0 = thr // Wg  # first river row: (y + 1) * Wg - 1 > thr  <=>  y >= yr0 (thr is not a multiple of Wg)
    assert (yr0 + 1) * Wg - 1 > thr >= yr0 * Wg - 1
    top = 0
    for t in tiles:
        t.check_status()  # int64 rasters: the overflow status is never raised
        y = torch.arange(t.gy0, t.gy0 + t.H, device=t.dev, dtype=torch.int64).view(-1, 1)
        x = torch.arange(t.gx0, t.gx0 + t.W, device=t.dev, dtype=torch.int64).view(1, -1)
        fac = t.core("fac")
        want = torch.where(x == 0, (y + 1) * Wg - 1, (Wg - 1 - x).expand(t.H, t.W))
        assert torch.equal(fac, want), (t.rank, int((fac != want).sum()))
        top = max(top, int(fac.max()))
        assert torch.equal(t.core("river"), ((x == 0) & (y >= yr0)).to(torch.int8).expand(t.H, t.W).contiguous())
        # HAND: a cell (y, x) walks x moves west, then max(yr0 - y, 0) moves south; beyond 20000 moves -> -100
        moves = x + torch.clamp(yr0 - y, min=0)
        ok = moves <= 20000
        idx = t.core("idx")
        yr = torch.maximum(y, torch.tensor(yr0, device=t.dev)).expand(t.H, t.W)
        want_idx = torch.where(ok, yr * Wg, torch.full_like(yr, -100))
        if not torch.equal(idx, want_idx):
            bad = torch.nonzero(idx != want_idx)
            b0 = bad[0]
            raise AssertionError("rank %d idx: %d cells differ, first at local %s: got %d want %d (moves %d); last at %s"
                                 % (t.rank, len(bad), b0.tolist(), int(idx[b0[0], b0[1]]), int(want_idx[b0[0], b0[1]]),
                                    int(moves.expand(t.H, t.W)[b0[0], b0[1]]), bad[-1].tolist()))
        del want_idx
        fd = t.core("fdist")
        assert torch.equal(fd, torch.where(ok, (moves.double() * px).float(), torch.full_like(fd, -100.0))), (t.rank, "fdist")
        ar = t.core("a_river")
        want_ar = torch.where(ok, (yr + 1) * Wg - 1, torch.full_like(yr, -100))
        assert torch.equal(ar, want_ar), (t.rank, "a_river")
        hand = t.core("hand")
        assert torch.equal(hand, torch.where(ok, torch.zeros_like(hand), torch.full_like(hand, -100.0)))
        # GFI / ln(hl/H) (gfi.py:268-294, :404-440) from 64-bit areas, in float64 on a strided sample
        sl = (slice(None, None, 97), slice(None, None, 89))
        g = torch.log(0.1 * torch.pow(want_ar[sl].double() * px * px, 0.4) / 0.01)
        f = torch.clamp(want[sl], min=1).double()
        l = torch.log(0.1 * torch.pow(f * px * px, 0.4) / 0.01)
        oks = ok.expand(t.H, t.W)[sl]
        gg, ll = t.core("gfi")[sl].double(), t.core("lnhlh")[sl].double()
        assert bool(((gg - g).abs() <= 1e-5 * g.abs())[oks].all()) and bool((gg[~oks] == -100).all())
        assert bool(((ll - l).abs() <= 1e-5 * l.abs())[oks].all()) and bool((ll[~oks] == -100).all())
        del y, x, want, moves, ok, yr, want_ar, g, f, l
    assert top == Hg * Wg - 1 and top > 2 ** 31
    # conservation: the one outlet (Hg - 1, 0) drains every cell
    last = tiles[(layout.ty - 1) * layout.tx]
    assert int(last.core("fac")[last.H - 1, 0]) + 1 == Hg * Wg
    for t in tiles:
        t.free()
