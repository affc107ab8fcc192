"""GPU (-m gpu): BASELINE.json configs[3] -- the 32768 x 32768 DEM (2^30 cells) -- on ONE MI355X: the untiled chain
(~65 GB) and the same DEM as 2 x 2 logical ranks of 16384^2 with the product's rank-level solves on the GPU
(tiling.simulate_dev; ~60 GB more), raster for raster identical; the size-independent properties of the chain; the
20000-move cap biting on real terrain (paths of > 20000 moves exist at this size) and resolved identically by the
tile hierarchy, by the rank hierarchy and by the independent global kernels (dt_set_flow_impl(1)); and the same DEM as
1 x 2 logical ranks of 32768 x 16384 -- the rank-tile shape of configs[4] (65536^2 over 8 GPUs), with that
configuration's int64 accumulation / river index -- through tiling.rank_ops."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config4_32768_tiled_2x2_equals_untiled():
    import torch
    from descriptools_amd import _lib, chain, tiling
    from descriptools_amd.device import Context
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if total < 200 * 2 ** 30:
        pytest.skip("needs an MI355X-sized HBM (288 GB)")
    n, px = 32768, 10.0
    thr = n * n // 512
    L = _lib.lib()
    ctx = Context()
    dem = torch.empty((n, n), dtype=torch.float32, device="cuda")
    _lib.check(L.dt_dev_synth_dem(ctx.h, 1, n, n, 0, 0, n, n, 0, dem.data_ptr()))
    keep = []

    def alloc(shape, dtp):
        t = torch.empty(shape, dtype={np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8,
                                      np.int32: torch.int32}[dtp], device="cuda")
        keep.append(t)
        return t.data_ptr()
    ch = chain.Chain(n, n, ctx=ctx, px=px, river_threshold=thr, alloc=alloc)
    ch.run(dem.data_ptr())
    ctx.sync()
    # the chain hands its blocks to the rasters by measured write-conflict class (placement.py): look them up by pointer
    by_ptr = {x.data_ptr(): x for x in keep}
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8, np.int32: torch.int32}
    t = {name: by_ptr[ch.p(name)].view(tdt[dt]) for name, dt in chain.OUTPUTS}
    fdr, fac, river, idx, hand, fdist = t["fdr"], t["fac"], t["river"], t["idx"], t["hand"], t["fdist"]
    # (1) conservation, river mask, HAND consistency (as test_full_size_16384_properties), in row blocks
    assert int((fac < 0).sum()) == 0
    dy = torch.zeros(256, dtype=torch.int32, device="cuda")
    dx = torch.zeros(256, dtype=torch.int32, device="cuda")
    for c, (a, b) in {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0),
                      128: (-1, 1)}.items():
        dy[c], dx[c] = a, b
    drained, capped, riv_cells = 0, 0, 0
    xx = torch.arange(n, device="cuda", dtype=torch.int32).view(1, -1)
    for y0 in range(0, n, 4096):
        sl = slice(y0, y0 + 4096)
        yy = torch.arange(y0, y0 + 4096, device="cuda", dtype=torch.int32).view(-1, 1)
        f = fdr[sl].long()
        ty, tx = yy + dy[f], xx + dx[f]
        outlet = (ty < 0) | (ty >= n) | (tx < 0) | (tx >= n) | (f == 0)
        drained += int((fac[sl][outlet].long() + 1).sum())
        assert torch.equal(river[sl], (fac[sl] > thr).to(torch.int8))
        rv = river[sl] == 1
        riv_cells += int(rv.sum())
        lin = (yy * n + xx)
        assert torch.equal(idx[sl][rv], lin[rv]) and (not bool(rv.any()) or float(fdist[sl][rv].abs().max()) == 0.0)
        ok = idx[sl] >= 0
        assert bool((river.view(-1)[idx[sl][ok].long()] == 1).all())
        assert bool((hand[sl][ok] >= 0).all()) and bool((hand[sl][~ok] == -100).all())
        capped += int((~ok).sum())
        del f, ty, tx, outlet, rv, lin, ok
    assert drained == n * n and riv_cells > 0
    # (2) the move cap bites on this terrain: cells whose path to the drainage network exceeds 20000 moves are -100
    assert capped > 0, "no path of > 20000 moves at 32768^2: the cap is not exercised"
    print("cells beyond the 20000-move cap: %d of %d" % (capped, n * n))
    # (3) the independent global kernels (raster-wide countdown / pointer doubling) agree cell for cell
    tmp = {k: torch.empty((n, n), dtype=v.dtype, device="cuda") for k, v in (("fac", fac), ("fdist", fdist),
                                                                             ("idx", idx), ("hand", hand))}
    _lib.check(L.dt_set_flow_impl(1))
    try:
        _lib.check(L.dt_dev_flowacc(ctx.h, fdr.data_ptr(), dem.data_ptr(), n, n, tmp["fac"].data_ptr()))
        _lib.check(L.dt_dev_flowhand(ctx.h, dem.data_ptr(), fdr.data_ptr(), river.data_ptr(), None, n, n, px,
                                     tmp["fdist"].data_ptr(), tmp["idx"].data_ptr(), tmp["hand"].data_ptr(), None))
    finally:
        _lib.check(L.dt_set_flow_impl(2))
    ctx.sync()
    for k, v in tmp.items():
        assert torch.equal(v, t[k]), "global kernels vs tile hierarchy: " + k
    del tmp
    # (4) 2 x 2 logical ranks of 16384^2 == untiled, every raster
    half = n // 2
    layout = tiling.Layout([half, half], [half, half])
    tiles = []
    for r in range(4):
        tl = tiling.RankTile(layout, r, device=0, px=px, river_threshold=thr)
        tl.synth_dem(1)
        tiles.append(tl)
    tiling.simulate_dev(tiles, layout)
    for tl in tiles:
        tl.check_status()
        assert tl.unresolved_downslope() == 0
        y0, x0 = layout.origin(tl.rank)
        for name in ("fdr", "fac", "river", "fdist", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"):
            a, b = tl.core(name), t[name][y0:y0 + tl.H, x0:x0 + tl.W]
            assert torch.equal(a, b), (tl.rank, name, int((a != b).sum()))
        gi = tl.core("idx")
        li = t["idx"][y0:y0 + tl.H, x0:x0 + tl.W].long()
        gy, gx = li // n, li % n
        want = torch.where(li >= 0, gy * n + gx, li)
        assert gi.dtype == torch.int32  # 2^30 cells: the 32-bit global index
        assert torch.equal(gi.long(), want), (tl.rank, "idx")
        tl.free()
    del tiles
    gc.collect()
    torch.cuda.empty_cache()
    # (5) BASELINE.json configs[4]'s RANK-TILE SHAPE -- 32768 x 16384 with its 64-cell halo, int64 accumulation and int64
    # global river index as a rank of the 65536^2 raster has them -- through tiling.rank_ops (the schedule bench.py
    # times at N = 8): 1 x 2 logical ranks of that shape are this very DEM
    layout = tiling.Layout([n], [half, half])
    assert layout.shape(0) == (32768, 16384)
    tiles = []
    for r in range(2):
        tl = tiling.RankTile(layout, r, device=0, px=px, river_threshold=thr, acc64=True, idx64=True,
                             tune_placement=False)
        tl.synth_dem(1)
        tiles.append(tl)
    tiling.run_ranks_local(tiles, layout)
    for tl in tiles:
        tl.check_status()
        assert tl.unresolved_downslope() == 0
        y0, x0 = layout.origin(tl.rank)
        for name in ("fdr", "river", "fdist", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"):
            a, b = tl.core(name), t[name][y0:y0 + tl.H, x0:x0 + tl.W]
            assert torch.equal(a, b), ("32768x16384 rank", tl.rank, name, int((a != b).sum()))
        assert tl.core("fac").dtype == torch.int64 and tl.core("idx").dtype == torch.int64
        for name in ("fac", "idx"):
            for yb in range(0, tl.H, 8192):  # (in row blocks: the int64 comparison of 2^29 cells at once needs no 8 GB)
                a, b = tl.core(name)[yb:yb + 8192], t[name][y0 + yb:y0 + yb + 8192, x0:x0 + tl.W]
                assert torch.equal(a, b.long()), ("32768x16384 rank", tl.rank, name)
        tl.free()
    del tiles, t, keep, dem
    gc.collect()
    torch.cuda.empty_cache()
