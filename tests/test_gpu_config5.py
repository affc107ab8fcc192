"""GPU (-m gpu): BASELINE.json configs[4] -- the raster of >= 2^31 cells tiled over 8 ranks, "full chain +
evaluation.py flood-map classifier" -- exercised on ONE MI355X as 2 x 4 logical ranks of 16384^2 (32768 x 65536 = 2^31
cells, ~155 GB of the 288 GB): int64 flow accumulation (what a raster beyond 2^31 cells gets by default; at exactly 2^31
cells int32 would still hold every value, so it is requested here), the rank-level solves on the GPU,
the global river index in both widths, and the classifier's rank reductions.  No computation of the whole raster
exists to compare with at this size (the global kernels index in 32 bits), so the checks are size-independent:

  * conservation: every cell drains through exactly one outlet of the GLOBAL raster;
  * tiling invariance: a second decomposition of the same DEM (4 x 2 ranks of 8192 x 32768) gives the same rasters
    (one checksum per 2048 x 2048 block per raster, so a mismatch is localised);
  * an independent walk of the reference's kernel (flowhand.py:566-846: follow the D8 codes until river == 1; leaving
    the raster / arriving on code 0 / more than 20000 moves -> -100) from sampled cells, on the host, against the
    river index and flow distance of the tiles -- including cells beyond the 20000-move cap;
  * the classifier: calibrated threshold and confusion counts identical under both decompositions, and the counts
    recomputed with plain tensor operations.
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLK = 2048
NAMES = ("fdr", "fac", "river", "fdist", "idx", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down")


def _block_sums(t, torch, name):
    """int64 checksum of every BLK x BLK block of the tile's core raster (bit patterns for floats, values for ints)"""
    c = t.core(name)
    if c.dtype in (torch.float32,):
        c = c.view(torch.int32)
    c = c.reshape(t.H // BLK, BLK, t.W // BLK, BLK)
    return c.sum(dim=(1, 3), dtype=torch.int64).cpu().numpy()


def _flood(t, torch):
    """synthetic benchmark flood map of a tile: low HAND, with a deterministic drop-out pattern of global coordinates"""
    gy = torch.arange(t.gy0, t.gy0 + t.H, device=t.dev, dtype=torch.int64).view(-1, 1)
    gx = torch.arange(t.gx0, t.gx0 + t.W, device=t.dev, dtype=torch.int64).view(1, -1)
    h = t.core("hand")
    keep = ((gy * 7 + gx * 13) % 10) != 0
    return ((h >= 0) & (h < 1.5) & keep).to(torch.int8).contiguous()


LOCAL = ("slope", "ti", "mti", "gfi", "lnhlh", "down")   # rasters no other stage or rank reads


def _run_layout(layout, seed, thr, idx64, sample, lean=False):
    """one decomposition: run the step on every logical rank, collect block checksums, properties, the classifier and
    the sampled cells; frees everything before returning.  lean: the six rasters nothing downstream reads (LOCAL) exist
    ONCE and serve every logical rank in turn -- each rank's are reduced to their block checksums before the next rank
    overwrites them -- so that eight ranks of 2^29 cells fit one device."""
    import gc
    import torch
    from descriptools_amd import tiling
    gc.collect()
    torch.cuda.empty_cache()  # earlier tests' cached blocks are invisible to the library's own allocations
    Hg, Wg = layout.Hg, layout.Wg
    tiles = []
    out = {"sums": {n: np.zeros((Hg // BLK, Wg // BLK), np.int64) for n in NAMES}}
    if lean:
        keep = ("dem", "fdr", "fac", "river", "fdist", "idx", "hand")
        shared = None
        for r in range(layout.size):
            t = tiling.RankTile(layout, r, device=0, px=10.0, river_threshold=thr, idx64=idx64, acc64=True, rasters=keep,
                                tune_placement=False)
            if shared is None:  # (every rank tile of a uniform layout has the same extended shape)
                shared = {n: torch.zeros((t.He, t.We), dtype=torch.float32, device="cuda") for n in LOCAL}
            assert all(tuple(shared[n].shape) == (t.He, t.We) for n in LOCAL)
            t.t.update(shared)
            t.synth_dem(seed)
            tiles.append(t)
        # tiling.simulate_dev, with the last stage rank by rank
        for t in tiles:
            t.d8()
            t.fa_local(sync=False)
            t.fill_ring_codes()
        for t in tiles:
            t.ctx.sync()
        rows = torch.cat([t.fa_row for t in tiles])
        torch.cuda.synchronize()
        for t in tiles:
            t.fa_solve_finish_fh_local(rows)
        for t in tiles:
            t.ctx.sync()
        rows = torch.cat([t.fh_row for t in tiles])
        torch.cuda.synchronize()
        for t in tiles:
            t.fh_solve_finish(rows, fuse_gfi=True, want_a_river=False)
            t.slope_twi()
            t.downslope()
            t.ctx.sync()
            y0, x0 = layout.origin(t.rank)
            for n in LOCAL:
                out["sums"][n][y0 // BLK:(y0 + t.H) // BLK, x0 // BLK:(x0 + t.W) // BLK] = _block_sums(t, torch, n)
    else:
        for r in range(layout.size):
            t = tiling.RankTile(layout, r, device=0, px=10.0, river_threshold=thr, idx64=idx64, acc64=True)
            assert t.acc64 and t.t["fac"].dtype == torch.int64 and t.t["a_river"].dtype == torch.int64
            t.synth_dem(seed)
            tiles.append(t)
        tiling.simulate_dev(tiles, layout)
    dy = torch.zeros(256, dtype=torch.int64, device="cuda")
    dx = torch.zeros(256, dtype=torch.int64, device="cuda")
    for c, (a, b) in {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0),
                      128: (-1, 1)}.items():
        dy[c], dx[c] = a, b
    drained = capped = rivers = 0
    fdr_g = np.zeros((Hg, Wg), np.uint8) if sample is not None else None
    riv_g = np.zeros((Hg, Wg), np.int8) if sample is not None else None
    picked = {}
    for t in tiles:
        t.check_status()                      # int64 rasters: never an overflow
        assert t.unresolved_downslope() == 0
        y0, x0 = layout.origin(t.rank)
        for n in NAMES:
            if lean and n in LOCAL:
                continue  # (reduced right after the rank computed them)
            out["sums"][n][y0 // BLK:(y0 + t.H) // BLK, x0 // BLK:(x0 + t.W) // BLK] = _block_sums(t, torch, n)
        fdr, fac, river, idx, fdist = (t.core(n) for n in ("fdr", "fac", "river", "idx", "fdist"))
        assert int((fac < 0).sum()) == 0
        gy = torch.arange(y0, y0 + t.H, device="cuda", dtype=torch.int64).view(-1, 1)
        gx = torch.arange(x0, x0 + t.W, device="cuda", dtype=torch.int64).view(1, -1)
        f = fdr.long()
        ty, tx = gy + dy[f], gx + dx[f]
        outlet = (ty < 0) | (ty >= Hg) | (tx < 0) | (tx >= Wg) | (f == 0)
        drained += int((fac[outlet] + 1).sum())
        del f, ty, tx, outlet
        assert torch.equal(river, (fac > thr).to(torch.int8))
        rv = river == 1
        rivers += int(rv.sum())
        lin = gy * Wg + gx
        assert torch.equal(idx.long()[rv], lin[rv]) and (not bool(rv.any()) or float(fdist[rv].abs().max()) == 0.0)
        assert idx.dtype == (torch.int64 if idx64 else torch.int32)
        capped += int((idx < 0).sum())
        del rv, lin
        if sample is not None:
            fdr_g[y0:y0 + t.H, x0:x0 + t.W] = fdr.cpu().numpy()
            riv_g[y0:y0 + t.H, x0:x0 + t.W] = river.cpu().numpy()
            sy, sx = sample
            m = (sy >= y0) & (sy < y0 + t.H) & (sx >= x0) & (sx < x0 + t.W)
            ly = torch.as_tensor(sy[m] - y0, device="cuda")
            lx = torch.as_tensor(sx[m] - x0, device="cuda")
            picked[t.rank] = (np.nonzero(m)[0], idx[ly, lx].long().cpu().numpy(), fdist[ly, lx].cpu().numpy())
    assert drained == Hg * Wg, (drained, Hg * Wg)
    assert rivers > 0 and capped > 0
    out["capped"] = capped
    # the classifier on the tiled HAND, every logical rank in its own thread (the product's evaluate_rank; the
    # all-gathers are tiling.LocalComm's)
    floods = [_flood(t, torch) for t in tiles]
    torch.cuda.synchronize()
    comms = tiling.LocalComm.create(layout.size)
    results, errors = [None] * layout.size, []

    def work(r):
        try:
            results[r] = tiling.evaluate_rank(tiles[r], floods[r], comms[r])
        except BaseException as e:  # noqa: BLE001 - reported below
            errors.append(e)
            comms[r].sh.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(layout.size)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for res in results[1:]:
        assert res["threshold"] == results[0]["threshold"] and np.array_equal(res["counts"], results[0]["counts"])
    out["eval"] = results[0]
    # the counts at the calibrated threshold recomputed with plain tensor operations (evaluation.py:5-9, 90-171)
    mn, mx = np.float32(results[0]["mn"]), np.float32(results[0]["mx"])
    th32 = torch.tensor(np.float32(results[0]["threshold"]), device="cuda")
    h00 = float(tiles[0].core("hand")[0, 0])
    first = float("nan") if h00 == -100.0 else float((np.float32(h00) - mn) / (mx - mn))
    want = np.zeros(4, np.int64)
    for t, fl in zip(tiles, floods):
        h = t.core("hand")
        desc = (h - float(mn)) / float(mx - mn)
        binary = (desc <= th32) & (h != -100.0)
        if first == first:
            binary &= desc != first
        k = binary.to(torch.int64) + torch.where(fl == 1, 2, 0)
        want += torch.bincount(k.reshape(-1), minlength=4).cpu().numpy()
        del h, desc, binary, k
    assert np.array_equal(want, results[0]["counts"]), (want, results[0]["counts"])
    out["picked"] = picked
    out["fdr_g"], out["riv_g"] = fdr_g, riv_g
    for t in tiles:
        t.free()
    del tiles, floods
    torch.cuda.empty_cache()
    return out


def _walk(fdr, river, sy, sx, px):
    """flowhand.py:566-846 from the sampled cells, vectorised over the sample: (global river index or -100,
    float32 distance or -100)"""
    Hg, Wg = fdr.shape
    DY = np.zeros(256, np.int64)
    DX = np.zeros(256, np.int64)
    for c, (a, b) in {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0),
                      128: (-1, 1)}.items():
        DY[c], DX[c] = a, b
    n = len(sy)
    y, x = sy.copy(), sx.copy()
    nc, nd = np.zeros(n, np.int64), np.zeros(n, np.int64)
    res = np.full(n, -100, np.int64)
    alive = fdr[y, x] != 0            # flowhand.py:601
    for k in range(20001):
        if not alive.any():
            break
        a = np.nonzero(alive)[0]
        on = river[y[a], x[a]] == 1    # the while condition (:622): k moves made, k <= 20000
        done = a[on]
        res[done] = y[done] * Wg + x[done]
        alive[done] = False
        a = a[~on]
        if k == 20000:
            alive[a] = False           # would need a 20001st move (:834-837)
            break
        code = fdr[y[a], x[a]]
        ny, nx = y[a] + DY[code], x[a] + DX[code]
        inside = (ny >= 0) & (ny < Hg) & (nx >= 0) & (nx < Wg)
        alive[a[~inside]] = False      # leaves the raster (:623-628)
        a, ny, nx, code = a[inside], ny[inside], nx[inside], code[inside]
        diag = (DY[code] != 0) & (DX[code] != 0)
        nd[a] += diag
        nc[a] += ~diag
        y[a], x[a] = ny, nx
        dead = fdr[ny, nx] == 0        # arrival on code 0 (:826-828)
        alive[a[dead]] = False
    ok = res != -100
    dist = np.where(ok, (px * nc + (px * np.sqrt(2.0)) * nd), -100.0).astype(np.float32)
    return res, dist


def test_config5_2x4_ranks_of_16384():
    import torch
    from descriptools_amd import tiling
    free, total = torch.cuda.mem_get_info()
    if total < 200 * 2 ** 30:
        pytest.skip("needs an MI355X-sized HBM (288 GB)")
    S, seed = 16384, 1
    la = tiling.Layout([S, S], [S, S, S, S])                  # configs[4]'s 2 x 4 rank grid
    lb = tiling.Layout([S // 2] * 4, [2 * S, 2 * S])          # the same raster as 4 x 2 ranks of 8192 x 32768
    Hg, Wg = la.Hg, la.Wg
    assert (Hg, Wg) == (lb.Hg, lb.Wg) and Hg * Wg == 2 ** 31
    thr = (Hg * Wg) // 512
    rng = np.random.default_rng(5)
    sy, sx = rng.integers(0, Hg, 6000), rng.integers(0, Wg, 6000)
    a = _run_layout(la, seed, thr, idx64=True, sample=(sy, sx))
    print("2 x 4 ranks: %d cells beyond the 20000-move cap; threshold %r fit %r" %
          (a["capped"], a["eval"]["threshold"], a["eval"]["fit"]))
    # independent walk of the sampled cells on the host
    want_idx, want_fd = _walk(a["fdr_g"], a["riv_g"], sy, sx, 10.0)
    got_idx, got_fd = np.zeros_like(want_idx), np.zeros_like(want_fd)
    for pos, gi, gf in a["picked"].values():
        got_idx[pos], got_fd[pos] = gi, gf
    assert np.array_equal(got_idx, want_idx), int((got_idx != want_idx).sum())
    assert np.array_equal(got_fd, want_fd), int((got_fd != want_fd).sum())
    assert (want_idx == -100).sum() > 50 and (want_idx >= 0).sum() > 1000, "both sides of the cap are sampled"
    del a["fdr_g"], a["riv_g"]
    b = _run_layout(lb, seed, thr, idx64=False, sample=None)   # 2^31 cells: the 32-bit global index still fits
    for n in NAMES:
        bad = np.argwhere(a["sums"][n] != b["sums"][n])
        assert len(bad) == 0, "%s differs between the decompositions in %d blocks, first %s" % (n, len(bad), bad[0])
    assert a["capped"] == b["capped"]
    assert a["eval"]["threshold"] == b["eval"]["threshold"] and np.array_equal(a["eval"]["counts"], b["eval"]["counts"])
    assert a["eval"]["mn"] == b["eval"]["mn"] and a["eval"]["mx"] == b["eval"]["mx"]


def test_config5_at_its_named_size_65536x65536_as_8_logical_ranks():
    """BASELINE.json configs[4] at the size it names: 65536 x 65536 = 2^32 cells as 2 x 4 logical ranks of
    32768 x 16384 on ONE MI355X (the six rasters nothing downstream reads exist once and serve the ranks in turn:
    ~200 GB of the 288).  What only this size has: a global river index beyond 2^32 (the int64 raster's upper half is
    live), accumulations of up to 2^32 - 1 cells, the rank-tile shape of the 8-GPU run in all eight positions of the
    grid.  Checks as at 2^31 cells: conservation over the global raster, river cells indexing themselves with their
    64-bit global index, an independent host walk of the reference's kernel from sampled cells on both sides of the
    20000-move cap, the classifier's rank reductions against plain tensor operations, and invariance of every raster
    under a second decomposition (4 x 2 ranks of 16384 x 32768)."""
    import gc
    import torch
    from descriptools_amd import device, tiling
    gc.collect()
    device.trim()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30 or free < 230 * 2 ** 30:
        pytest.skip("needs ~200 GB of free HBM (an MI355X to itself)")
    la = tiling.Layout([32768, 32768], [16384] * 4)            # the bench's N = 8 layout
    lb = tiling.Layout([16384] * 4, [32768, 32768])
    Hg, Wg = la.Hg, la.Wg
    assert (Hg, Wg) == (65536, 65536) == (lb.Hg, lb.Wg) and Hg * Wg == 2 ** 32
    thr = (Hg * Wg) // 512
    rng = np.random.default_rng(6)
    sy, sx = rng.integers(0, Hg, 6000), rng.integers(0, Wg, 6000)
    a = _run_layout(la, 1, thr, idx64=True, sample=(sy, sx), lean=True)
    print("65536^2 as 2 x 4 ranks: %d cells beyond the 20000-move cap; threshold %r fit %r" %
          (a["capped"], a["eval"]["threshold"], a["eval"]["fit"]))
    want_idx, want_fd = _walk(a["fdr_g"], a["riv_g"], sy, sx, 10.0)
    got_idx, got_fd = np.zeros_like(want_idx), np.zeros_like(want_fd)
    for pos, gi, gf in a["picked"].values():
        got_idx[pos], got_fd[pos] = gi, gf
    assert np.array_equal(got_idx, want_idx), int((got_idx != want_idx).sum())
    assert np.array_equal(got_fd, want_fd), int((got_fd != want_fd).sum())
    assert (want_idx >= 2 ** 32 // 2).sum() > 500 and (want_idx >= 2 ** 31).sum() > 1000, "river indices beyond 2^31 sampled"
    assert int(want_idx.max()) > 2 ** 31 and (want_idx == -100).sum() > 50
    del a["fdr_g"], a["riv_g"]
    b = _run_layout(lb, 1, thr, idx64=True, sample=None, lean=True)
    for n in NAMES:
        bad = np.argwhere(a["sums"][n] != b["sums"][n])
        assert len(bad) == 0, "%s differs between the decompositions in %d blocks, first %s" % (n, len(bad), bad[0])
    assert a["capped"] == b["capped"]
    assert a["eval"]["threshold"] == b["eval"]["threshold"] and np.array_equal(a["eval"]["counts"], b["eval"]["counts"])
