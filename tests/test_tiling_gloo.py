"""CPU (not gpu): the N > 1 path's communication and rank-level solves, world_size 2 on gloo.

Each process plays one rank: it builds its ring summaries with a small pure-numpy model of what the
per-rank GPU phase emits (path following inside its tile), all-gathers them with the SAME
all_gather_summaries the RCCL path uses, runs the SAME solve_flowacc / solve_flowhand, finishes in numpy
and compares with the oracle's untiled result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_DY = {1: 0, 2: 1, 4: 1, 8: 1, 16: 0, 32: -1, 64: -1, 128: -1}
_DX = {1: 1, 2: 1, 4: 0, 8: -1, 16: -1, 32: -1, 64: 0, 128: 1}


def _rank_model(layout, r, fdr_g, river_g, dem_g, acc_g):
    """numpy model of one rank's phase-1 outputs on its core tile (rows of the ring)."""
    from descriptools_amd import tiling
    y0, x0 = layout.origin(r)
    H, W = layout.shape(r)
    ys, xs = tiling.ring_coords(H, W)

    def step(y, x):  # -> (ny, nx) in core coords or None / 'out'
        c = int(fdr_g[y0 + y, x0 + x])
        if c not in _DY:
            return None
        ny, nx = y + _DY[c], x + _DX[c]
        gy, gx = y0 + ny, x0 + nx
        if not (0 <= gy < layout.Hg and 0 <= gx < layout.Wg):
            return None
        if not (0 <= ny < H and 0 <= nx < W):
            return "out"
        return ny, nx

    # local accumulation (paths leaving the core are sinks)
    local = np.zeros((H, W), np.int64)
    for y in range(H):
        for x in range(W):
            cy, cx, n = y, x, 0
            while n < H * W:
                s = step(cy, cx)
                if s is None or s == "out":
                    break
                cy, cx = s
                local[cy, cx] += 1
                n += 1
    A = np.zeros(len(ys), np.int64)
    xr = np.full(len(ys), -1, np.int32)
    code = np.zeros(len(ys), np.uint8)
    kind = np.full(len(ys), tiling.K_DEAD, np.uint8)
    ref = np.full(len(ys), -1, np.int32)
    nc = np.zeros(len(ys), np.int32)
    nd = np.zeros(len(ys), np.int32)
    zr = np.full(len(ys), -100, np.float32)
    ar = np.zeros(len(ys), np.int32)
    for i, (y, x) in enumerate(zip(ys, xs)):
        if step(y, x) == "out":
            A[i] = local[y, x] + 1
            code[i] = fdr_g[y0 + y, x0 + x]
        # flowacc: where does a path entering here leave the rank?
        cy, cx = int(y), int(x)
        for _ in range(H * W + 1):
            s = step(cy, cx)
            if s == "out":
                xr[i] = tiling.ring_index(H, W, np.array([cy]), np.array([cx]))[0]
                break
            if s is None:
                break
            cy, cx = s
        # flowhand: flowhand.py:566-846 semantics restricted to the rank
        cy, cx, c_, d_ = int(y), int(x), 0, 0
        if fdr_g[y0 + cy, x0 + cx] == 0:
            continue
        while True:
            if river_g[y0 + cy, x0 + cx] == 1:
                kind[i], ref[i], nc[i], nd[i] = tiling.K_RIVER, cy * W + cx, c_, d_
                zr[i], ar[i] = dem_g[y0 + cy, x0 + cx], acc_g[y0 + cy, x0 + cx]
                break
            c = int(fdr_g[y0 + cy, x0 + cx])
            s = step(cy, cx)
            if s is None:
                break
            ny, nx = (cy + _DY[c], cx + _DX[c])
            if fdr_g[y0 + ny, x0 + nx] == 0:
                break
            diag = _DY[c] != 0 and _DX[c] != 0
            c_, d_ = c_ + (0 if diag else 1), d_ + (1 if diag else 0)
            if s == "out":
                kind[i], ref[i], nc[i], nd[i] = tiling.K_REXIT, tiling.ring_index(H, W, np.array([cy]), np.array([cx]))[0], c_, d_
                break
            cy, cx = s
            if c_ + d_ > H * W:
                break
    return local, (A, xr, code), (kind, ref, nc, nd, zr, ar)


def _worker(rank, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    from descriptools_amd import tiling
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        layout = tiling.Layout([40], [64, 37])  # 1 x 2 ranks, ragged last column
        dem = oracle.synth_dem(9, 512, 512, 100, 100, layout.Hg, layout.Wg, 0)
        _, fdr = oracle.slope_d8(dem, 10.0)
        fdr[1::4, 58:70] = 1    # make plenty of paths cross the rank border (x = 64), eastwards ...
        fdr[3::4, 59:71] = 16   # ... and westwards
        acc_g = oracle.flowacc(fdr)
        assert (acc_g >= 0).all(), "test field must be acyclic"
        river = (acc_g > 60).astype(np.int8)
        local, fa, fh = _rank_model(layout, rank, fdr, river, dem, acc_g)
        # ---- flow accumulation: all-gather + solve + finish ----
        summ = tiling.all_gather_summaries(tuple(torch.as_tensor(a) for a in fa), layout, rank)
        ext = tiling.solve_flowacc(layout, summ)[rank]
        y0, x0 = layout.origin(rank)
        H, W = layout.shape(rank)
        ys, xs = tiling.ring_coords(H, W)
        acc = local.copy()
        for i in np.nonzero(ext)[0]:
            v = int(ext[i] & ~tiling.FA_CYCLE)
            cy, cx = int(ys[i]), int(xs[i])
            while True:
                acc[cy, cx] += v
                c = int(fdr[y0 + cy, x0 + cx])
                if c not in _DY:
                    break
                cy, cx = cy + _DY[c], cx + _DX[c]
                if not (0 <= cy < H and 0 <= cx < W):
                    break
        ok_acc = bool(np.array_equal(acc, acc_g[y0:y0 + H, x0:x0 + W]))
        # ---- HAND: all-gather (with ring codes) + solve ----
        codes = fdr[y0 + ys, x0 + xs]
        allfh = tiling.all_gather_summaries(tuple(torch.as_tensor(a) for a in fh) + (torch.as_tensor(codes),),
                                            layout, rank)
        res = tiling.solve_flowhand(layout, [s[:6] for s in allfh], [s[6] for s in allfh])[rank]
        fd_o, idx_o, hand_o = oracle.flowhand(dem, fdr, river, 10.0)
        # check every ring cell whose step leaves the rank against the oracle's untiled answer
        ok_fh, n_checked = True, 0
        for i, (y, x) in enumerate(zip(ys, xs)):
            c = int(fdr[y0 + y, x0 + x])
            if c not in _DY:
                continue
            gy, gx = y0 + y + _DY[c], x0 + x + _DX[c]
            if 0 <= y + _DY[c] < H and 0 <= x + _DX[c] < W:
                continue
            if not (0 <= gy < layout.Hg and 0 <= gx < layout.Wg):
                continue
            n_checked += 1
            want_idx = idx_o[gy, gx]  # the path continues from the cell it steps onto
            if fdr[gy, gx] == 0:
                continue
            if want_idx == -100:
                ok_fh &= res[0][i] == 0
            else:
                d = 10.0 * res[1][i] + 10.0 * np.sqrt(2.0) * res[2][i]
                ok_fh &= bool(res[0][i] == 1 and res[3][i] == want_idx and abs(d - fd_o[gy, gx]) < 1e-3
                              and res[4][i] == dem.reshape(-1)[want_idx] and res[5][i] == acc_g.reshape(-1)[want_idx])
        q.put((rank, ok_acc, bool(ok_fh), n_checked))
    finally:
        dist.destroy_process_group()


def _halo_worker(rank, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from descriptools_amd import tiling
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="4")
    dist.init_process_group("gloo", rank=rank, world_size=4)
    try:
        layout = tiling.Layout([64, 70], [128, 65])  # 2 x 2 ranks, ragged last row / column
        h = 64
        rng = np.random.default_rng(5)
        glob = rng.random((layout.Hg, layout.Wg)).astype(np.float32)
        y0, x0 = layout.origin(rank)
        H, W = layout.shape(rank)
        ext = torch.full((H + 2 * h, W + 2 * h), -7.0)
        ext[h:h + H, h:h + W] = torch.as_tensor(glob[y0:y0 + H, x0:x0 + W])
        tiling.exchange_halo(ext, layout, rank, h)
        # every extended cell inside the global raster must now hold the global value; the rest is untouched
        pad = np.full((layout.Hg + 2 * h, layout.Wg + 2 * h), -7.0, np.float32)
        pad[h:h + layout.Hg, h:h + layout.Wg] = glob
        want = pad[y0:y0 + H + 2 * h, x0:x0 + W + 2 * h]
        q.put((rank, bool(np.array_equal(ext.numpy(), want))))
    finally:
        dist.destroy_process_group()


def _halo_multi_worker(rank, port, q):
    """two processes, three logical ranks each (2 x 3 layout, interleaved ownership): neighbours in the other process
    over gloo send / recv, neighbours in the same process by copy -- tiling.exchange_halos"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from descriptools_amd import tiling
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        layout = tiling.Layout([128, 70], [64, 128, 65])
        proc_of = [0, 1, 0, 1, 0, 1]
        h = 64
        rng = np.random.default_rng(11)
        glob = rng.random((layout.Hg, layout.Wg)).astype(np.float32)
        exts = {}
        for r in range(layout.size):
            if proc_of[r] != rank:
                continue
            y0, x0 = layout.origin(r)
            H, W = layout.shape(r)
            e = torch.full((H + 2 * h, W + 2 * h), -7.0)
            e[h:h + H, h:h + W] = torch.as_tensor(glob[y0:y0 + H, x0:x0 + W])
            exts[r] = e
        tiling.exchange_halos(exts, layout, proc_of, h)
        pad = np.full((layout.Hg + 2 * h, layout.Wg + 2 * h), -7.0, np.float32)
        pad[h:h + layout.Hg, h:h + layout.Wg] = glob
        ok = True
        for r, e in exts.items():
            y0, x0 = layout.origin(r)
            H, W = layout.shape(r)
            ok &= bool(np.array_equal(e.numpy(), pad[y0:y0 + H + 2 * h, x0:x0 + W + 2 * h]))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_gloo_halo_exchange_three_logical_ranks_per_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_multi_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), sorted(res)


@pytest.mark.timeout(300)
def test_world4_gloo_halo_exchange():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), sorted(res)


@pytest.mark.timeout(300)
def test_world2_gloo_rank_level_solves():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_acc, ok_fh, n in sorted(res):
        assert ok_acc, "rank %d: tiled flow accumulation differs from the oracle" % rank
        assert ok_fh, "rank %d: rank-exit resolution differs from the oracle" % rank
        assert n > 5, "rank %d: too few border crossings exercised" % rank


def test_layout_and_ring_geometry():
    from descriptools_amd import tiling
    for H, W in [(5, 7), (1, 9), (6, 1), (2, 2), (64, 64)]:
        ys, xs = tiling.ring_coords(H, W)
        assert len(ys) == tiling.perim_count(H, W)
        assert np.array_equal(tiling.ring_index(H, W, ys, xs), np.arange(len(ys)))
    lay = tiling.Layout([64, 30], [128, 64, 10])
    assert lay.size == 6 and (lay.Hg, lay.Wg) == (94, 202)
    assert lay.origin(4) == (64, 128) and lay.shape(5) == (30, 10)
    assert np.array_equal(lay.owner(np.array([0, 63, 64, 93]), np.array([0, 127, 128, 201])), [0, 0, 4, 5])
    with pytest.raises(AssertionError):
        tiling.Layout([65, 64], [64])
