"""One rank of tests/test_gpu_run_rank.py (launched by torch.distributed.run): builds its RankTile, fills the DEM
halo (generated locally, or exchanged point-to-point with the neighbouring ranks), drives tiling.run_rank -- the
product's N > 1 step -- and writes its core rasters for the parent to compare with the untiled chain."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rough_dem(Hg, Wg, seed):
    """synthetic terrain with noise, pits and integer plateaus (depressions and flats across the rank borders): what
    the conditioning is for; the same array in the ranks and in the parent test"""
    import oracle  # test infrastructure: the generator on the host
    base = oracle.synth_dem(seed, Hg, Wg, 0, 0, Hg, Wg, 2)
    rng = np.random.default_rng(seed)
    nod = base == -100
    dem = np.floor(base + rng.normal(0, 4.0, base.shape).astype(np.float32)).astype(np.float32)
    dem[rng.random(dem.shape) < 0.02] -= 30
    dem[nod] = -100
    return dem


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--h", type=int, required=True)
    ap.add_argument("--w", type=int, required=True)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--nodata", type=int, default=0)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--halo", default="synth", choices=["synth", "exchange"])
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--force-world", type=int, default=0)
    ap.add_argument("--terrain", default="synth", choices=["synth", "plane", "rough"])
    ap.add_argument("--condition", type=int, default=0, help="tiling.condition_rank (depression filling + flat routing "
                    "over the ranks: point-to-point halo exchanges, an all-reduce of the flag per iteration), then "
                    "run_rank(d8=False) on the conditioned codes")
    ap.add_argument("--logical", type=int, default=0, help="this ONE process plays that many logical ranks: their DEM "
                    "halos are exchanged with isend / irecv pairs to itself over the process group (RCCL: the "
                    "point-to-point path of the 8-GPU run on a 1-GPU box), then the ranks step in lock-step")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from descriptools_amd import tiling
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # the logical ranks share the one GPU of the box
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    if a.logical:
        return logical_ranks(a, torch, dist, tiling)
    layout = tiling.Layout.uniform(world, a.h, a.w)
    thr = (layout.Hg * layout.Wg) // 512
    # (the plane terrain runs downslope with the long-walk workspace: the queue and the tables in real processes)
    tile = tiling.RankTile(layout, rank, device=0, px=10.0, river_threshold=thr,
                           long_walks=(a.terrain == "plane" or bool(a.condition)))
    if a.terrain == "plane":
        # a 1 per mille plane: every downslope walk is thousands of moves long and crosses the rank borders
        h = tiling.HALO
        y0, x0 = layout.origin(rank)
        yy, xx = np.mgrid[y0 - h:y0 + tile.H + h, x0 - h:x0 + tile.W + h]
        tile.set_dem_ext((200.0 - 0.001 * xx - 0.0002 * yy).astype(np.float32))
    elif a.terrain == "rough":
        h = tiling.HALO
        y0, x0 = layout.origin(rank)
        full = rough_dem(layout.Hg, layout.Wg, a.seed)
        ext = torch.full((tile.He, tile.We), float("nan"))  # NaN = never received, must never be read
        ext[h:h + tile.H, h:h + tile.W] = torch.as_tensor(full[y0:y0 + tile.H, x0:x0 + tile.W])
        tiling.exchange_halo(ext, layout, rank)
        tile.set_dem_ext(torch.nan_to_num(ext, nan=-100.0).numpy())  # outside the raster: nodata
    elif a.halo == "synth":
        tile.synth_dem(a.seed, a.nodata)
    else:
        import oracle  # test infrastructure: the same generator on the host
        h = tiling.HALO
        y0, x0 = layout.origin(rank)
        core = oracle.synth_dem(a.seed, layout.Hg, layout.Wg, y0, x0, tile.H, tile.W, a.nodata)
        ext = torch.full((tile.He, tile.We), float("nan"))  # NaN = never received, must never be read
        ext[h:h + tile.H, h:h + tile.W] = torch.as_tensor(core)
        tiling.exchange_halo(ext, layout, rank)  # gloo: CPU tensors (RCCL: the same call on device tensors)
        tile.set_dem_ext(ext.numpy())
    exchange = tiling.Exchange(tile, layout, max(world, a.force_world))
    if a.condition:
        left, it_fill, it_flat = tiling.condition_rank(tile, layout)
        assert left == 0 and it_fill >= 1 and it_flat >= 1
    for _ in range(2):  # twice: the step reuses its buffers
        tiling.run_rank(tile, layout, exchange, overlap=bool(a.overlap), d8=not a.condition)
    tile.check_status()
    # walks that left this rank's memory travel on as walkers (none on the synthetic terrain: returns 0 at once)
    sent = tiling.finish_downslope(tile, tiling.DistComm())
    assert (sent > 0) == (a.terrain == "plane") or a.condition
    assert tile.unresolved_downslope() == 0
    names = ["dem", "fdr", "fac", "river", "fdist", "idx", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"]
    np.savez(os.path.join(a.out, "rank%d.npz" % rank), origin=np.array(layout.origin(rank)),
             **{n: tile.host(n) for n in names})
    dist.barrier()
    dist.destroy_process_group()


def logical_ranks(a, torch, dist, tiling):
    import oracle  # test infrastructure: the same generator on the host
    layout = tiling.Layout.uniform(a.logical, a.h, a.w)
    thr = (layout.Hg * layout.Wg) // 512
    h = tiling.HALO
    on_gpu = a.backend == "nccl"
    tiles, exts = [], {}
    for r in range(layout.size):
        t = tiling.RankTile(layout, r, device=0, px=10.0, river_threshold=thr)
        y0, x0 = layout.origin(r)
        core = oracle.synth_dem(a.seed, layout.Hg, layout.Wg, y0, x0, t.H, t.W, a.nodata)
        ext = torch.full((t.He, t.We), float("nan"), device="cuda" if on_gpu else "cpu")
        ext[h:h + t.H, h:h + t.W] = torch.as_tensor(core).to(ext.device)
        tiles.append(t)
        exts[r] = ext
    torch.cuda.synchronize()
    tiling.exchange_halos(exts, layout, [0] * layout.size)  # every neighbour is this process: isend / irecv to itself
    torch.cuda.synchronize()
    for t in tiles:
        t.set_dem_ext(exts[t.rank].cpu().numpy())
    for _ in range(2):
        tiling.run_ranks_local(tiles, layout)  # rank_ops(): the serial schedule bench.py times, stage by stage
    names = ["dem", "fdr", "fac", "river", "fdist", "idx", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"]
    for t in tiles:
        t.check_status()
        assert t.unresolved_downslope() == 0
        np.savez(os.path.join(a.out, "rank%d.npz" % t.rank), origin=np.array(layout.origin(t.rank)),
                 **{n: t.host(n) for n in names})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
