"""Downslope on REAL terrain IN RANKS (one GPU plays every rank in turn): the bundled Example with its GIS D8 raster
tiled 4 x 4, cropped to multiples of 64 and split 2 x 2.  Per rank: the window kernel's time and the walks that left
the rank's memory (the ones tiling.finish_downslope carries on as walkers).
   python tools/ds_ranks_real.py [plain] [check] [finish]     finish: then tiling.finish_downslope; plain: without the long-walk workspace; check: against the untiled raster"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from conftest import load_example  # noqa: E402
from descriptools_amd import tiling  # noqa: E402

reps = 5
ex = load_example()
dem0, fdr0 = np.asarray(ex[0], np.float32), np.ascontiguousarray(ex[1], np.uint8)
dem, fdr = np.tile(dem0, (4, 4))[:8704, :6080], np.tile(fdr0, (4, 4))[:8704, :6080]
Hg, Wg = dem.shape
layout = tiling.Layout([3264, 5440], [2304, 3776])  # borders through the middle of Example copies (their rims are nodata)
h = tiling.HALO
dem_p = np.pad(dem, h, constant_values=-100.0)
fdr_p = np.pad(fdr, h, constant_values=0)
ref = None
if "check" in sys.argv:  # the untiled raster, to compare the cells the ranks resolved themselves
    from descriptools_amd import downslope as ds_mod
    ref = ds_mod.downsloper(dem, fdr, 12.5, 5.0)
long_walks = "plain" not in sys.argv
total = 0.0
tiles = []
for r in range(layout.size):
    tile = tiling.RankTile(layout, r, px=12.5, dz=5.0, rasters=("dem", "fdr", "down"), tune_placement=False,
                           long_walks=long_walks)
    gy0, gx0 = layout.origin(r)
    tile.set_dem_ext(dem_p[gy0:gy0 + tile.He, gx0:gx0 + tile.We])
    with tile.on_stream():
        tile.t["fdr"].copy_(torch.as_tensor(np.ascontiguousarray(fdr_p[gy0:gy0 + tile.He, gx0:gx0 + tile.We])))
    tile.ctx.sync()
    ts = []
    for _ in range(reps):
        tile.n_unres.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tile.downslope()
        tile.ctx.sync()
        ts.append(time.perf_counter() - t0)
    un = tile.unresolved_downslope()
    total += min(ts)
    if ref is not None:
        got = tile.host("down")
        want = ref[gy0:gy0 + tile.H, gx0:gx0 + tile.W]
        own = got != -50
        assert int((~own).sum()) == un and np.array_equal(got[own].view(np.int32), want[own].view(np.int32)), "rank %d" % r
    print("rank %d (%d x %d at %d, %d): downslope %.2f ms, %d walks leave the rank's memory (%.2f %% of its cells)"
          % (r, tile.H, tile.W, gy0, gx0, min(ts) * 1e3, un, 100.0 * un / (tile.H * tile.W)), flush=True)
    if "finish" in sys.argv:
        tiles.append(tile)
    else:
        tile.free()
if tiles:  # the walks that left their rank travel on as walkers (one thread plays each rank)
    import threading
    comms = tiling.LocalComm.create(layout.size)
    done = [None] * layout.size
    stats = [dict() for _ in range(layout.size)]
    def work(r):
        done[r] = tiling.finish_downslope(tiles[r], comms[r], stats=stats[r])
    for rnd in ("first call (torch loads its kernels, its allocator asks the driver for every block)", "second call",
                "third call"):
        t0 = time.perf_counter()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(layout.size)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        print("finish_downslope, %s: %d walkers in %.1f ms (four threads play the ranks; walker records emitted by the "
              "downslope kernel, advanced / routed / grouped on the device, exchanged as device buffers)" % (rnd, done[0], (time.perf_counter() - t0) * 1e3), flush=True)
        print("   rank 0: %d iterations; ms: " % stats[0]["iterations"] +
              ", ".join("%s %.2f" % (k[2:], v * 1e3) for k, v in stats[0].items() if k.startswith("s_")), flush=True)
        if rnd != "third call":
            for tile in tiles:  # the same step again
                tile.downslope()
                tile.ctx.sync()
    if ref is not None:
        for tile in tiles:
            gy0, gx0 = layout.origin(tile.rank)
            assert np.array_equal(tile.host("down").view(np.int32), ref[gy0:gy0 + tile.H, gx0:gx0 + tile.W].view(np.int32))
        print("every rank's raster equals the untiled one")
print("all four ranks%s: %.2f ms" % ("" if long_walks else " (without the long-walk workspace)", total * 1e3))
