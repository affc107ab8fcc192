"""Randomised tiled == untiled check (logical ranks on one GPU, rank-level graphs solved on the GPU) over random
layouts.   python tools/stress_tiled.py [cases] [seed0]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from descriptools_amd import chain, tiling

NAMES = ["fdr", "fac", "river", "fdist", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"]
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for k in range(cases):
    rng = np.random.default_rng(seed0 + k)
    ty, tx = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    heights = [64 * int(rng.integers(1, 5)) for _ in range(ty)]
    widths = [64 * int(rng.integers(1, 5)) for _ in range(tx)]
    heights[-1] += int(rng.integers(0, 64)) * int(rng.integers(0, 2))   # ragged last row / column
    widths[-1] += int(rng.integers(0, 64)) * int(rng.integers(0, 2))
    layout = tiling.Layout(heights, widths)
    Hg, Wg = layout.Hg, layout.Wg
    dem = oracle.synth_dem(seed0 + k, 4096, 4096, int(rng.integers(0, 3000)), int(rng.integers(0, 3000)), Hg, Wg, int(rng.integers(0, 5)))
    px, thr = 10.0, max(1, (Hg * Wg) // int(rng.choice([64, 512])))
    ref = chain.run_host(dem, px, river_threshold=thr)
    h = tiling.HALO
    pad = np.full((Hg + 2 * h, Wg + 2 * h), np.nan, np.float32)
    pad[h:h + Hg, h:h + Wg] = dem
    tiles = []
    for r in range(layout.size):
        # every other case with the long-walk workspace (queue + skip tables over core + halo)
        t = tiling.RankTile(layout, r, device=0, px=px, river_threshold=thr, long_walks=(k % 2 == 1))
        y0, x0 = layout.origin(r)
        t.set_dem_ext(pad[y0:y0 + t.He, x0:x0 + t.We])
        tiles.append(t)
    tiling.simulate_dev(tiles, layout)
    for t in tiles:
        y0, x0 = layout.origin(t.rank)
        sl = (slice(y0, y0 + t.H), slice(x0, x0 + t.W))
        for name in NAMES:
            got, want = t.host(name), ref[name][sl]
            if not np.array_equal(got, want.astype(got.dtype), equal_nan=True):
                print("MISMATCH", k, heights, widths, "rank", t.rank, name, int((got != want).sum())); bad += 1
        if not np.array_equal(t.host("idx"), ref["idx"][sl]):
            print("MISMATCH", k, heights, widths, "rank", t.rank, "idx"); bad += 1
        if t.unresolved_downslope():
            print("UNRESOLVED downslope", k, t.rank); bad += 1
    if k % 5 == 0:
        print("case", k, heights, widths, "ok" if bad == 0 else "BAD %d" % bad, flush=True)
print("cases", cases, "mismatches", bad)
sys.exit(1 if bad else 0)
