"""Randomised differential test, GPU path vs oracle, over many shapes / seeds (not part of the pytest suites:
minutes of oracle time).  Exits non-zero on the first mismatch.   python tools/stress_parity.py [cases] [seed0] [max_side]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import descriptools_amd.slope as slope
import descriptools_amd.flowdir as flowdir
import descriptools_amd.flowacc as flowacc
import descriptools_amd.flowhand as flowhand
import descriptools_amd.downslope as downslope
from descriptools_amd import chain, tiling

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 700
codes = np.array([1, 2, 4, 8, 16, 32, 64, 128], np.uint8)
bad = 0
for k in range(cases):
    rng = np.random.default_rng(seed0 + k)
    H, W = int(rng.integers(3, max_side)), int(rng.integers(3, max_side))
    mode = k % 4
    if max_side > 1000:
        mode = 0 if k % 2 == 0 else 3  # terrain only: the oracle's literal walks on adversarial fields take hours
    px = float(rng.choice([10.0, 12.5, 30.0, 1.0]))
    if mode == 0:      # synthetic terrain with nodata, the chain's own D8
        dem = oracle.synth_dem(seed0 + k, 4096, 4096, int(rng.integers(0, 3000)), int(rng.integers(0, 3000)), H, W, int(rng.integers(0, 6)))
        sl_o, fdr = oracle.slope_d8(dem, px)
        assert np.array_equal(slope.sloper(dem, px).astype(np.float32), sl_o), (k, "slope")
        assert np.array_equal(flowdir.d8(dem, px), fdr), (k, "d8")
    elif mode == 1:    # random direction field with cycles and a few invalid codes, rough terrain
        fdr = codes[rng.integers(0, 8, size=(H, W))]
        fdr[rng.random((H, W)) < 0.4] = 4
        fdr[rng.random((H, W)) < 0.01] = rng.choice([0, 3, 255])
        dem = (rng.random((H, W)) * rng.choice([1.0, 40.0, 3000.0])).astype(np.float32)
        dem[rng.random((H, W)) < 0.02] = -100
    elif mode == 2:    # long serpentine paths (wide HAND words, downslope far beyond the window)
        fdr = np.zeros((H, W), np.uint8)
        for y in range(H):
            fdr[y, :] = 1 if y % 2 == 0 else 16
            fdr[y, W - 1 if y % 2 == 0 else 0] = 4
        dem = np.full((H, W), 7.0, np.float32) - np.arange(H * W, dtype=np.float32).reshape(H, W) * np.float32(rng.choice([0.0, 0.001, 0.05]))
    else:              # integer DEM (int16 like the Example), D8 from it
        dem = (oracle.synth_dem(seed0 + k, 2048, 2048, 0, 0, H, W, 2)).astype(np.int16).astype(np.float32)
        dem[dem == -100] = -100
        _, fdr = oracle.slope_d8(dem, px)
    acc_o = oracle.flowacc(fdr, dem)
    acc = flowacc.accumulate(fdr, dem)
    if not np.array_equal(acc, acc_o):
        print("MISMATCH flowacc", k, H, W, mode, int((acc != acc_o).sum())); bad += 1
    river = (acc_o > max(1, (H * W) // int(rng.choice([64, 512, 4096])))).astype(np.int8)
    if mode == 1:
        river = (rng.random((H, W)) < 0.002).astype(np.int8)   # arbitrary masks too
    fd_o, idx_o, hand_o = oracle.flowhand(dem, fdr, river, px)
    fd, idx, hand = flowhand.flow_hand_index(dem, fdr, river, px)
    ok = np.array_equal(idx, idx_o) and np.array_equal(hand, hand_o) and np.array_equal(fd == -100, fd_o == -100) \
        and np.allclose(fd, fd_o, rtol=1e-6, atol=0)
    if not ok:
        print("MISMATCH flowhand", k, H, W, mode, int((idx != idx_o).sum()), int((hand != hand_o).sum())); bad += 1
    dz = float(rng.choice([5.0, 0.3, 50.0]))
    ds_o = oracle.downslope(dem, fdr, px, dz)
    ds = downslope.downsloper(dem, fdr, px, dz)
    if not np.array_equal(ds, ds_o, equal_nan=True):
        print("MISMATCH downslope", k, H, W, mode, dz, int((ds != ds_o).sum())); bad += 1
    if k % 10 == 0 or max_side > 1000:
        print("case", k, H, W, "mode", mode, "ok" if bad == 0 else "BAD %d" % bad, flush=True)
print("cases", cases, "mismatching ops", bad)
sys.exit(1 if bad else 0)
