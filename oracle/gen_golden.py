#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's unmodified source -- TEST INFRASTRUCTURE.

Runs only in the build container (needs /root/reference, which never travels):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [case ...]

The reference's modules are imported by path with numba replaced by the interpretive
stand-in in oracle/numba_standin/ (numba itself is not installed here).  Inputs and the
reference's outputs are stored; nothing of the reference's source is.

Arithmetic fidelity: Numba promotes ``float32 (op) float64`` to float64 inside kernels, while
numpy-2 scalar arithmetic under the stand-in would stay in float32; float32 rasters are
therefore handed to the reference as float64 copies (exact), which reproduces Numba's
promote-then-compute arithmetic on inputs whose DEM differences are float32-exact (all the
fixtures here: int16 Example data and 1/256-m synthetic heights).

D8 / flow accumulation do not exist in the reference; for the synthetic cases they come from
the build's own oracle (oracle/dt_oracle.c) and are stored as INPUTS of the golden file.
"""
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "numba_standin"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)

import numba.cuda as standin_cuda  # noqa: E402  (the stand-in)
import descriptools.slope as R_slope  # noqa: E402
import descriptools.topoindexes as R_topo  # noqa: E402
import descriptools.flowhand as R_flowhand  # noqa: E402
import descriptools.gfi as R_gfi  # noqa: E402
import descriptools.downslope as R_down  # noqa: E402
import descriptools.evaluation as R_eval  # noqa: E402
import descriptools.helpers as R_helpers  # noqa: E402

import oracle  # noqa: E402  (only for synthetic inputs: DEM, D8, flow accumulation)

GOLD = os.path.join(ROOT, "tests", "golden")
warnings.filterwarnings("ignore")


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("  wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024), flush=True)


def load_example():
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    ex = os.path.join(GOLD, "example")
    dem_f = np.array(Image.open(os.path.join(ex, "12_dem.tif")))
    fac_f = np.array(Image.open(os.path.join(ex, "12_fac.tif")))
    fdr = np.array(Image.open(os.path.join(ex, "12_fdr.tif"))).astype(np.uint8)
    flood = np.array(Image.open(os.path.join(ex, "WB_12_100y.tif"))).astype(np.int8)
    klass = np.array(Image.open(os.path.join(ex, "hand_class.tif"))).astype(np.uint8)
    # example.py:33-43 -- int16 DEM / int64 fac, nodata (the value at [0,0]) -> -100
    dem = np.where(dem_f < -1e30, -100, dem_f).astype(np.int16)
    fac = np.where(fac_f < -1e30, -100, fac_f).astype(np.int64)
    return dem, fdr, fac, flood, klass


def run_chain(dem, fdr, fac, river, px, n_top=0.1, n_gfi=0.4, b=0.1, dz=5):
    """example.py:59-91 on one raster.  `dem` int16 or float64-copy-of-float32."""
    out = {}
    t = time.time()
    sl = R_slope.sloper(dem, px).astype("float32")
    out["slope"] = sl
    slr = np.arctan(sl / 100).astype("float32")
    slr = np.where(dem == -100, -100, slr).astype("float32")
    out["slope_rad"] = slr
    # float64 copy so that "slope + 0.01" is float64 as under Numba (topoindexes.py:257)
    ti, mti = R_topo.topographic_index(fac, slr.astype(np.float64), px, n_top)
    out["ti"], out["mti"] = ti.astype(np.float32), mti.astype(np.float32)
    down = R_down.downsloper(dem, fdr, px, dz)
    out["down"] = down.astype(np.float32)
    flow, idx, hand = R_flowhand.flow_hand_index(dem, fdr, river, px)
    out["fdist"], out["idx"], out["hand"] = flow.astype(np.float32), idx.astype(np.int64), hand
    g = R_gfi.gfi_calculator(hand, fac, idx, n_gfi, b, px)
    out["gfi"] = g.astype(np.float32)
    l = R_gfi.ln_hl_H_calculator(hand, fac, n_gfi, b, px)
    out["lnhlh"] = l.astype(np.float32)
    print("   chain %dx%d in %.1fs" % (dem.shape[0], dem.shape[1], time.time() - t), flush=True)
    return out


# ------------------------------------------------------------------------------------------
def case_synth():
    """G-syn: synthetic DEM windows, f32 heights (exact 1/256 m), with and without nodata."""
    for name, seed, H, W, nod, px, thr in [("syn_a", 1, 96, 128, 0, 10.0, 40),
                                           ("syn_b", 2, 120, 88, 6, 10.0, 30),
                                           ("syn_c", 3, 64, 64, 0, 12.5, 20)]:
        print(name, flush=True)
        Hg, Wg = 1024, 1024
        y0, x0 = 300, 200
        dem32 = oracle.synth_dem(seed, Hg, Wg, y0, x0, H, W, nod)
        _, fdr = oracle.slope_d8(dem32, px)
        fac = oracle.flowacc(fdr, dem32)
        river = (fac > thr).astype(np.int8)
        out = run_chain(dem32.astype(np.float64), fdr, fac, river, px)
        out["hand"] = out["hand"].astype(np.float32)
        save(name, dem=dem32, fdr=fdr, fac=fac, river=river, px=px, n_top=0.1, n_gfi=0.4, b=0.1,
             dz=5.0, **out)


def case_f64():
    """A genuinely float64 DEM (heights that are NOT float32 values: the synthetic terrain plus a smooth float64
    ripple; no float64 <- float32 copy): the reference takes every height difference in float64 here (slope.py:244-258,
    flowhand.py:436-438, downslope.py:468).  D8 / accumulation inputs come from the float32-rounded terrain's oracle run
    (they are INPUTS of the reference).  Pins what the float32 boundary of the build gives up on such a raster."""
    print("f64", flush=True)
    H, W, px, thr = 80, 96, 10.0, 25
    dem32 = oracle.synth_dem(5, 1024, 1024, 400, 100, H, W, 4)
    yy, xx = np.mgrid[0:H, 0:W]
    ripple = 1e-3 * np.sin(0.37 * yy + 0.11 * xx) + 1e-6 * np.cos(1.3 * xx)   # sub-float32-ulp structure at ~200 m
    dem64 = np.where(dem32 == -100, -100.0, dem32.astype(np.float64) + ripple)
    assert (dem64.astype(np.float32).astype(np.float64) != dem64).sum() > 0.9 * (dem32 != -100).sum()
    _, fdr = oracle.slope_d8(dem64.astype(np.float32), px)
    fac = oracle.flowacc(fdr, dem64.astype(np.float32))
    river = (fac > thr).astype(np.int8)
    out = run_chain(dem64, fdr, fac, river, px)
    assert out["hand"].dtype == np.float64
    save("f64", dem=dem64, fdr=fdr, fac=fac, river=river, px=px, n_top=0.1, n_gfi=0.4, b=0.1, dz=5.0, **out)


def case_example_windows():
    """G-ex: windows of the bundled Example rasters (int16 DEM), each run as its own raster."""
    dem, fdr, fac, flood, _ = load_example()
    river = np.where(fac > 128000, 1, 0).astype("int8")
    ry, rx = np.nonzero(river)
    k = len(ry) // 2
    wins = {"ex_river": (max(ry[k] - 80, 0), max(rx[k] - 70, 0), 160, 144)}
    valid = dem != -100
    best = None
    for y in range(0, dem.shape[0] - 128, 64):
        for x in range(0, dem.shape[1] - 128, 64):
            fr = valid[y:y + 128, x:x + 128].mean()
            nr = river[y:y + 128, x:x + 128].sum()
            if 0.35 < fr < 0.65 and nr > 20 and best is None:
                best = (y, x, 128, 128)
            if fr == 1.0 and nr == 0 and "ex_head" not in wins:
                wins["ex_head"] = (y, x, 112, 128)
    wins["ex_edge"] = best
    for name, (y, x, h, w) in wins.items():
        print(name, (y, x, h, w), flush=True)
        d, f, a, r = (dem[y:y + h, x:x + w].copy(), fdr[y:y + h, x:x + w].copy(),
                      fac[y:y + h, x:x + w].copy(), river[y:y + h, x:x + w].copy())
        out = run_chain(d, f, a, r, 12.5)
        save(name, dem=d, fdr=f, fac=a, river=r, px=12.5, n_top=0.1, n_gfi=0.4, b=0.1, dz=5.0,
             window=np.array([y, x, h, w]), **out)


def case_edge():
    """G-edge: hand-made micro cases for every early-out of the two walk kernels."""
    print("edge", flush=True)
    res = {}
    px = 10.0
    # (1) flow-distance micro rasters: 2-cycle, 3-cycle, arrival on fdr==0 (also when that cell is
    # a river), raster exits on all four sides, non-D8 code, river cell, diagonal chain.
    fdr = np.array([[1, 16, 4, 0, 2, 1],     # (0,0)<->(0,1) 2-cycle; (0,2) S; (0,3) nodata
                    [1, 8, 4, 16, 3, 1],     # (1,4) non-D8 code 3; (1,5) exits E
                    [64, 1, 2, 4, 16, 4],    # (2,0) N -> (1,0) -> (1,1) SW -> (2,0): 3-cycle
                    [2, 1, 1, 1, 128, 32],
                    [4, 2, 1, 1, 1, 64]], np.uint8)
    # embed in 10 x 12 (the reference indexes dem[-100]: rasters need >= 100 cells)
    big = np.full((10, 12), 1, np.uint8)
    big[:5, :6] = fdr
    big[5:, :] = 4
    big[7, 3] = 128
    fdr = big
    river = np.zeros(fdr.shape, np.int8)
    river[3, 3] = 1
    river[0, 3] = 1          # river cell with fdr == 0
    river[4, 4] = 1
    river[8, 7] = 1
    river[2, 9] = 1
    dem = (np.arange(120, dtype=np.int16).reshape(10, 12)[::-1, ::-1] * 3 + 7).astype(np.int16)
    dem[0, 3] = -100
    dem[9, 7] = 500          # river cell's successor irrelevant; cell above river lower than river
    dem[7, 7] = 3            # negative HAND -> clipped to 0
    f, i, h = R_flowhand.flow_hand_index(dem, fdr, river, px)
    res.update(fh_dem=dem, fh_fdr=fdr, fh_river=river, fh_fdist=f.astype(np.float32),
               fh_idx=i.astype(np.int64), fh_hand=h)
    # (2) 20000-move cap: 1 x 20012 raster, all E, river at the last cell; only threads 0..14
    # are executed (path lengths 20011 .. 19997).
    W = 20012
    fdr2 = np.ones((1, W), np.uint8)
    river2 = np.zeros((1, W), np.int8)
    river2[0, W - 1] = 1
    dem2 = np.full((1, W), 5, np.int16)
    sel = list(range(15))
    standin_cuda.THREAD_FILTER = sel
    f2, i2, _ = R_flowhand.flow_hand_index(dem2, fdr2, river2, px)
    standin_cuda.THREAD_FILTER = None
    res.update(cap_W=W, cap_sel=np.array(sel), cap_fdist=f2[0, sel].astype(np.float32),
               cap_idx=i2[0, sel].astype(np.int64))
    # diagonal variant (sqrt(2) accumulation over a long path): 400 x 400, all SE, river at the
    # SE corner; threads on the main diagonal only.
    n = 400
    fdr3 = np.full((n, n), 2, np.uint8)
    river3 = np.zeros((n, n), np.int8)
    river3[n - 1, n - 1] = 1
    dem3 = np.full((n, n), 5, np.int16)
    sel3 = [k * n + k for k in range(0, n, 7)] + [3, n * 5]
    standin_cuda.THREAD_FILTER = sel3
    f3, i3, _ = R_flowhand.flow_hand_index(dem3, fdr3, river3, 12.5)
    standin_cuda.THREAD_FILTER = None
    res.update(diag_n=n, diag_sel=np.array(sel3), diag_fdist=f3.reshape(-1)[sel3].astype(np.float32),
               diag_idx=i3.reshape(-1)[sel3].astype(np.int64))
    # (3) downslope micro raster: edge exits with and without a move, nodata ahead
    # (stop-before-move), normal termination, plateau start.
    demd = np.array([[50, 48, 46, 44, 20],
                     [49, 47, -100, 43, 19],
                     [48, 46, 44, 42, 18],
                     [30, 29, 28, 27, 17]], np.int16)
    fdrd = np.array([[1, 1, 4, 1, 4],
                     [4, 1, 0, 2, 4],
                     [4, 2, 64, 1, 4],
                     [1, 1, 1, 1, 4]], np.uint8)
    dd = R_down.downsloper(demd, fdrd, px, 5)
    res.update(ds_dem=demd, ds_fdr=fdrd, ds_out=dd.astype(np.float32))
    # (4) 5000-iteration cap: 1 x 5012 raster, heights falling by 1 every 1500 cells (drop never
    # reaches 5 within 5000 moves); only threads 0..6 run on the GPU kernel, the CPU repair then
    # re-walks exactly those (all other cells keep the initial 0 != -50).
    W = 5012
    demc = (10 - (np.arange(W) // 1500)).astype(np.int16).reshape(1, W)
    fdrc = np.ones((1, W), np.uint8)
    selc = list(range(7)) + [W - 3000, W - 1]
    standin_cuda.THREAD_FILTER = selc
    dc = R_down.downsloper(demc, fdrc, px, 5)
    standin_cuda.THREAD_FILTER = None
    res.update(dcap_dem=demc, dcap_sel=np.array(selc), dcap_out=dc[0, selc].astype(np.float32))
    # (5) pointwise special values: fac 0 / -100, slope nodata, hand 0 / -100 (negative fac and
    # zero river area are NaN / -inf under CUDA but raise under the interpretive stand-in: not pinned)
    pad = 100
    fac = np.array([[0, 1, 5, -100, 3, 1000000, 7, 0] + [11] * pad], np.int64)
    slr = np.array([[0.0, 0.2, 1.4, 0.3, 0.3, 1.55, -100.0, 1.5607] + [0.05] * pad], np.float32)
    ti, mti = R_topo.topographic_index(fac, slr.astype(np.float64), 12.5, 0.1)
    hand = np.array([[0, 3, -100, -100, 1, 250, 0, 7] + [2] * pad], np.int16)
    idx = np.array([[1, 5, -100, 1, 2, 5, 6, 8] + [9] * pad], np.int64)
    g = R_gfi.gfi_calculator(hand, fac, idx, 0.4, 0.1, 12.5)
    l = R_gfi.ln_hl_H_calculator(hand, fac, 0.4, 0.1, 12.5)
    res.update(pw_fac=fac, pw_slr=slr, pw_ti=ti.astype(np.float32), pw_mti=mti.astype(np.float32),
               pw_hand=hand, pw_idx=idx, pw_gfi=g.astype(np.float32), pw_lnhlh=l.astype(np.float32))
    # (6) helpers.divisor
    res.update(div_a=np.array(R_helpers.divisor(100, 37, 3, 2)[0]),
               div_b=np.array(R_helpers.divisor(100, 37, 3, 2)[1]))
    save("edge", **res)


def case_eval():
    """G-eval: evaluation.py on random descriptors (pure numpy in the reference: exact)."""
    print("eval", flush=True)
    rng = np.random.default_rng(7)
    res = {}
    for k, (shape, under) in enumerate([((64, 80), "under"), ((50, 50), "over"),
                                        ((40, 96), "under")]):
        hand = rng.integers(0, 60, size=shape).astype(np.int16)
        hand[rng.random(shape) < 0.2] = -100
        hand[0, 0] = -100
        flood = (rng.random(shape) < 0.3).astype(np.int8)
        flood[(hand > 25) & (rng.random(shape) < 0.8)] = 0
        if k == 2:
            flood[rng.random(shape) < 0.05] = -100
        el = np.unique(hand)
        mn, mx = el[1], el[-1]
        desc = R_eval.minMaxScale(hand, mn, mx, -100)
        fl = flood.copy()
        th = R_eval.calibration(desc, fl, under)
        binary = R_eval.binary_map(desc, th, under)
        fl2 = flood.copy()
        c, f, cm = R_eval.avaliacao(binary, fl2)
        res.update({"e%d_hand" % k: hand, "e%d_flood" % k: flood, "e%d_under" % k: under,
                    "e%d_mn" % k: mn, "e%d_mx" % k: mx, "e%d_desc" % k: desc, "e%d_th" % k: th,
                    "e%d_binary" % k: binary.astype(np.int8), "e%d_c" % k: c, "e%d_f" % k: f,
                    "e%d_class" % k: cm.astype(np.int8), "e%d_flood_after" % k: fl2})
    save("eval", **res)


def case_example_full():
    """The reference's only known-answer test: HAND -> minMaxScale -> calibration -> binary_map ->
    avaliacao must reproduce Example/output/hand_class.tif (example.py:82-147).  ~5 min."""
    print("example_full", flush=True)
    dem, fdr, fac, flood, klass = load_example()
    river = np.where(fac > 128000, 1, 0).astype("int8")
    t = time.time()
    flow, idx, hand = R_flowhand.flow_hand_index(dem, fdr, river, 12.5)
    print("   flow_hand_index %.0fs" % (time.time() - t), flush=True)
    el = np.unique(hand)
    mx, mn = el[-1], el[1]
    desc = R_eval.minMaxScale(hand, mn, mx, -100)
    fl = flood.copy()
    th = R_eval.calibration(desc, fl, "under")
    binary = R_eval.binary_map(desc, th, "under")
    c, f, cm = R_eval.avaliacao(binary, fl)
    mism = int((cm.astype(np.uint8) != klass).sum())
    print("   mn,mx=%s,%s th=%r c=%r f=%r mismatches vs hand_class.tif=%d" % (mn, mx, th, c, f, mism),
          flush=True)
    assert mism == 0
    # exact full-size outputs (int) + float distance; floats of the other descriptors are
    # covered by the windows above.
    save("example_full", idx=idx.astype(np.int32), hand=hand.astype(np.int16),
         fdist=flow.astype(np.float32), mn=mn, mx=mx, th=th, c=c, f=f,
         counts=np.bincount(cm.reshape(-1).astype(np.int64), minlength=4))



def case_shims():
    """G-shim: the tile-level shims called the way the reference's own tile loops call them
    (slope.py:126-147, flowhand.py:293-405): slope_cpu with all 16 `extra` combinations, sloper with divisions,
    flow_distance_index_cpu with neighbouring tiles on every combination of sides (boundary vectors, global
    index arithmetic with row_start / col_start / matrix_columns)."""
    print("shims", flush=True)
    res = {}
    px = 10.0
    dem32 = oracle.synth_dem(4, 1024, 1024, 100, 100, 40, 56, 5)
    dem = dem32.astype(np.float64)
    res["sl_dem"] = dem32
    standin_cuda.TOLERATE_UNBOUND = True
    try:
        k = 0
        for u in (0, 1):
            for l in (0, 1):
                for r in (0, 1):
                    for d in (0, 1):
                        mS, mE = (0 if u else 10), (40 if d else 30)
                        nS, nE = (0 if l else 12), (56 if r else 44)
                        tile = dem[mS - 1 + u:mE + 1 - d, nS - 1 + l:nE + 1 - r]
                        out = R_slope.slope_cpu(tile, px, np.array([u, l, r, d]))
                        res["sl_extra%d" % k] = np.array([u, l, r, d, mS, mE, nS, nE])
                        res["sl_out%d" % k] = out.astype(np.float32)
                        k += 1
        res["sl_tiled22"] = R_slope.sloper(dem, px, 2, 2).astype(np.float32)
    finally:
        standin_cuda.TOLERATE_UNBOUND = False
    res["sl_untiled"] = R_slope.sloper(dem, px).astype(np.float32)
    # flow distance: a 48 x 60 raster solved untiled, then tiles of it through flow_distance_index_cpu with the
    # ring of the untiled solution as boundary vectors, exactly as flowhand.py:313-390 slices them
    H, W = 48, 60
    dem32 = oracle.synth_dem(6, 1024, 1024, 500, 300, H, W, 4)
    _, fdr = oracle.slope_d8(dem32, px)
    fac = oracle.flowacc(fdr, dem32)
    river = (fac > 25).astype(np.int8)
    fd_full, idx_full, _ = R_flowhand.flow_hand_index(dem32.astype(np.float64), fdr, river, px)
    res.update(fd_dem=dem32, fd_fdr=fdr, fd_river=river, fd_full=fd_full.astype(np.float32),
               fd_idx_full=idx_full.astype(np.int64))
    k = 0
    for (mS, mE, nS, nE) in ((-1, 20, -1, 25), (20, 48, 25, 60), (10, 30, 15, 45), (-1, 48, 30, 60), (25, 48, -1, 60),
                             (-1, 15, 40, 60), (12, 13 + 12, 5, 5 + 9)):
        # separator lines at rows mS / mE and columns nS / nE (-1 / H / W = raster edge: no neighbour)
        out = np.array([1 if mS >= 0 else 0, 1 if nS >= 0 else 0, 1 if nE < W else 0, 1 if mE < H else 0], float)

        def line(vals, lo, hi, fixed, axis, before, after):
            seg = vals[fixed, lo + 1:hi] if axis == 0 else vals[lo + 1:hi, fixed]
            seg = np.asarray(seg, float)
            if after:
                seg = np.append(seg, vals[fixed, hi] if axis == 0 else vals[hi, fixed])
            if before:
                seg = np.insert(seg, 0, vals[fixed, lo] if axis == 0 else vals[lo, fixed])
            return seg
        vecs_d, vecs_i = [np.zeros(1)] * 4, [np.zeros(1)] * 4
        if out[0]:
            vecs_d[0] = line(fd_full, nS, nE, mS, 0, nS >= 0, nE < W)
            vecs_i[0] = line(idx_full, nS, nE, mS, 0, nS >= 0, nE < W)
        if out[3]:
            vecs_d[3] = line(fd_full, nS, nE, mE, 0, nS >= 0, nE < W)
            vecs_i[3] = line(idx_full, nS, nE, mE, 0, nS >= 0, nE < W)
        if out[1]:
            vecs_d[1] = line(fd_full, mS, mE, nS, 1, mS >= 0, mE < H)
            vecs_i[1] = line(idx_full, mS, mE, nS, 1, mS >= 0, mE < H)
        if out[2]:
            vecs_d[2] = line(fd_full, mS, mE, nE, 1, mS >= 0, mE < H)
            vecs_i[2] = line(idx_full, mS, mE, nE, 1, mS >= 0, mE < H)
        size = max(len(v) for v in vecs_d)
        bound, bound_i = np.zeros((4, size)), np.zeros((4, size))
        for q in range(4):
            bound[q, :len(vecs_d[q])] = vecs_d[q]
            bound_i[q, :len(vecs_i[q])] = vecs_i[q]
        r0, c0 = mS + 1, nS + 1
        f, i = R_flowhand.flow_distance_index_cpu(dem32[r0:mE, c0:nE].astype(np.float64), fdr[r0:mE, c0:nE],
                                                  river[r0:mE, c0:nE], px, bound, bound_i, out, r0, c0, W)
        res.update({"fd_tile%d" % k: np.array([r0, mE, c0, nE]), "fd_out%d" % k: out, "fd_bound%d" % k: bound,
                    "fd_boundi%d" % k: bound_i, "fd_f%d" % k: np.asarray(f, np.float32),
                    "fd_i%d" % k: np.asarray(i, np.float64)})
        k += 1
    res["fd_ntiles"] = k
    # the separator pre-solve itself (flowhand.py:128-239, :283-286): only the cells marked -50 are solved
    marks = np.zeros((H, W), np.float32)
    marks[:, [20, 40]] = -50
    marks[[16, 32], :] = -50
    res["sep_marks"] = marks.copy()
    sf, si = R_flowhand.fdist_indexes_sequential_jit(fdr, river, px, marks)
    res.update(sep_fdist=np.asarray(sf, np.float32), sep_idx=np.asarray(si, np.int32))
    sf, si = R_flowhand.fdist_indexes_sequential_jit(fdr, river, px)
    res.update(sep_all_fdist=np.asarray(sf, np.float32), sep_all_idx=np.asarray(si, np.int32))
    # whole tiled driver (flowhand.py:242-411) with 2 x 1 divisions: what callers with division_* > 0 get
    tf, tidx, thand = R_flowhand.flow_hand_index(dem32.astype(np.float64), fdr, river, px, 1, 2)
    res.update(fh_tiled_fdist=np.asarray(tf, np.float32), fh_tiled_idx=np.asarray(tidx, np.int64),
               fh_tiled_hand=np.asarray(thand, np.float32))
    save("shims", **res)


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def case_example_descriptors():
    """The six float / walk descriptors of the whole bundled Example raster (example.py:59-91), ~8 min:
    sha256 of the bit-exact rasters (slope %, downslope), and for the float descriptors a strided sample, the
    nodata count and min / max / mean."""
    print("example_descriptors", flush=True)
    dem, fdr, fac, flood, _ = load_example()
    river = np.where(fac > 128000, 1, 0).astype("int8")
    g = np.load(os.path.join(GOLD, "example_full.npz"))
    t = time.time()
    sl = R_slope.sloper(dem, 12.5).astype("float32")
    slr = np.where(dem == -100, -100, np.arctan(sl / 100).astype("float32")).astype("float32")
    ti, mti = R_topo.topographic_index(fac, slr.astype(np.float64), 12.5, 0.1)
    print("   slope + TI %.0fs" % (time.time() - t), flush=True)
    down = R_down.downsloper(dem, fdr, 12.5, 5).astype(np.float32)
    print("   downslope %.0fs" % (time.time() - t), flush=True)
    hand, idx = g["hand"], g["idx"].astype(np.int64)
    gf = R_gfi.gfi_calculator(hand, fac, idx, 0.4, 0.1, 12.5)
    ln = R_gfi.ln_hl_H_calculator(hand, fac, 0.4, 0.1, 12.5)
    print("   gfi %.0fs" % (time.time() - t), flush=True)
    res = {"sha_slope": _sha(sl), "sha_down": _sha(down), "slope_rad_sample": slr[::13, ::11]}
    for name, a in (("slope", sl), ("ti", ti), ("mti", mti), ("gfi", gf), ("lnhlh", ln), ("down", down)):
        a32 = np.asarray(a, np.float32)
        v = a32[a32 != -100]
        res[name + "_sample"] = a32[::13, ::11].copy()
        res[name + "_stats"] = np.array([int((a32 == -100).sum()), float(v.min()), float(v.max()),
                                         float(v.astype(np.float64).mean())])
        print("   %-6s nodata %d min %r max %r mean %r sha %s" % (name, (a32 == -100).sum(), v.min(), v.max(),
                                                                  v.astype(np.float64).mean(), _sha(a32)), flush=True)
    save("example_desc", **res)


CASES = {"f64": case_f64, "shims": case_shims, "example_descriptors": case_example_descriptors, "synth": case_synth, "example_windows": case_example_windows, "edge": case_edge,
         "eval": case_eval, "example_full": case_example_full}

if __name__ == "__main__":
    names = sys.argv[1:] or ["edge", "eval", "synth", "example_windows"]
    for n in names:
        CASES[n]()
