"""GPU parity (-m gpu): the HIP path, called through the C ABI (descriptools_amd -> ctypes ->
libdescriptools_hip.so), against (1) the golden vectors generated from the reference's source,
(2) the oracle on seeded synthetic DEMs, (3) the reference's own known-answer Example."""
import numpy as np
import pytest

import oracle
from conftest import assert_float_close, golden, load_example

pytestmark = pytest.mark.gpu

CASES = ["syn_a", "syn_b", "syn_c", "ex_river", "ex_head", "ex_edge"]


@pytest.fixture(scope="module")
def dt():
    import descriptools_amd.slope as slope
    import descriptools_amd.flowdir as flowdir
    import descriptools_amd.flowacc as flowacc
    import descriptools_amd.flowhand as flowhand
    import descriptools_amd.topoindexes as topoindexes
    import descriptools_amd.gfi as gfi
    import descriptools_amd.downslope as downslope
    import descriptools_amd.evaluation as evaluation
    from descriptools_amd import _lib
    assert _lib.lib().dt_device_count() >= 1, "no GPU visible: the HIP path cannot run"

    class NS:
        pass
    ns = NS()
    ns.slope, ns.flowdir, ns.flowacc, ns.flowhand = slope, flowdir, flowacc, flowhand
    ns.topoindexes, ns.gfi, ns.downslope, ns.evaluation = topoindexes, gfi, downslope, evaluation
    return ns


@pytest.mark.parametrize("name", CASES)
def test_golden_slope(dt, name):
    g = golden(name)
    sl = dt.slope.sloper(g["dem"], float(g["px"]))
    assert sl.dtype == np.float64
    assert np.array_equal(sl.astype(np.float32), g["slope"]), "slope: bit-exact"


@pytest.mark.parametrize("name", CASES)
def test_golden_twi(dt, name):
    g = golden(name)
    ti, mti = dt.topoindexes.topographic_index(g["fac"], g["slope_rad"], float(g["px"]), float(g["n_top"]))
    assert_float_close(ti, g["ti"], rtol=1e-5, what="ti")
    assert_float_close(mti, g["mti"], rtol=1e-5, atol=1e-6, what="mti")


@pytest.mark.parametrize("name", CASES)
def test_golden_flowhand(dt, name):
    g = golden(name)
    fd, idx, hand = dt.flowhand.flow_hand_index(g["dem"], g["fdr"], g["river"], float(g["px"]))
    assert idx.dtype == np.int64 and fd.dtype == np.float32 and hand.dtype == g["dem"].dtype
    assert np.array_equal(idx, g["idx"]), "river index: bit-exact"
    assert np.array_equal(hand, g["hand"].astype(hand.dtype)), "HAND: bit-exact"
    assert_float_close(fd, g["fdist"], rtol=1e-6, what="flow distance")
    assert np.array_equal(dt.flowhand.hand_calculator(g["dem"], g["idx"]), g["hand"].astype(hand.dtype))


@pytest.mark.parametrize("name", CASES)
def test_golden_gfi(dt, name):
    g = golden(name)
    out = dt.gfi.gfi_calculator(g["hand"], g["fac"], g["idx"], float(g["n_gfi"]), float(g["b"]), float(g["px"]))
    assert_float_close(out, g["gfi"], rtol=1e-5, atol=1e-6, what="gfi")
    out = dt.gfi.ln_hl_H_calculator(g["hand"], g["fac"], float(g["n_gfi"]), float(g["b"]), float(g["px"]))
    assert_float_close(out, g["lnhlh"], rtol=1e-5, atol=1e-6, what="lnhlh")
    ra = dt.gfi.river_accumulation(g["fac"], g["idx"])
    assert np.array_equal(ra, np.where(g["idx"] != -100, g["fac"].reshape(-1)[g["idx"]], g["fac"].reshape(-1)[0]))


@pytest.mark.parametrize("name", CASES)
def test_golden_downslope(dt, name):
    g = golden(name)
    out = dt.downslope.downsloper(g["dem"], g["fdr"], float(g["px"]), float(g["dz"]))
    ref = g["down"]
    pit = np.isnan(ref)  # 0/0 in the reference: undefined; the build returns 0
    assert np.array_equal(out, np.where(pit, 0, ref)), "downslope: bit-exact"
    raw = dt.downslope.downslope_cpu(g["dem"], g["fdr"], float(g["px"]), float(g["dz"]))
    ok = raw != -50
    assert np.array_equal(raw[ok].astype(np.float32), out[ok])
    fixed = dt.downslope.downslope_sequential_jit(g["dem"], g["fdr"], float(g["px"]), float(g["dz"]),
                                                  raw.astype(np.float32))
    assert np.array_equal(fixed, out)


def test_golden_edge_cases(dt):
    g = golden("edge")
    fd, idx, hand = dt.flowhand.flow_hand_index(g["fh_dem"], g["fh_fdr"], g["fh_river"], 10.0)
    assert np.array_equal(idx, g["fh_idx"])
    assert np.array_equal(hand, g["fh_hand"])
    assert_float_close(fd, g["fh_fdist"], rtol=1e-6)
    # 20000-move cap (flowhand.py:834-837)
    W = int(g["cap_W"])
    fdr = np.ones((1, W), np.uint8)
    river = np.zeros((1, W), np.int8)
    river[0, W - 1] = 1
    fd, idx, _ = dt.flowhand.flow_hand_index(np.full((1, W), 5, np.int16), fdr, river, 10.0)
    sel = g["cap_sel"]
    assert np.array_equal(idx[0, sel], g["cap_idx"])
    assert_float_close(fd[0, sel], g["cap_fdist"], rtol=1e-6)
    assert idx[0, W - 1 - 20000] == W - 1 and idx[0, W - 1 - 20001] == -100
    # long diagonal chain: sqrt(2) accumulation
    n = int(g["diag_n"])
    fdr = np.full((n, n), 2, np.uint8)
    river = np.zeros((n, n), np.int8)
    river[n - 1, n - 1] = 1
    fd, idx, _ = dt.flowhand.flow_hand_index(np.full((n, n), 5, np.int16), fdr, river, 12.5)
    sel = g["diag_sel"]
    assert np.array_equal(idx.reshape(-1)[sel], g["diag_idx"])
    assert_float_close(fd.reshape(-1)[sel], g["diag_fdist"], rtol=1e-6)
    # downslope micro cases + 5000-iteration cap
    out = dt.downslope.downsloper(g["ds_dem"], g["ds_fdr"], 10.0, 5)
    assert np.array_equal(out, g["ds_out"])
    out = dt.downslope.downsloper(g["dcap_dem"], np.ones(g["dcap_dem"].shape, np.uint8), 10.0, 5)
    assert np.array_equal(out[0, g["dcap_sel"]], g["dcap_out"])
    # pointwise special values
    ti, mti = dt.topoindexes.topographic_index(g["pw_fac"], g["pw_slr"], 12.5, 0.1)
    assert_float_close(ti, g["pw_ti"], rtol=1e-5)
    assert_float_close(mti, g["pw_mti"], rtol=1e-5)
    assert_float_close(dt.gfi.gfi_calculator(g["pw_hand"], g["pw_fac"], g["pw_idx"], 0.4, 0.1, 12.5),
                       g["pw_gfi"], rtol=1e-5)
    assert_float_close(dt.gfi.ln_hl_H_calculator(g["pw_hand"], g["pw_fac"], 0.4, 0.1, 12.5),
                       g["pw_lnhlh"], rtol=1e-5)


def test_golden_eval(dt):
    g = golden("eval")
    for k in range(3):
        under = str(g["e%d_under" % k])
        flood = g["e%d_flood" % k].copy()
        desc = dt.evaluation.minMaxScale(g["e%d_hand" % k], g["e%d_mn" % k], g["e%d_mx" % k], -100)
        assert np.array_equal(desc, g["e%d_desc" % k], equal_nan=True)
        th = dt.evaluation.calibration(desc, flood, under)
        assert th == float(g["e%d_th" % k]), "calibrated threshold: exact"
        assert np.array_equal(flood, g["e%d_flood_after" % k]), "benchmark map remapped in place"
        binary = dt.evaluation.binary_map(desc, th, under)
        assert np.array_equal(binary, g["e%d_binary" % k])
        c, f, cm = dt.evaluation.avaliacao(binary, flood)
        assert c == float(g["e%d_c" % k]) and f == float(g["e%d_f" % k])
        assert np.array_equal(cm, g["e%d_class" % k])
        # numpy's dtypes for these expressions: an int16 raster scales in float64, np.where(..., 1, 0) is int64
        assert desc.dtype == np.float64 and binary.dtype == np.int64 and cm.dtype == np.int64
        d32 = dt.evaluation.minMaxScale(g["e%d_hand" % k].astype(np.float32), np.float32(g["e%d_mn" % k]),
                                        np.float32(g["e%d_mx" % k]), -100)
        h32 = g["e%d_hand" % k].astype(np.float32)
        want = np.where(h32 == -100, np.nan, (h32 - np.float32(g["e%d_mn" % k])) /
                        (np.float32(g["e%d_mx" % k]) - np.float32(g["e%d_mn" % k]))).astype(np.float32)
        assert d32.dtype == np.float32 and np.array_equal(d32, want, equal_nan=True), "float32 rasters scale in float32"
        # avaliacao on a fresh copy of the benchmark map remaps that copy in place (evaluation.py:149-150)
        fresh = g["e%d_flood" % k].copy()
        c2, f2, cm2 = dt.evaluation.avaliacao(binary, fresh)
        assert (c2, f2) == (c, f) and np.array_equal(cm2, cm) and np.array_equal(fresh, g["e%d_flood_after" % k])


@pytest.mark.parametrize("seed,H,W,nod", [(1, 257, 300, 0), (2, 512, 512, 4), (3, 1000, 1536, 0),
                                           (4, 33, 1027, 3), (5, 1, 50, 0), (6, 70, 1, 0)])
def test_oracle_synthetic_chain(dt, seed, H, W, nod):
    """Whole chain vs the oracle on seeded synthetic DEMs (ragged shapes, nodata blobs)."""
    px = 10.0
    dem = oracle.synth_dem(seed, 2048, 2048, 100, 50, H, W, nod)
    # device generator is bit-identical to the oracle's
    from descriptools_amd import _lib
    dev_dem = np.empty((H, W), np.float32)
    _lib.check(_lib.lib().dt_synth_dem(seed, 2048, 2048, 100, 50, H, W, nod, _lib.ptr(dev_dem, _lib.c_f32p)))
    assert np.array_equal(dev_dem, dem)
    sl_o, fdr_o = oracle.slope_d8(dem, px)
    fdr, sl = dt.flowdir.d8(dem, px, return_slope=True)
    assert np.array_equal(fdr, fdr_o), "D8: bit-exact"
    assert np.array_equal(sl, sl_o), "slope: bit-exact"
    acc_o = oracle.flowacc(fdr_o, dem)
    acc = dt.flowacc.accumulate(fdr, dem)
    assert np.array_equal(acc, acc_o), "flow accumulation: bit-exact"
    river = (acc > max(8, H * W // 512)).astype(np.int8)
    fd_o, idx_o, hand_o = oracle.flowhand(dem, fdr, river, px)
    fd, idx, hand = dt.flowhand.flow_hand_index(dem, fdr, river, px)
    assert np.array_equal(idx, idx_o) and np.array_equal(hand, hand_o)
    assert_float_close(fd, fd_o, rtol=1e-6, what="flow distance")
    slr = np.where(dem == -100, -100, np.arctan(sl / 100)).astype(np.float32)
    ti_o, mti_o = oracle.twi(acc, slr, px, 0.1)
    ti, mti = dt.topoindexes.topographic_index_cpu(acc, slr, px, 0.1)
    assert_float_close(ti, ti_o, rtol=1e-5, what="ti")
    assert_float_close(mti, mti_o, rtol=1e-5, atol=1e-6, what="mti")
    assert_float_close(dt.gfi.gfi_calculator(hand, acc, idx, 0.4, 0.1, px), oracle.gfi(hand, acc, idx, 0.4, 0.1, px),
                       rtol=1e-5, atol=1e-6, what="gfi")
    assert_float_close(dt.gfi.ln_hl_H_calculator(hand, acc, 0.4, 0.1, px), oracle.lnhlh(hand, acc, 0.4, 0.1, px),
                       rtol=1e-5, atol=1e-6, what="lnhlh")
    assert np.array_equal(dt.downslope.downsloper(dem, fdr, px, 5), oracle.downslope(dem, fdr, px, 5.0))


def test_empty_rasters(dt):
    assert dt.slope.sloper(np.zeros((0, 5), np.float32), 10.0).shape == (0, 5)
    fd, idx, hand = dt.flowhand.flow_hand_index(np.zeros((0, 0), np.int16), np.zeros((0, 0), np.uint8),
                                                np.zeros((0, 0), np.int8), 10.0)
    assert fd.shape == (0, 0)


def test_example_known_answer(dt):
    """The reference's only KAT (Example/example.py:82-147): HAND -> minMaxScale -> calibration ->
    binary_map -> avaliacao must reproduce Example/output/hand_class.tif exactly."""
    dem, fdr, fac, river, flood, klass = load_example()
    flow, idx, hand = dt.flowhand.flow_hand_index(dem, fdr, river, 12.5)
    g = golden("example_full")
    assert np.array_equal(idx, g["idx"].astype(np.int64))
    assert np.array_equal(hand, g["hand"])
    assert_float_close(flow, g["fdist"], rtol=1e-6, what="flow distance")
    el = np.unique(hand)
    mx, mn = el[-1], el[1]
    assert (mn, mx) == (0, 259)
    desc = dt.evaluation.minMaxScale(hand, mn, mx, -100)
    th = dt.evaluation.calibration(desc, flood, 'under')
    assert th == 0.012
    binary = dt.evaluation.binary_map(desc, th, 'under')
    c, f, cm = dt.evaluation.avaliacao(binary, flood)
    assert c == 0.8581615676712259 and f == 0.7240945135019289
    assert np.array_equal(cm.astype(np.uint8), klass), "class map == Example/output/hand_class.tif"


def _random_fdr(rng, H, W, mode):
    codes = np.array([1, 2, 4, 8, 16, 32, 64, 128], np.uint8)
    if mode == "random":      # arbitrary field: cycles, dead ends, exits everywhere
        fdr = codes[rng.integers(0, 8, size=(H, W))]
    elif mode == "south":     # long parallel chains crossing many tiles
        fdr = np.full((H, W), 4, np.uint8)
        fdr[rng.random((H, W)) < 0.2] = 2
        fdr[rng.random((H, W)) < 0.2] = 8
    else:                     # east-flowing with junk codes and holes
        fdr = np.full((H, W), 1, np.uint8)
        fdr[rng.random((H, W)) < 0.15] = 128
        fdr[rng.random((H, W)) < 0.15] = 2
        fdr[rng.random((H, W)) < 0.02] = 0
        fdr[rng.random((H, W)) < 0.01] = 3
    return fdr


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("H,W,mode", [(64, 64, "random"), (65, 129, "random"), (200, 333, "south"),
                                      (130, 700, "east"), (1, 300, "east"), (300, 1, "south"),
                                      (257, 256, "random"), (1024, 1024, "south")])
def test_flowacc_arbitrary_fields(dt, impl, H, W, mode):
    """Flow accumulation on arbitrary direction fields (cycles inside and across tiles, non-D8
    codes, ragged shapes) equals the oracle bit for bit, for both implementations."""
    from descriptools_amd import _lib
    rng = np.random.default_rng(H * 1000 + W)
    fdr = _random_fdr(rng, H, W, mode)
    _lib.check(_lib.lib().dt_set_flow_impl(impl))
    try:
        acc = dt.flowacc.accumulate(fdr)
    finally:
        _lib.check(_lib.lib().dt_set_flow_impl(2))
    ref = oracle.flowacc(fdr)
    assert np.array_equal(acc, ref), "%d cells differ" % int((acc != ref).sum())


def test_full_size_4096_vs_oracle(dt):
    """BASELINE.json configs[1]: 4096 x 4096 synthetic DEM, every descriptor of the resident chain
    against the oracle (O(N) oracle variants; ~20 s of CPU)."""
    from descriptools_amd import chain
    n, px = 4096, 10.0
    dem = oracle.synth_dem(1, n, n)
    thr = n * n // 512
    out = chain.run_host(dem, px, river_threshold=thr)
    sl_o, fdr_o = oracle.slope_d8(dem, px)
    assert np.array_equal(out["fdr"], fdr_o) and np.array_equal(out["slope"], sl_o)
    acc_o = oracle.flowacc(fdr_o, dem)
    assert np.array_equal(out["fac"], acc_o)
    river = (acc_o > thr).astype(np.int8)
    assert np.array_equal(out["river"], river)
    idx_o, nc, nd = oracle.flowhand_fast(fdr_o, river)
    assert np.array_equal(out["idx"], idx_o), "river index: bit-exact at full size"
    ok = idx_o != -100
    d_o = np.where(ok, (px * nc + (px * np.sqrt(2.0)) * nd), -100.0).astype(np.float32)
    assert np.array_equal(out["fdist"], d_o), "count-form distance: identical arithmetic"
    flat = dem.reshape(-1)
    hand_o = np.where(ok, np.maximum(dem - flat[np.where(ok, idx_o, 0)], 0), -100).astype(np.float32)
    assert np.array_equal(out["hand"], hand_o)
    assert np.array_equal(out["down"], oracle.downslope(dem, fdr_o, px, 5.0))
    slr = np.where(dem == -100, -100, np.arctan(sl_o / 100)).astype(np.float32)
    # the chain's radians come from its own float64 atan: <= 1 float32 ulp from numpy's float32 arctan
    assert np.max(np.abs(out["slope_rad"].astype(np.float64) - slr)) <= 2.4e-7
    ti_o, mti_o = oracle.twi(acc_o, out["slope_rad"], px, 0.1)
    assert_float_close(out["ti"], ti_o, rtol=1e-5, what="ti")
    assert_float_close(out["mti"], mti_o, rtol=1e-5, atol=1e-6, what="mti")
    g_o = oracle.gfi(hand_o, acc_o, idx_o, 0.4, 0.1, px)
    assert_float_close(out["gfi"], g_o, rtol=1e-5, atol=1e-6, what="gfi")
    l_o = oracle.lnhlh(hand_o, acc_o, 0.4, 0.1, px)
    assert_float_close(out["lnhlh"], l_o, rtol=1e-5, atol=1e-6, what="lnhlh")
    # how far inside the tolerance the float descriptors are (reported, not asserted tightly)
    v = ti_o != -100
    print("max rel err  ti %.2e  mti-abs %.2e  gfi-abs %.2e" % (
        np.max(np.abs(out["ti"][v] - ti_o[v]) / np.abs(ti_o[v])), np.max(np.abs(out["mti"][v] - mti_o[v])),
        np.max(np.abs(out["gfi"][g_o != -100] - g_o[g_o != -100]))))


def test_full_size_16384_properties(dt):
    """BASELINE.json configs[2] size: size-independent properties of the 16384^2 chain on the GPU
    (the oracle would need minutes): conservation of flow accumulation, river / HAND consistency, and
    equality with a 2 x 2 tiling of the same DEM run as four logical ranks."""
    import torch
    from descriptools_amd import _lib, chain, tiling
    from descriptools_amd.device import Context
    n, px = 16384, 10.0
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
    # (1) every cell drains to exactly one outlet: sum over outlets of (acc + 1) == number of cells
    #     (synthetic DEM: no nodata, no cycles); outlets = cells whose D8 step leaves the raster
    yy = torch.arange(n, device="cuda").view(-1, 1).expand(n, n)
    xx = torch.arange(n, device="cuda").view(1, -1).expand(n, n)
    dy = torch.zeros(256, dtype=torch.int64, device="cuda")
    dx = torch.zeros(256, dtype=torch.int64, device="cuda")
    for c, (a, b) in {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0),
                      128: (-1, 1)}.items():
        dy[c], dx[c] = a, b
    f = fdr.long()
    ty, tx = yy + dy[f], xx + dx[f]
    outlet = (ty < 0) | (ty >= n) | (tx < 0) | (tx >= n) | (f == 0)
    assert int((fac[outlet].long() + 1).sum()) == n * n
    assert int((fac < 0).sum()) == 0
    # (2) river mask, and HAND consistency: river cells drain to themselves at distance 0
    assert torch.equal(river, (fac > thr).to(torch.int8))
    lin = (yy * n + xx).to(torch.int32)
    rv = river == 1
    assert torch.equal(idx[rv], lin[rv]) and float(fdist[rv].abs().max()) == 0.0
    ok = idx >= 0
    assert bool((river.view(-1)[idx[ok].long()] == 1).all()), "every river index points at a river cell"
    assert bool((hand[ok] >= 0).all()) and bool((hand[~ok] == -100).all())
    assert bool((fdist[ok & ~rv] >= px).all())
    # (3) the same DEM as 2 x 2 logical ranks == untiled
    layout = tiling.Layout([n // 2, n // 2], [n // 2, n // 2])
    tiles = []
    for r in range(4):
        tl = tiling.RankTile(layout, r, device=0, px=px, river_threshold=thr)
        tl.synth_dem(1)
        tiles.append(tl)
    tiling.simulate(tiles, layout)
    for tl in tiles:
        assert tl.unresolved_downslope() == 0
        y0, x0 = layout.origin(tl.rank)
        for name in ("fdr", "fac", "river", "fdist", "hand", "a_river", "slope", "ti", "mti", "gfi", "lnhlh",
                     "down"):
            a, b = tl.core(name), t[name][y0:y0 + tl.H, x0:x0 + tl.W]
            assert torch.equal(a, b), (tl.rank, name, int((a != b).sum()), torch.nonzero(a != b)[:4].tolist())
        gi = tl.core("idx")
        li = t["idx"][y0:y0 + tl.H, x0:x0 + tl.W].long()
        assert torch.equal(gi, li), (tl.rank, "idx")


def test_evaluate_resident_matches_host_api(dt):
    """Device-resident evaluation (extremes -> minMaxScale -> calibration -> counts) == the drop-in host
    functions on the same float32 HAND raster, on the Example data and on a synthetic chain."""
    from descriptools_amd import chain, evaluation
    from descriptools_amd.device import Context
    dem, fdr, fac, river, flood, klass = load_example()
    _, _, hand = dt.flowhand.flow_hand_index(dem, fdr, river, 12.5)
    hand32 = hand.astype(np.float32)
    ctx = Context()
    d_hand, d_flood = ctx.to_device(hand32), ctx.to_device(flood.astype(np.int8))
    res = evaluation.evaluate_resident(ctx, d_hand.ptr, d_flood.ptr, hand32.size, 'under')
    d_hand.free(); d_flood.free(); ctx.close()
    # host reference on the SAME float32 raster (numpy keeps float32 in minMaxScale / binary_map)
    el = np.unique(hand32)
    desc = evaluation.minMaxScale(hand32, el[1], el[-1], -100)
    fl = flood.astype(np.int8).copy()
    th = evaluation.calibration(desc, fl, 'under')
    c, f, cm = evaluation.avaliacao(evaluation.binary_map(desc, th, 'under'), fl)
    assert (res["mn"], res["mx"]) == (float(el[1]), float(el[-1])) == (0.0, 259.0)
    assert res["threshold"] == th == 0.012
    assert np.array_equal(res["counts"], np.bincount(cm.reshape(-1).astype(np.int64), minlength=4))
    assert res["correctness"] == c and res["fit"] == f
    # the int16 pipeline of the example gives the same class counts (exact small integers in float32)
    assert np.array_equal(res["counts"], np.bincount(klass.reshape(-1).astype(np.int64), minlength=4))


@pytest.mark.parametrize("seed", [1, 2])
def test_downslope_arbitrary_direction_field(dt, seed):
    """random D8 codes (cycles, spirals, non-D8 codes, moves off the raster), rough and flat terrain, nodata:
    every exit of the windowed walk (drop reached, stop flags, window ring, the 256-move limit, stepping
    onto nodata, the unsafe-rounding recheck) against the oracle's literal walk."""
    rng = np.random.default_rng(seed)
    H, W = 333, 417
    codes = np.array([1, 2, 4, 8, 16, 32, 64, 128, 0, 3, 255], np.uint8)
    fdr = codes[rng.integers(0, len(codes), size=(H, W))]
    fdr[rng.random((H, W)) < 0.5] = 4
    dem = (rng.random((H, W)) * 40).astype(np.float32)
    dem[:, 200:] = np.float32(7.25)              # flat half: walks run until a cycle, the edge or the cap
    dem[rng.random((H, W)) < 0.02] = -100
    for dz in (5.0, 0.3):
        want = oracle.downslope(dem, fdr, 10.0, dz)
        got = dt.downslope.downsloper(dem, fdr, 10.0, dz)
        assert np.array_equal(got, want, equal_nan=True), int((got != want).sum())


def test_chain_overlap_branch_gives_the_same_rasters(dt):
    """Chain(overlap=True) runs downslope on a second stream (dt_ctx_fork / dt_ctx_join): same results."""
    from descriptools_amd import chain
    dem = oracle.synth_dem(3, 700, 900, 0, 0, 700, 900, 2)
    a = chain.run_host(dem, 10.0)
    b = chain.run_host(dem, 10.0, overlap=True)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_flowhand_long_in_tile_paths(dt):
    """boustrophedon paths that stay inside 64 x 64 tiles for thousands of moves: the in-tile move counts
    exceed 8 bits (the wide form of the HAND word cache) and the doubling needs all its rounds."""
    H, W = 128, 192
    fdr = np.zeros((H, W), np.uint8)
    for y in range(H):
        fdr[y, :] = 1 if y % 2 == 0 else 16          # east on even rows, west on odd rows
        fdr[y, W - 1 if y % 2 == 0 else 0] = 4        # turn south at the end of the row
    fdr[H - 1, 0 if (H - 1) % 2 else W - 1] = 4       # last cell flows off the raster
    # a second family: serpentines confined to 64-wide columns
    for x0 in (0, 64):
        for y in range(64):
            fdr[y, x0:x0 + 64] = 1 if y % 2 == 0 else 16
            fdr[y, x0 + 63 if y % 2 == 0 else x0] = 4
    rng = np.random.default_rng(5)
    dem = (rng.random((H, W)) * 50 + 10).astype(np.float32)
    river = np.zeros((H, W), np.int8)
    river[63, 0] = river[63, 64] = 1
    river[H - 1, W // 2] = 1
    fd_o, idx_o, hand_o = oracle.flowhand(dem, fdr, river, 10.0)
    fd, idx, hand = dt.flowhand.flow_hand_index(dem, fdr, river, 10.0)
    assert np.array_equal(idx, idx_o) and np.array_equal(hand, hand_o)
    assert (idx_o >= 0).sum() > 8000 and fd_o.max() > 40000
    assert_float_close(fd, fd_o, rtol=1e-6, what="flow distance")


def test_slope_twi_cold_path_equals_hot_path(dt):
    """The fused slope + TI + MTI stencil is a branch-free fast kernel plus an exact fix-up kernel for the cells
    it flags (dt_stencil.hip).  With every cell flagged (test knob) the exact path writes all of them: slope and
    radians must not change at all, TI / MTI only inside the fast path's error bound, and both agree with the
    oracle; ragged shapes, nodata blobs, negative / zero / huge accumulations."""
    from descriptools_amd import _lib, chain
    L = _lib.lib()
    rng = np.random.default_rng(7)
    for (H, W, nod) in ((300, 777, 3), (64, 1024, 0), (17, 5, 0)):
        dem = oracle.synth_dem(5, 2048, 2048, 10, 20, H, W, nod)
        dem[rng.random((H, W)) < 0.01] = -250.0          # below the nodata sentinel: slope -100, finite radians
        outs = []
        for flag_all in (0, 1):
            _lib.check(L.dt_debug_set(0, flag_all))
            try:
                outs.append(chain.run_host(dem, 10.0, river_threshold=max(8, H * W // 512)))
            finally:
                _lib.check(L.dt_debug_set(0, 0))
        a, b = outs
        assert np.array_equal(a["slope"], b["slope"]) and np.array_equal(a["slope_rad"], b["slope_rad"])
        sl_o, _ = oracle.slope_d8(dem, 10.0)
        assert np.array_equal(a["slope"], sl_o)
        for k in ("ti", "mti"):
            assert_float_close(a[k], b[k], rtol=4e-6, atol=1e-6, what=k + " hot vs cold")
        ti_o, mti_o = oracle.twi(a["fac"], a["slope_rad"], 10.0, 0.1)
        for got in (a, b):
            assert_float_close(got["ti"], ti_o, rtol=1e-5, what="ti")
            assert_float_close(got["mti"], mti_o, rtol=1e-5, atol=1e-6, what="mti")


def test_d8_nodata_mask_and_the_flow_pass_that_reads_it():
    """round 4: the D8 kernel writes one nodata bit per cell on its way (dt_dev_slope_d8_m; a 16-bit word per 4 x 4
    patch) and the fused accumulation /
    HAND pass reads that instead of the DEM (dt_dev_flowacc_river_flowhand_local_m).  The mask must be exactly
    `dem <= -100`, row by row, for ragged widths too, and the rasters must equal the DEM-reading entry point's."""
    import ctypes as C
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    L = _lib.lib()
    ctx = Context()
    for H, W, nod in ((192, 256, 4), (130, 201, 3), (64, 64, 0), (257, 515, 6)):
        dem = oracle.synth_dem(21, 2048, 2048, 100, 300, H, W, nod)
        dem[5, 7] = -250.0                       # below the sentinel: nodata as well (dem <= -100)
        d = ctx.to_device(dem)
        fdr = ctx.empty((H, W), np.uint8)
        nb = int(L.dt_nodata_mask_bytes(H, W))
        ldw = int(L.dt_nodata_mask_bytes(4, W)) // 2          # 16-bit words per row of 4 x 4 patches
        assert nb == ((H + 3) // 4) * ldw * 2 and ldw >= (W + 3) // 4
        mask = ctx.empty((nb,), np.uint8)
        _lib.check(L.dt_dev_slope_d8_m(ctx.h, d.ptr, H, W, 10.0, fdr.ptr, mask.ptr))
        ctx.sync()
        m = mask.to_host().view(np.uint16).reshape((H + 3) // 4, ldw)
        bits = np.zeros((4 * m.shape[0], 4 * ldw), bool)
        for j in range(4):
            for k in range(4):
                bits[j::4, k::4] = (m >> (4 * j + k)) & 1
        assert np.array_equal(bits[:H, :W], dem <= -100), (H, W)
        assert np.array_equal(fdr.to_host(), oracle.slope_d8(dem, 10.0)[1])
        outs = []
        for use_mask in (False, True):
            fac, river = ctx.empty((H, W), np.int32), ctx.empty((H, W), np.int8)
            if use_mask:
                _lib.check(L.dt_dev_flowacc_river_flowhand_local_m(ctx.h, fdr.ptr, d.ptr, mask.ptr, H, W, 50, fac.ptr, river.ptr))
            else:
                _lib.check(L.dt_dev_flowacc_river_flowhand_local(ctx.h, fdr.ptr, d.ptr, H, W, 50, fac.ptr, river.ptr))
            ctx.sync()
            outs.append((fac.to_host(), river.to_host()))
            fac.free()
            river.free()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[1][0], oracle.flowacc(oracle.slope_d8(dem, 10.0)[1], dem))
        for b in (d, fdr, mask):
            b.free()
    ctx.close()
