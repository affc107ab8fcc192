"""The DEM dtype contract.  The reference takes height differences in the DEM's own dtype (slope.py:244-258 under
Numba typing, flowhand.py:436-438, downslope.py:468); the tuned kernels take them in float32.  Heights that are float32
values -- in whatever container -- give the reference's arithmetic exactly on that path.  A raster float32 cannot hold
(a genuinely float64 DEM, integer heights beyond 2^24) is computed in float64 by the drop-in descriptor functions
(csrc/dt_wide.hip; round 4 -- rounds 2-3 refused it), and still refused by the float32-only entry points (resident
chain, net-new D8).  The fixture tests/golden/f64.npz is the reference's own run on such a DEM (oracle/gen_golden.py
f64); the oracle's float64 restatements are pinned by it in tests/test_oracle_golden.py."""
import numpy as np
import pytest

from conftest import golden


def test_heights_picks_the_tier():
    from descriptools_amd import _lib
    g = golden("f64")
    dem64 = g["dem"]
    assert dem64.dtype == np.float64 and (dem64.astype(np.float32).astype(np.float64) != dem64).any()
    d, wide = _lib.heights(dem64)
    assert wide and d.dtype == np.float64 and np.array_equal(d, dem64)
    d, wide = _lib.heights(np.full((4, 4), 2 ** 24 + 1, np.int32))
    assert wide and d.dtype == np.float64 and d[0, 0] == 2 ** 24 + 1
    with pytest.raises(ValueError, match="float64 cannot represent"):
        _lib.heights(np.full((4, 4), 2 ** 53 + 1, np.int64))
    # float32-only entry points still refuse (before anything reaches the GPU)
    with pytest.raises(ValueError, match="not exactly representable in float32"):
        _lib.dem_f32(dem64)
    from descriptools_amd import chain, flowdir
    for call in (lambda: chain.run_host(dem64, 10.0), lambda: flowdir.d8(dem64, 10.0)):
        with pytest.raises(ValueError, match="not exactly representable in float32"):
            call()


def test_wide_containers_of_float32_values_pass():
    from descriptools_amd import _lib
    rng = np.random.default_rng(0)
    a32 = (rng.random((50, 60)) * 3000).astype(np.float32)
    a32[3, 4] = -100
    for a in (a32.astype(np.float64), np.round(a32).astype(np.int32), np.round(a32).astype(np.int64),
              np.where(a32 > 100, np.nan, a32).astype(np.float64)):
        d = _lib.dem_f32(a)
        assert d.dtype == np.float32 and np.array_equal(d.astype(a.dtype), a, equal_nan=True)
        d2, wide = _lib.heights(a)
        assert not wide and d2.dtype == np.float32


def test_oracle_float64_restatement_vs_reference_fixture():
    """the checker of the float64 path is itself pinned by the reference's run (CPU)"""
    import oracle
    g = golden("f64")
    dem, px = g["dem"], float(g["px"])
    assert np.array_equal(oracle.slope_f64(dem, px), g["slope"])
    assert np.array_equal(oracle.hand_f64(dem, g["idx"]), g["hand"])
    ref = np.where(np.isnan(g["down"]), 0, g["down"])
    assert np.array_equal(oracle.downslope_f64(dem, g["fdr"], px, 5.0), ref)
    assert np.allclose(oracle.gfi_f64h(g["hand"], g["fac"], g["idx"], 0.4, 0.1, px), g["gfi"], rtol=1e-6, atol=0)
    assert np.allclose(oracle.lnhlh_f64h(g["hand"], g["fac"], 0.4, 0.1, px), g["lnhlh"], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_float64_container_of_float32_values_takes_the_float32_path():
    from descriptools_amd import downslope, flowhand, slope
    s = golden("syn_b")
    px = float(s["px"])
    d64 = s["dem"].astype(np.float64)
    assert np.array_equal(slope.sloper(d64, px).astype(np.float32), s["slope"])
    fd, idx, hand = flowhand.flow_hand_index(d64, s["fdr"], s["river"], px)
    assert hand.dtype == np.float64 and np.array_equal(hand.astype(np.float32), s["hand"]) and np.array_equal(idx, s["idx"])
    ref = np.where(np.isnan(s["down"]), 0, s["down"])  # 0 / 0 at the reference's pits: the build returns 0
    assert np.array_equal(downslope.downsloper(d64, s["fdr"], px, 5), ref)


@pytest.mark.gpu
def test_genuinely_float64_dem_equals_the_reference(monkeypatch):
    """tests/golden/f64.npz, no opt-in: slope, HAND and downslope bit for bit, GFI / ln(hl/H) within 1e-5"""
    monkeypatch.delenv("DT_ALLOW_DEM_ROUNDING", raising=False)
    from descriptools_amd import downslope, flowhand, gfi, slope
    g = golden("f64")
    dem64, px = g["dem"], float(g["px"])
    sl = slope.sloper(dem64, px)
    assert sl.dtype == np.float64 and np.array_equal(sl.astype(np.float32), g["slope"])
    assert np.array_equal(slope.slope_cpu(dem64, px, [1, 1, 1, 1]), g["slope"])
    fd, idx, hand = flowhand.flow_hand_index(dem64, g["fdr"], g["river"], px)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(fd, g["fdist"])
    assert hand.dtype == np.float64 and np.array_equal(hand, g["hand"])
    assert np.array_equal(flowhand.hand_calculator(dem64, g["idx"]), g["hand"])
    ref = np.where(np.isnan(g["down"]), 0, g["down"])
    assert np.array_equal(downslope.downsloper(dem64, g["fdr"], px, 5), ref)
    gf = gfi.gfi_calculator(hand, g["fac"], idx, 0.4, 0.1, px)
    ln = gfi.ln_hl_H_calculator(hand, g["fac"], 0.4, 0.1, px)
    assert np.array_equal(gf == -100, g["gfi"] == -100) and np.allclose(gf, g["gfi"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(ln == -100, g["lnhlh"] == -100) and np.allclose(ln, g["lnhlh"], rtol=1e-5, atol=1e-6)
    # the explicit-area shim on the same HAND
    ar = gfi.river_accumulation(g["fac"], idx)
    assert np.allclose(gfi.geomorphic_flood_index_cpu(hand, ar, 0.4, 0.1, px), g["gfi"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_wide_dems_against_the_float64_oracle(monkeypatch):
    """larger rasters than the fixture: a float64 DEM with sub-float32 structure and nodata, and an int32 DEM beyond
    2^24 (millimetres), through the drop-in functions against the oracle's float64 restatement"""
    monkeypatch.delenv("DT_ALLOW_DEM_ROUNDING", raising=False)
    import oracle
    from descriptools_amd import downslope, flowhand, gfi, slope
    H, W, px = 300, 420, 10.0
    d32 = oracle.synth_dem(8, 1024, 1024, 100, 200, H, W, 3)
    yy, xx = np.mgrid[0:H, 0:W]
    d64 = np.where(d32 == -100, -100.0, d32.astype(np.float64) + 1e-3 * np.sin(0.3 * yy + 0.2 * xx) + 1e-7 * xx)
    mm = np.where(d32 == -100, -100, np.round(d32.astype(np.float64) * 1000.0) + 2 ** 25).astype(np.int32)
    _, fdr = oracle.slope_d8(d32, px)
    fac = oracle.flowacc(fdr, d32)
    river = (fac > 40).astype(np.int8)
    for dem, dz in ((d64, 5.0), (mm, 5000.0)):
        dd = dem.astype(np.float64)
        assert np.array_equal(slope.sloper(dem, px).astype(np.float32), oracle.slope_f64(dd, px))
        fd, idx, hand = flowhand.flow_hand_index(dem, fdr, river, px)
        fd_o, idx_o, _ = oracle.flowhand(d32, fdr, river, px)
        assert np.array_equal(idx, idx_o) and np.allclose(fd, fd_o, rtol=1e-6, atol=0)
        assert hand.dtype == dem.dtype and np.array_equal(hand.astype(np.float64), oracle.hand_f64(dd, idx_o))
        assert np.array_equal(downslope.downsloper(dem, fdr, px, dz), oracle.downslope_f64(dd, fdr, px, dz))
        gf = gfi.gfi_calculator(hand, fac, idx, 0.4, 0.1, px)
        assert np.allclose(gf, oracle.gfi_f64h(hand.astype(np.float64), fac, idx_o, 0.4, 0.1, px), rtol=1e-5, atol=1e-6)
        ln = gfi.ln_hl_H_calculator(hand, fac, 0.4, 0.1, px)
        assert np.allclose(ln, oracle.lnhlh_f64h(hand.astype(np.float64), fac, 0.4, 0.1, px), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_rounding_to_float32_is_still_an_opt_in(monkeypatch):
    """DT_ALLOW_DEM_ROUNDING=1: the fast float32 kernels on the rounded heights; results within the rounding of the
    heights (2^-24 relative per height) of the reference's"""
    from descriptools_amd import flowhand, slope
    g = golden("f64")
    dem64, px = g["dem"], float(g["px"])
    monkeypatch.setenv("DT_ALLOW_DEM_ROUNDING", "1")
    valid = dem64 != -100
    ez = 2.0 ** -24 * float(np.abs(dem64[valid]).max())
    sl = slope.sloper(dem64, px)
    assert np.array_equal(sl == -100, g["slope"] == -100)
    assert np.abs(sl - g["slope"])[valid].max() <= 100.0 * 2 * ez / px * 1.01 + 1e-6 * float(g["slope"].max())
    fd, idx, hand = flowhand.flow_hand_index(dem64, g["fdr"], g["river"], px)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(fd, g["fdist"])
    ok = g["hand"] != -100
    assert np.array_equal(hand == -100, ~ok) and np.abs(hand - g["hand"])[ok].max() <= 2 * ez * 1.01
