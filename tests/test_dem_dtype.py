"""The DEM dtype contract.  The reference takes height differences in the DEM's own dtype
(slope.py:244-258 under Numba typing, flowhand.py:436-438, downslope.py:468); the kernels take them in float32.
Heights that are float32 values -- in whatever container -- give the reference's arithmetic exactly and are accepted;
a raster that float32 cannot hold is refused before anything reaches the GPU (no silent narrowing).  The fixture
tests/golden/f64.npz is the reference's own run on such a genuinely float64 DEM (oracle/gen_golden.py f64)."""
import numpy as np
import pytest

from conftest import golden


def test_inexact_dem_is_refused_before_the_gpu():
    from descriptools_amd import downslope, flowhand, slope
    g = golden("f64")
    dem64, fdr, river, px = g["dem"], g["fdr"], g["river"], float(g["px"])
    assert dem64.dtype == np.float64 and (dem64.astype(np.float32).astype(np.float64) != dem64).any()
    for call in (lambda: slope.sloper(dem64, px), lambda: flowhand.flow_hand_index(dem64, fdr, river, px),
                 lambda: downslope.downsloper(dem64, fdr, px, 5), lambda: slope.slope_sequential_jit(dem64, px)):
        with pytest.raises(ValueError, match="not exactly representable in float32"):
            call()
    with pytest.raises(ValueError, match="int32"):
        slope.sloper(np.full((4, 4), 2 ** 24 + 1, np.int32), px)


def test_wide_containers_of_float32_values_pass():
    from descriptools_amd import _lib
    rng = np.random.default_rng(0)
    a32 = (rng.random((50, 60)) * 3000).astype(np.float32)
    a32[3, 4] = -100
    for a in (a32.astype(np.float64), np.round(a32).astype(np.int32), np.round(a32).astype(np.int64),
              np.where(a32 > 100, np.nan, a32).astype(np.float64)):
        d = _lib.dem_f32(a)
        assert d.dtype == np.float32 and np.array_equal(d.astype(a.dtype), a, equal_nan=True)


@pytest.mark.gpu
def test_float64_container_equals_float32_and_rounding_is_opt_in(monkeypatch):
    from descriptools_amd import downslope, flowhand, slope
    # (a) the reference-generated synthetic fixture handed over as float64 (what gen_golden feeds the reference)
    s = golden("syn_b")
    px = float(s["px"])
    d64 = s["dem"].astype(np.float64)
    assert np.array_equal(slope.sloper(d64, px).astype(np.float32), s["slope"])
    fd, idx, hand = flowhand.flow_hand_index(d64, s["fdr"], s["river"], px)
    assert hand.dtype == np.float64 and np.array_equal(hand.astype(np.float32), s["hand"]) and np.array_equal(idx, s["idx"])
    ref = np.where(np.isnan(s["down"]), 0, s["down"])  # 0 / 0 at the reference's pits: the build returns 0
    assert np.array_equal(downslope.downsloper(d64, s["fdr"], px, 5), ref)
    # (b) the genuinely float64 DEM: refused by default, and with the documented opt-in the results are the
    # reference's up to the rounding of the heights to float32 (2^-24 relative per height)
    g = golden("f64")
    dem64, px = g["dem"], float(g["px"])
    monkeypatch.setenv("DT_ALLOW_DEM_ROUNDING", "1")
    valid = dem64 != -100
    zmax = float(np.abs(dem64[valid]).max())
    ez = 2.0 ** -24 * zmax                      # rounding error of one height
    sl = slope.sloper(dem64, px)
    assert np.array_equal(sl == -100, g["slope"] == -100)
    assert np.abs(sl - g["slope"])[valid].max() <= 100.0 * 2 * ez / px * 1.01 + 1e-6 * float(g["slope"].max())
    fd, idx, hand = flowhand.flow_hand_index(dem64, g["fdr"], g["river"], px)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(fd, g["fdist"])
    ok = g["hand"] != -100
    assert np.array_equal(hand == -100, ~ok) and np.abs(hand - g["hand"])[ok].max() <= 2 * ez * 1.01
