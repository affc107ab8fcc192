"""GPU (-m gpu): the integer window walk of the downslope kernel (k_downslope_q: DEMs whose heights lie on a binary
quantum -- integer rasters, the 2^-8 m synthetic DEM) against the oracle's literal walk (downslope.py:161-314 +
:435-532), its hand-backs to the float kernel (window off the quantum / out of the 16-bit range / not interior), and
equality with the float kernel alone (dt_debug_set(7, 1)) at 4096^2.  Rasters here are >= 256 x 512: below that the
quantised kernel is not launched (tests/test_gpu_parity.py covers the float kernel)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import oracle
    import descriptools_amd.downslope as downslope
    from descriptools_amd import _lib
    return oracle, downslope, _lib.lib()


def _same(got, want):
    assert np.array_equal(got, want, equal_nan=True), int((got != want).sum())


@pytest.mark.parametrize("quantum", [1.0, 1.0 / 256, 8.0])
def test_arbitrary_direction_field_on_a_quantum(env, quantum):
    """every exit of the integer walk: drop reached, non-D8 codes, window ring, the 256-move limit, stepping onto
    nodata, start cells at / below the nodata value, the unsafe-rounding recheck"""
    oracle, downslope, L = env
    rng = np.random.default_rng(int(quantum * 256))
    H, W = 420, 900
    codes = np.array([1, 2, 4, 8, 16, 32, 64, 128, 0, 3, 255], np.uint8)
    fdr = codes[rng.integers(0, len(codes), size=(H, W))]
    fdr[rng.random((H, W)) < 0.5] = 4
    dem = (rng.integers(0, int(48 / quantum) + 2, size=(H, W)) * quantum).astype(np.float32)
    dem[:, 500:] = np.float32(8.0)               # flat part: walks run until a cycle, the edge or the cap
    dem[rng.random((H, W)) < 0.02] = -100
    dem[200:203, 300:303] = np.float32(-104.0)   # below the nodata value but not equal to it (on every quantum here)
    for dz in (5.0, 0.3, 0.0):
        _same(downslope.downsloper(dem, fdr, 10.0, dz), oracle.downslope(dem, fdr, 10.0, dz))
    # the float kernel alone gives the same raster
    L.dt_debug_set(7, 1)
    try:
        alone = downslope.downsloper(dem, fdr, 10.0, 5.0)
    finally:
        L.dt_debug_set(7, 0)
    _same(alone, oracle.downslope(dem, fdr, 10.0, 5.0))


def test_terrain_with_windows_handed_back(env):
    """2^-8 m terrain with nodata blobs; windows that must fall back: one height off the quantum, a plateau lifted
    beyond the 16-bit range, a NaN, a huge negative height"""
    oracle, downslope, _ = env
    H, W = 1100, 1500
    dem = oracle.synth_dem(7, 4096, 4096, 300, 500, H, W, 3)
    dem[400, 700] += np.float32(2.0 ** -12)      # off the quantum: that window (and its neighbours' margins) fail the check
    dem[600:700, 200:330] += np.float32(1000.0)  # 1000 m step: range > 65534 quanta in the windows across it
    dem[900, 1200] = np.nan
    dem[150, 1300] = np.float32(-9999.0)
    _, fdr = oracle.slope_d8(np.nan_to_num(dem, nan=0.0), 10.0)
    for dz in (5.0, 1.0):
        _same(downslope.downsloper(dem, fdr, 10.0, dz), oracle.downslope(dem, fdr, 10.0, dz))


def test_dem_off_any_quantum_takes_the_float_kernel(env):
    oracle, downslope, _ = env
    rng = np.random.default_rng(3)
    H, W = 300, 700
    dem = (oracle.synth_dem(5, 2048, 2048, 0, 0, H, W, 2) + rng.random((H, W)).astype(np.float32) * np.float32(0.01))
    dem[oracle.synth_dem(5, 2048, 2048, 0, 0, H, W, 2) == -100] = -100
    _, fdr = oracle.slope_d8(dem, 12.5)
    _same(downslope.downsloper(dem, fdr, 12.5, 5.0), oracle.downslope(dem, fdr, 12.5, 5.0))


def test_int16_example_like_dem(env):
    """integer metres (qe >= 0), the Example's dtype: D8 from the DEM itself, raw -50 marks included"""
    oracle, downslope, _ = env
    H, W = 700, 1300
    dem = np.floor(oracle.synth_dem(11, 2048, 2048, 100, 200, H, W, 2) * 0.5).astype(np.float32)
    dem[oracle.synth_dem(11, 2048, 2048, 100, 200, H, W, 2) == -100] = -100
    _, fdr = oracle.slope_d8(dem, 30.0)
    want = oracle.downslope(dem, fdr, 30.0, 5.0)
    _same(downslope.downsloper(dem, fdr, 30.0, 5.0), want)
    raw = downslope.downslope_cpu(dem, fdr, 30.0, 5.0)
    keep = raw != -50
    _same(raw[keep], want[keep])


def test_equals_the_float_kernel_at_4096(env):
    oracle, downslope, L = env
    import torch
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    n = 4096
    ctx = Context()
    dem = torch.empty((n, n), dtype=torch.float32, device="cuda")
    fdr = torch.empty((n, n), dtype=torch.uint8, device="cuda")
    a = torch.empty((n, n), dtype=torch.float32, device="cuda")
    b = torch.empty((n, n), dtype=torch.float32, device="cuda")
    _lib.check(L.dt_dev_synth_dem(ctx.h, 3, n, n, 0, 0, n, n, 2, dem.data_ptr()))
    _lib.check(L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), n, n, 10.0, None, fdr.data_ptr(), None))
    _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), n, n, 10.0, 5.0, 0, a.data_ptr()))
    L.dt_debug_set(7, 1)
    try:
        _lib.check(L.dt_dev_downslope(ctx.h, dem.data_ptr(), fdr.data_ptr(), n, n, 10.0, 5.0, 0, b.data_ptr()))
        ctx.sync()
    finally:
        L.dt_debug_set(7, 0)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert int((a > 0).sum()) > n * n // 2  # real walks, not an early-out
