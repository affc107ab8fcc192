"""GPU (-m gpu): dense sweeps of the device math of dt_math.h / dt_stencil.hip through the public entry points,
against float64 libm (numpy): the table logarithms (GFI, ln(hl/H), the exact TI / MTI path), ln-tan up to pi/2, both
sides of the DT_FAST_MIN switch, the arctangent, and the fused stencil's arctangent-free TI / MTI.  The contract is
1e-5 relative; the float64 paths must in addition round to the SAME float32 as libm except at rounding ties."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulps(a, b):
    """distance in float32 units in the last place between float32 arrays a and b (finite, same sign)"""
    ia = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_gfi_logarithms_round_like_libm():
    """k_gfi / dt_gfi_both_cell: ln(b (A size^2)^n / (h + 0.01)) over 2 M (h, A) pairs, A up to 2^40, h from 0 to
    1e4 including the zero crossings of the index."""
    from descriptools_amd import gfi
    rng = np.random.default_rng(3)
    n = 1 << 21
    hand = np.concatenate([rng.random(n // 2) * 2.0, 10 ** (rng.random(n // 2) * 5 - 1)]).astype(np.float32)
    area = np.concatenate([rng.integers(1, 1 << 20, n // 2), 2 ** rng.integers(0, 40, n // 2) + rng.integers(0, 5, n // 2)])
    area = area.astype(np.int64)
    hand[::97] = -100
    for own_cell, fn in ((False, gfi.geomorphic_flood_index_cpu), (True, gfi.ln_hl_H_cpu)):
        got = np.asarray(fn(hand.reshape(1, -1), area.reshape(1, -1), 0.4, 0.1, 12.5)).reshape(-1)
        a = np.where((area == 0) & own_cell, 1, area).astype(np.float64)
        ref = np.log(0.1 * (a * 12.5 ** 2) ** 0.4 / (hand.astype(np.float64) + 0.01))
        ref32 = np.where(hand <= -100, -100, ref).astype(np.float32)
        ok = hand > -100
        u = _ulps(got[ok], ref32[ok])
        # float64 inside, one rounding: identical float32 except where libm's own last-bit errors meet a tie
        assert u.max() <= 1 and (u != 0).mean() < 1e-5, (u.max(), (u != 0).mean())
        assert np.array_equal(got[~ok], ref32[~ok])
        near0 = ok & (np.abs(ref) < 0.25)
        assert near0.sum() > 1000, "the sweep straddles the zero crossings"


def test_twi_pointwise_both_sides_of_the_fast_switch():
    """dt_twi (k_twi / dt_twi_cell): ln(A / tan(s + 0.01)) for s up to pi/2 - 0.01 and A up to 2^40; results with
    |TI| or |MTI| < DT_FAST_MIN take the float64 path (identical float32 rounding), the others the float32 fast
    path (<= 1e-5 relative by construction: <= 2e-7 absolute on values >= 0.25)."""
    from descriptools_amd import topoindexes
    rng = np.random.default_rng(5)
    n = 1 << 21
    s = np.concatenate([rng.random(n // 2) * 1.2, 1.2 + rng.random(n // 4) * (np.pi / 2 - 0.0101 - 1.2),
                        np.pi / 2 - 0.01 - 10 ** (-rng.random(n // 4) * 6 - 1)]).astype(np.float32)
    fac = np.concatenate([rng.integers(0, 50, n // 2), 2 ** rng.integers(0, 40, n // 2)]).astype(np.int64)
    rng.shuffle(fac)
    ti, mti = topoindexes.topographic_index_cpu(fac.reshape(1, -1), s.reshape(1, -1), 10.0, 0.1)
    A = np.maximum(fac, 1).astype(np.float64) * 100.0
    t = np.tan(s.astype(np.float64) + 0.01)
    for got, ref in ((ti, np.log(A / t)), (mti, np.log(A ** 0.1 / t))):
        got = np.asarray(got, np.float32).reshape(-1)
        ref32 = ref.astype(np.float32)
        rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-300)
        assert rel.max() <= 1e-5, rel.max()
        slow = (np.abs(ref) < 0.2) | (s > 1.21)  # safely inside the float64 path
        assert slow.sum() > 10000
        u = _ulps(got[slow], ref32[slow])
        assert u.max() <= 1 and (u != 0).mean() < 1e-4, (u.max(), (u != 0).mean())
        fast = (np.abs(ref) > 0.3) & (s < 1.19)
        assert np.abs(got[fast].astype(np.float64) - ref[fast]).max() <= 4e-7 + 6e-8 * np.abs(ref[fast]).max()


def _slope_probe(q):
    """a DEM whose cell (3k, 0) has slope % = 100 q[k] exactly representable as 100 * h / px: rows [h, 0] separated
    by nodata rows"""
    n = len(q)
    dem = np.full((3 * n, 2), -100.0, np.float32)
    dem[::3, 0] = q
    dem[::3, 1] = 0.0
    return dem


def test_stencil_arctangent_and_fused_twi_sweep():
    """slope -> radians (dt_slope_rad) within 2 float32 ulp (2.4e-7) of the float64 arctangent over q = 1e-6 .. 1e6, and the
    fused stencil's arctangent-free TI / MTI (sd_twi_fast) plus its exact fix-up path against the float64
    expression, both sides of DT_FAST_MIN and of the q <= 2.5 fast domain."""
    import torch
    from descriptools_amd import _lib
    from descriptools_amd.device import Context
    rng = np.random.default_rng(11)
    n = 1 << 17
    q = np.concatenate([10 ** (rng.random(n // 2) * 12 - 6), rng.random(n // 4) * 3.0,
                        0.41421356 + (rng.random(n // 8) - 0.5) * 1e-4, 2.41421356 + (rng.random(n // 8) - 0.5) * 1e-4])
    q = q.astype(np.float32)
    dem = _slope_probe(q)           # px = 1: slope % = 100 * h
    H, W = dem.shape
    fac = np.zeros((H, W), np.int32)
    fac[::3, 0] = np.concatenate([rng.integers(0, 3, n // 2), 2 ** rng.integers(0, 31, n // 2) - 1]).astype(np.int32)
    L = _lib.lib()
    ctx = Context()
    d_dem, d_fac = ctx.to_device(dem), ctx.to_device(fac)
    outs = {k: ctx.empty((H, W), np.float32) for k in ("slope", "rad", "ti", "mti")}
    _lib.check(L.dt_dev_slope_twi(ctx.h, d_dem.ptr, d_fac.ptr, H, W, 1.0, 0.1, outs["slope"].ptr, outs["rad"].ptr,
                                  outs["ti"].ptr, outs["mti"].ptr))
    ctx.sync()
    sl, rad, ti, mti = (outs[k].to_host()[::3, 0] for k in ("slope", "rad", "ti", "mti"))
    for b in list(outs.values()) + [d_dem, d_fac]:
        b.free()
    ctx.close()
    assert np.array_equal(sl, (q.astype(np.float64) / 1.0 * 100.0).astype(np.float32)), "slope %: exact"
    qq = (sl / np.float32(100.0)).astype(np.float32)
    want = np.arctan(qq.astype(np.float64))
    u = _ulps(rad, want.astype(np.float32))
    assert u.max() <= 2 and (u > 1).mean() < 1e-2, (u.max(), (u > 1).mean())  # float32 polynomial: 2 ulp at most
    assert np.abs(rad.astype(np.float64) - want).max() <= 2.4e-7
    # TI / MTI against the reference's expression on OUR float32 radians (what Example/example.py:63-69 feeds it)
    f = fac[::3, 0].astype(np.float64)
    A = np.maximum(f, 1.0)  # px = 1
    t = np.tan(rad.astype(np.float64) + 0.01)
    for got, ref in ((ti, np.log(A / t)), (mti, np.log(A ** 0.1 / t))):
        fin = np.isfinite(ref)
        rel = np.abs(got[fin].astype(np.float64) - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-30)
        # the contract, with the absolute floor the index needs right at a zero crossing (MTI crosses zero)
        bad = (rel > 1e-5) & (np.abs(got[fin].astype(np.float64) - ref[fin]) > 1e-6)
        assert not bad.any(), (int(bad.sum()), float(rel.max()))
        assert ((np.abs(ref[fin]) < 0.25).sum() > 100) and ((qq > 2.5).sum() > 1000) and ((qq < 2.5).sum() > 1000)
