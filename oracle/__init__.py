"""CPU oracle for the descriptools hot path -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/dt_oracle.c (see its header for what it restates and how it is
pinned).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; descriptools_amd/ never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdt_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "dt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_SO_BENCH = os.path.join(_HERE, "_build", "libdt_oracle_bench.so")


def build_bench(force=False):
    """the timed CPU-baseline build (bench.py cpu_baseline): -O3 -march=native + OpenMP over the per-cell loops"""
    src = os.path.join(_HERE, "dt_oracle.c")
    if force or not os.path.exists(_SO_BENCH) or os.path.getmtime(_SO_BENCH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "bench"])
    return _SO_BENCH


_lib = None


def use_bench_build(threads):
    """switch this module to the baseline build with `threads` OpenMP threads (None: back to the checker build)"""
    global _lib
    if threads is None:
        _lib = None
        return
    _lib = C.CDLL(build_bench())
    C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def synth_dem(seed, Hg, Wg, y0=0, x0=0, h=None, w=None, nodata_pct=0):
    h = Hg if h is None else h
    w = Wg if w is None else w
    out = np.empty((h, w), np.float32)
    lib().dt_oracle_synth_dem(C.c_uint32(seed), C.c_int64(Hg), C.c_int64(Wg), C.c_int64(y0),
                              C.c_int64(x0), C.c_int64(h), C.c_int64(w), C.c_int(nodata_pct),
                              _p(out, C.c_float))
    return out


def slope_d8(dem, px):
    dem = np.ascontiguousarray(dem, np.float32)
    H, W = dem.shape
    sl = np.empty((H, W), np.float32)
    fdr = np.empty((H, W), np.uint8)
    lib().dt_oracle_slope_d8_f32(_p(dem, C.c_float), C.c_int64(H), C.c_int64(W), C.c_double(px),
                                 _p(sl, C.c_float), _p(fdr, C.c_uint8))
    return sl, fdr


def flowacc(fdr, dem=None):
    fdr = np.ascontiguousarray(fdr, np.uint8)
    H, W = fdr.shape
    if dem is not None:
        dem = np.ascontiguousarray(dem, np.float32)
    acc = np.empty((H, W), np.int64)
    rc = lib().dt_oracle_flowacc(_p(fdr, C.c_uint8), _p(dem, C.c_float), C.c_int64(H), C.c_int64(W),
                                 _p(acc, C.c_int64))
    assert rc == 0
    return acc


def flowhand(dem, fdr, river, px):
    dem = np.ascontiguousarray(dem, np.float32)
    fdr = np.ascontiguousarray(fdr, np.uint8)
    river = np.ascontiguousarray(river, np.int8)
    H, W = fdr.shape
    fd = np.empty((H, W), np.float32)
    idx = np.empty((H, W), np.int64)
    hand = np.empty((H, W), np.float32)
    lib().dt_oracle_flowhand(_p(dem, C.c_float), _p(fdr, C.c_uint8), _p(river, C.c_int8),
                             C.c_int64(H), C.c_int64(W), C.c_double(px), _p(fd, C.c_float),
                             _p(idx, C.c_int64), _p(hand, C.c_float))
    return fd, idx, hand


def flowhand_fast(fdr, river):
    fdr = np.ascontiguousarray(fdr, np.uint8)
    river = np.ascontiguousarray(river, np.int8)
    H, W = fdr.shape
    idx = np.empty((H, W), np.int64)
    nc = np.empty((H, W), np.int32)
    nd = np.empty((H, W), np.int32)
    rc = lib().dt_oracle_flowhand_fast(_p(fdr, C.c_uint8), _p(river, C.c_int8), C.c_int64(H),
                                       C.c_int64(W), _p(idx, C.c_int64), _p(nc, C.c_int32),
                                       _p(nd, C.c_int32))
    assert rc == 0
    return idx, nc, nd


def twi(fac, slope_rad, px, n):
    fac = np.ascontiguousarray(fac, np.int64)
    sl = np.ascontiguousarray(slope_rad, np.float32)
    ti = np.empty(fac.shape, np.float32)
    mti = np.empty(fac.shape, np.float32)
    lib().dt_oracle_twi(_p(fac, C.c_int64), _p(sl, C.c_float), C.c_int64(fac.size), C.c_double(px),
                        C.c_double(n), _p(ti, C.c_float), _p(mti, C.c_float))
    return ti, mti


def gfi(hand, fac, idx, n, b, size):
    hand = np.ascontiguousarray(hand, np.float32)
    fac = np.ascontiguousarray(fac, np.int64)
    idx = np.ascontiguousarray(idx, np.int64)
    out = np.empty(hand.shape, np.float32)
    lib().dt_oracle_gfi(_p(hand, C.c_float), _p(fac, C.c_int64), _p(idx, C.c_int64),
                        C.c_int64(hand.size), C.c_double(n), C.c_double(b), C.c_double(size),
                        _p(out, C.c_float))
    return out


def lnhlh(hand, fac, n, b, size):
    hand = np.ascontiguousarray(hand, np.float32)
    fac = np.ascontiguousarray(fac, np.int64)
    out = np.empty(hand.shape, np.float32)
    lib().dt_oracle_lnhlh(_p(hand, C.c_float), _p(fac, C.c_int64), C.c_int64(hand.size),
                          C.c_double(n), C.c_double(b), C.c_double(size), _p(out, C.c_float))
    return out


def downslope(dem, fdr, px, dz):
    dem = np.ascontiguousarray(dem, np.float32)
    fdr = np.ascontiguousarray(fdr, np.uint8)
    H, W = dem.shape
    out = np.empty((H, W), np.float32)
    lib().dt_oracle_downslope(_p(dem, C.c_float), _p(fdr, C.c_uint8), C.c_int64(H), C.c_int64(W),
                              C.c_double(px), C.c_double(dz), _p(out, C.c_float))
    return out


def slope_f64(dem, px):
    """slope (%) of a float64 DEM, differences in float64 (slope.py:244-258 on such a raster)"""
    dem = np.ascontiguousarray(dem, np.float64)
    H, W = dem.shape
    sl = np.empty((H, W), np.float32)
    lib().dt_oracle_slope_f64(_p(dem, C.c_double), C.c_int64(H), C.c_int64(W), C.c_double(px), _p(sl, C.c_float))
    return sl


def hand_f64(dem, idx):
    dem = np.ascontiguousarray(dem, np.float64)
    idx = np.ascontiguousarray(idx, np.int64)
    hand = np.empty(dem.shape, np.float64)
    lib().dt_oracle_hand_f64(_p(dem, C.c_double), _p(idx, C.c_int64), C.c_int64(dem.size), _p(hand, C.c_double))
    return hand


def downslope_f64(dem, fdr, px, dz):
    dem = np.ascontiguousarray(dem, np.float64)
    fdr = np.ascontiguousarray(fdr, np.uint8)
    H, W = dem.shape
    out = np.empty((H, W), np.float32)
    lib().dt_oracle_downslope_f64(_p(dem, C.c_double), _p(fdr, C.c_uint8), C.c_int64(H), C.c_int64(W),
                                  C.c_double(px), C.c_double(dz), _p(out, C.c_float))
    return out


def gfi_f64h(hand, fac, idx, n, b, size):
    hand = np.ascontiguousarray(hand, np.float64)
    fac = np.ascontiguousarray(fac, np.int64)
    idx = np.ascontiguousarray(idx, np.int64)
    out = np.empty(hand.shape, np.float32)
    lib().dt_oracle_gfi_f64h(_p(hand, C.c_double), _p(fac, C.c_int64), _p(idx, C.c_int64), C.c_int64(hand.size),
                             C.c_double(n), C.c_double(b), C.c_double(size), _p(out, C.c_float))
    return out


def lnhlh_f64h(hand, fac, n, b, size):
    hand = np.ascontiguousarray(hand, np.float64)
    fac = np.ascontiguousarray(fac, np.int64)
    out = np.empty(hand.shape, np.float32)
    lib().dt_oracle_lnhlh_f64h(_p(hand, C.c_double), _p(fac, C.c_int64), C.c_int64(hand.size), C.c_double(n),
                               C.c_double(b), C.c_double(size), _p(out, C.c_float))
    return out


def confusion_multi(desc, flood, th, under=True):
    desc = np.ascontiguousarray(desc, np.float64)
    flood = np.ascontiguousarray(flood, np.int8)
    th = np.ascontiguousarray(th, np.float64)
    counts = np.zeros((th.size, 4), np.int64)
    lib().dt_oracle_confusion_multi(_p(desc, C.c_double), _p(flood, C.c_int8), C.c_int64(desc.size),
                                    _p(th, C.c_double), C.c_int(th.size), C.c_int(1 if under else 0),
                                    _p(counts, C.c_int64))
    return counts


def condition_d8(dem, px):
    """(fdr, filled): D8 on the depression-filled surface with flats resolved (the definition of
    dt_d8_conditioned_f32)"""
    dem = np.ascontiguousarray(dem, np.float32)
    H, W = dem.shape
    filled = np.empty((H, W), np.float32)
    fdr = np.empty((H, W), np.uint8)
    f = lib().dt_oracle_condition_d8
    f.restype = C.c_int64
    rc = f(_p(dem, C.c_float), C.c_int64(H), C.c_int64(W), C.c_double(px), _p(filled, C.c_float), _p(fdr, C.c_uint8))
    assert rc == 0, "%d flat cells without a code" % rc
    return fdr, filled
