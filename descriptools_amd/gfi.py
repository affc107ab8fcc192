"""Geomorphic flood index and ln(hl/H) -- HIP replacement of descriptools/gfi.py."""
import numpy as np

from . import _lib
from ._lib import c_f32p, c_f64p, c_i64p, check, heights, ptr
from .device import host_empty, widen64


def river_accumulation(flow_accumulation, indices):
    """gfi.py:119-147: fac.flat[idx] where idx != -100, else fac.flat[0]."""
    fac = np.ascontiguousarray(flow_accumulation, np.int64)
    idx = np.ascontiguousarray(indices, np.int64)
    out = host_empty(fac.shape, np.int64)
    check(_lib.lib().dt_river_accumulation(ptr(fac, c_i64p), ptr(idx, c_i64p), fac.size, ptr(out, c_i64p)))
    return out


def _area_index(hand, area, expoent, scale_factor, size, zero_guard):
    h, wide = heights(hand, "HAND")
    a = np.ascontiguousarray(area, np.int64)
    out = host_empty(h.shape, np.float32)
    if wide:  # a HAND that float32 cannot hold (from a float64 DEM): hand + 0.01 on the float64 value, gfi.py:292-294
        check(_lib.lib().dt_gfi_f64h(ptr(h, c_f64p), ptr(a, c_i64p), None, h.size, float(expoent), float(scale_factor),
                                     float(size), 1 if zero_guard else 2, ptr(out, c_f32p)))
        return out
    check(_lib.lib().dt_gfi_area(ptr(h, c_f32p), ptr(a, c_i64p), h.size, float(expoent), float(scale_factor),
                                 float(size), zero_guard, ptr(out, c_f32p)))
    return out


def geomorphic_flood_index_cpu(hand, river_flow_accumulation, expoent, scale_factor, size, blocks=0,
                               threads=0):
    """gfi.py:210-294 -> float32; no zero-area guard (as the kernel)."""
    return _area_index(hand, river_flow_accumulation, expoent, scale_factor, size, 0)


def gfi_calculator(hand, flow_accumulation, indices, n_gfi, scale_factor, size, division_column=0,
                   division_row=0):
    """gfi.py:150-207 -> float64 raster holding float32 values."""
    h, wide = heights(hand, "HAND")
    fac = np.ascontiguousarray(flow_accumulation, np.int64)
    idx = np.ascontiguousarray(indices, np.int64)
    out = host_empty(h.shape, np.float32)
    if wide:
        check(_lib.lib().dt_gfi_f64h(ptr(h, c_f64p), ptr(fac, c_i64p), ptr(idx, c_i64p), h.size, float(n_gfi),
                                     float(scale_factor), float(size), 0, ptr(out, c_f32p)))
    else:
        check(_lib.lib().dt_gfi(ptr(h, c_f32p), ptr(fac, c_i64p), ptr(idx, c_i64p), h.size, float(n_gfi),
                                float(scale_factor), float(size), ptr(out, c_f32p)))
    return widen64(out)


def ln_hl_H_cpu(hand, flow_accumulation, expoent, scale_factor, size, blocks=0, threads=0):
    """gfi.py:349-440 -> float32; own-cell area, fac == 0 -> 1."""
    return _area_index(hand, flow_accumulation, expoent, scale_factor, size, 1)


def ln_hl_H_calculator(hand, flow_accumulation, n_gfi, scale_factor, size, division_column=0,
                       division_row=0):
    """gfi.py:297-346 -> float64 raster holding float32 values."""
    return widen64(ln_hl_H_cpu(hand, flow_accumulation, n_gfi, scale_factor, size))


def geomorphic_flood_index_sequential_jit(hand, river_flow_accumulation, expoent, scale_factor, size):
    """Name of gfi.py:46; HIP path, kernel semantics."""
    return geomorphic_flood_index_cpu(hand, river_flow_accumulation, expoent, scale_factor, size)


def ln_hl_H_sequential_jit(hand, flow_accumulation, expoent, scale_factor, size):
    """Name of gfi.py:65; HIP path, kernel semantics."""
    return ln_hl_H_cpu(hand, flow_accumulation, expoent, scale_factor, size)


geomorphic_flood_index_sequential = geomorphic_flood_index_sequential_jit
ln_hl_H_sequential = ln_hl_H_sequential_jit
