"""Topographic index / modified topographic index -- HIP replacement of descriptools/topoindexes.py."""
import numpy as np

from . import _lib
from ._lib import c_f32p, c_i64p, check, ptr
from .device import host_empty, widen64


def topographic_index_cpu(flow_accumulation, slope, px, expoent, blocks=0, threads=0):
    """topoindexes.py:170-230 -> (ti, mti) float32; slope in RADIANS; nodata mask on fac only."""
    fac = np.ascontiguousarray(flow_accumulation, np.int64)
    sl = np.ascontiguousarray(slope, np.float32)
    ti = host_empty(fac.shape, np.float32)
    mti = host_empty(fac.shape, np.float32)
    check(_lib.lib().dt_twi(ptr(fac, c_i64p), ptr(sl, c_f32p), fac.size, float(px), float(expoent),
                            ptr(ti, c_f32p), ptr(mti, c_f32p)))
    return ti, mti


def topographic_index(flow_accumulation, slope, px, n_top, div_col=0, div_row=0):
    """topoindexes.py:109-167 -> two float64 rasters holding float32 values."""
    ti, mti = topographic_index_cpu(flow_accumulation, slope, px, n_top)
    return widen64(ti), widen64(mti)


def topographic_index_sequential_jit(flow_accumulation, slope, px):
    """Name of topoindexes.py:37; HIP path with the kernel's (normative) semantics."""
    return topographic_index_cpu(flow_accumulation, slope, px, 1.0)[0]


def modified_topographic_index_sequential_jit(flow_accumulation, slope, px, expoent):
    """Name of topoindexes.py:57; HIP path with the kernel's (normative) semantics."""
    return topographic_index_cpu(flow_accumulation, slope, px, expoent)[1]


topographic_index_sequential = topographic_index_sequential_jit
modified_topographic_index_sequential = modified_topographic_index_sequential_jit
