"""Flood-map evaluation -- replacement of descriptools/evaluation.py.

Every raster-returning function is an H2D -> kernel -> D2H shim like the other modules: `minMaxScale`
(k_minmax_scale_t), `binary_map` and `avaliacao` (k_classify).  `calibration` -- 61 x (binary_map + avaliacao) in
the reference -- runs as 5 multi-threshold confusion-count passes over rasters uploaded once (k_confusion); counts
are exact integers, so the returned threshold is identical.  The numpy dtype rules of the reference's expressions
are kept at the boundary (which arithmetic a float32 / integer raster is scaled and compared in, int64 maps out,
the in-place remap of the benchmark map)."""
import numpy as np

from . import _lib
from ._lib import C, c_f64p, c_i8p, c_i32p, c_i64p, c_u8p, check, ptr
from .device import Context


def _float_view(mat, *scalars):
    """(contiguous float array, is_f32): the float dtype numpy's expressions give a raster of this dtype when
    combined with `scalars` -- float32 rasters stay float32 unless a float64 numpy scalar is involved (Python
    numbers are weak), every other dtype is computed in float64."""
    mat = np.asarray(mat)
    base = np.float32 if mat.dtype in (np.float32, np.float16) else np.float64
    rt = np.result_type(base, *scalars)
    rt = np.float32 if rt == np.float32 else np.float64
    return np.ascontiguousarray(mat, rt), rt == np.float32


def minMaxScale(mat, mn, mx, nodata):
    """evaluation.py:5-9 on the GPU: NaN where mat == nodata (or mat is NaN), (mat - mn) / (mx - mn) elsewhere."""
    x, is_f32 = _float_view(mat, mn, mx)
    out = np.empty_like(x)
    check(_lib.lib().dt_minmax_scale(x.ctypes.data_as(C.c_void_p), int(is_f32), x.size, float(mn), float(mx),
                                     float(nodata), out.ctypes.data_as(C.c_void_p)))
    return out


def binary_map(descriptor_matrix, threshold, under):
    """evaluation.py:90-123 on the GPU; int64 map out; the value at [0, 0] counts as nodata (:111)."""
    desc, is_f32 = _float_view(descriptor_matrix, threshold)
    first = float(desc.reshape(-1)[0]) if desc.size else 0.0
    out = np.empty(desc.shape, np.uint8)
    check(_lib.lib().dt_binary_map(desc.ctypes.data_as(C.c_void_p), int(is_f32), desc.size, first, float(threshold),
                                   1 if under == 'under' else 0, ptr(out, c_u8p)))
    return out.astype(np.int64)


def correctness(count):
    """evaluation.py:174-191: share of the benchmark's flooded cells the descriptor map also floods,
    class 3 / (class 2 + class 3)."""
    hit, miss = count[3], count[2]
    return hit / (miss + hit)


def fit(count):
    """evaluation.py:194-211: class 3 over every cell either map floods."""
    hit, miss, false_alarm = count[3], count[2], count[1]
    return hit / (hit + miss + false_alarm)


def avaliacao(descriptor_flood_map, comparison_flood_map):
    """evaluation.py:126-171 on the GPU.  Like the reference it rewrites comparison_flood_map in place
    (1 -> 2, -100 -> 0: the kernel's remapped copy is written back into the caller's array); returns
    (correctness, fit, class map = descriptor map + remapped benchmark map)."""
    b32 = np.ascontiguousarray(descriptor_flood_map, np.int32)
    cmp8 = np.ascontiguousarray(comparison_flood_map, np.int8)
    klass = np.empty(b32.shape, np.int32)
    counts = np.zeros(4, np.int64)
    check(_lib.lib().dt_avaliacao(ptr(b32, c_i32p), ptr(cmp8, c_i8p), b32.size, ptr(klass, c_i32p),
                                  ptr(counts, c_i64p)))
    comparison_flood_map[...] = cmp8.reshape(np.shape(comparison_flood_map))
    result = klass.astype(np.result_type(np.asarray(descriptor_flood_map).dtype,
                                         np.asarray(comparison_flood_map).dtype))
    with np.errstate(divide='ignore', invalid='ignore'):
        return correctness(counts), fit(counts), result


class _Calibrator:
    """descriptor + benchmark resident on the GPU; one launch per calibration stage."""

    def __init__(self, descriptor_matrix, comparison_matrix, under):
        desc = np.asarray(descriptor_matrix)
        self.f32 = desc.dtype == np.float32
        self.nodata = float(desc.reshape(-1)[0]) if desc.size else 0.0
        d64 = np.ascontiguousarray(desc, np.float64)
        cmp8 = np.ascontiguousarray(comparison_matrix, np.int8)
        self.n = d64.size
        self.under = 1 if under == 'under' else 0
        self.ctx = Context()
        self.d_desc = self.ctx.to_device(d64)
        self.d_cmp = self.ctx.to_device(cmp8)
        self.d_counts = self.ctx.empty(24 * 4, np.int64)

    def fits(self, thresholds):
        th = np.asarray(thresholds, np.float64)
        if self.f32:  # numpy compares a float32 raster with a Python float in float32
            th = th.astype(np.float32).astype(np.float64)
        th = np.ascontiguousarray(th)
        check(_lib.lib().dt_dev_confusion_multi(self.ctx.h, self.d_desc.ptr, self.d_cmp.ptr, self.n,
                                                self.nodata, ptr(th, c_f64p), len(th), self.under,
                                                self.d_counts.ptr))
        counts = self.d_counts.to_host()[:4 * len(th)].reshape(len(th), 4)
        with np.errstate(divide='ignore', invalid='ignore'):
            return [fit(c) for c in counts]

    def remap_into(self, comparison_matrix):
        """the benchmark map remapped on the device (1 -> 2, -100 -> 0, evaluation.py:149-150) and copied back
        into the caller's array, which the reference's first avaliacao call mutates"""
        check(_lib.lib().dt_dev_classify(self.ctx.h, self.d_desc.ptr, self.d_cmp.ptr, self.n, self.nodata, 0.0,
                                         self.under, 1, None, None, self.d_counts.ptr))
        np.copyto(comparison_matrix, self.d_cmp.to_host().reshape(np.shape(comparison_matrix)), casting='unsafe')

    def close(self):
        for b in (self.d_desc, self.d_cmp, self.d_counts):
            b.free()
        self.ctx.close()


def _grid_search(fits):
    """evaluation.py:12-87: the 4-stage grid search, given fits(list of thresholds) -> list of Fit indexes."""
    f1, f2, f3 = fits([25 / 100, 50 / 100, 75 / 100])
    if f3 > f2:
        fit_index, iteration_value = (f3, 75) if f3 > f1 else (f1, 25)
    else:
        fit_index, iteration_value = (f2, 50) if f2 > f1 else (f1, 25)
    rng = list(range(iteration_value - 20, iteration_value + 30, 10))
    for i, f in zip(rng, fits([i / 100 for i in rng])):
        if f >= fit_index:
            fit_index, threshold = f, i
    iteration_value = threshold
    rng = list(range(iteration_value - 5, iteration_value + 6, 1))
    for i, f in zip(rng, fits([i / 100 for i in rng])):
        if f > fit_index:
            fit_index, threshold = f, i
    for div in (1000, 10000):
        iteration_value = threshold * 10
        threshold = iteration_value
        rng = list(range(iteration_value - 10, iteration_value + 11, 1))
        for i, f in zip(rng, fits([i / div for i in rng])):
            if f > fit_index:
                fit_index, threshold = f, i
    return threshold / 10000


def calibration(descriptor_matrix, comparison_matrix, under):
    """evaluation.py:12-87: 4-stage grid search for the threshold maximising the Fit index."""
    cal = _Calibrator(descriptor_matrix, comparison_matrix, under)
    try:
        cal.remap_into(comparison_matrix)
        return _grid_search(cal.fits)
    finally:
        cal.close()


def combine_extremes(per_rank):
    """np.unique extremes of a raster split over ranks from each rank's (smallest, second-smallest
    distinct, largest): the global second-smallest is the smallest candidate above the global minimum."""
    a = np.asarray(per_rank, np.float64).reshape(-1, 3)
    lo = np.nanmin(a[:, 0])
    cand = np.where(a[:, 0] > lo, a[:, 0], a[:, 1])
    return np.array([lo, np.nanmin(cand) if np.isfinite(cand).any() else np.nan, np.nanmax(a[:, 2])],
                    np.float32)


def evaluate_resident(ctx, x_ptr, flood_ptr, n, under='under', nodata=-100.0, desc_ptr=None,
                      reduce_extremes=None, reduce_counts=None, nodata_first=None, integer_valued=False,
                      binary_ptr=None, class_ptr=None, remap_flood=False):
    """Example/example.py:113-147 on rasters that are already in HBM (net-new):
    np.unique extremes -> minMaxScale -> calibration -> confusion counts at the calibrated threshold.
    x is a float32 descriptor raster (e.g. the chain's HAND), flood an int8 benchmark map, both flat
    device pointers of n cells.  reduce_extremes(np.float32[3]) / reduce_counts(np.int64[k, 4]) hook in
    the extremes combination (combine_extremes over an all-gather) and the sum all-reduce of a multi-GPU
    run; nodata_first = the scaled value at global cell [0, 0] (binary_map's nodata, evaluation.py:111)
    when this rank does not own that cell.  integer_valued: x holds integers (the example's int16 HAND kept as
    float32 on the device), which numpy scales and compares in float64 (float32 otherwise).  binary_ptr (uint8[n])
    / class_ptr (int32[n]): device rasters that receive binary_map's map and avaliacao's class map at the
    calibrated threshold (Example/example.py:139-147, what example.py:201-217 writes to disk); remap_flood: the
    benchmark map is left remapped in place as avaliacao leaves it.  Returns a dict."""
    import ctypes as C
    L = _lib.lib()
    own_desc = None
    ext = ctx.empty(3, np.float32)
    counts = ctx.empty(24 * 4, np.int64)
    try:
        check(L.dt_dev_unique_extremes_f32(ctx.h, x_ptr, n, ext.ptr))
        e = ext.to_host()
        if reduce_extremes is not None:
            e = reduce_extremes(e)
        lo, mn, mx = (np.float32(v) for v in e)
        if desc_ptr is None:
            own_desc = ctx.empty(n, np.float64)
            desc_ptr = own_desc.ptr
        if integer_valued:
            check(L.dt_dev_minmax_scale_f32_f64(ctx.h, x_ptr, n, float(mn), float(mx), float(nodata), desc_ptr))
        else:
            check(L.dt_dev_minmax_scale_f32(ctx.h, x_ptr, n, mn, mx, np.float32(nodata), desc_ptr))
        # binary_map treats the value at [0, 0] as nodata (evaluation.py:111); after minMaxScale that is
        # NaN wherever the first cell is nodata -- NaN never compares equal, so pass NaN
        first = np.empty(1, np.float64)
        check(L.dt_dev_d2h(ctx.h, first.ctypes.data_as(C.c_void_p), desc_ptr, 8))
        # (a callable nodata_first is evaluated here, after the extremes are known: a rank that does not own the
        # global cell [0, 0] learns its scaled value with the extremes exchange)
        nod_val = float(first[0]) if nodata_first is None else (nodata_first() if callable(nodata_first) else nodata_first)
        under_i = 1 if under == 'under' else 0

        def count(ths):
            # a float32 descriptor compares with the thresholds in float32 (numpy's weak Python scalars)
            th = np.asarray(ths, np.float64)
            if not integer_valued:
                th = th.astype(np.float32).astype(np.float64)
            th = np.ascontiguousarray(th)
            check(L.dt_dev_confusion_multi(ctx.h, desc_ptr, flood_ptr, n, nod_val, ptr(th, c_f64p), len(th),
                                           under_i, counts.ptr))
            c = counts.to_host()[:4 * len(th)].reshape(len(th), 4).copy()
            return reduce_counts(c) if reduce_counts is not None else c

        def fits(ths):
            with np.errstate(divide='ignore', invalid='ignore'):
                return [fit(c) for c in count(ths)]

        th = _grid_search(fits)
        if binary_ptr is not None or class_ptr is not None or remap_flood:
            thc = float(th) if integer_valued else float(np.float32(th))
            check(L.dt_dev_classify(ctx.h, desc_ptr, flood_ptr, n, nod_val, thc, under_i, 1 if remap_flood else 0,
                                    binary_ptr, class_ptr, counts.ptr))
            c4 = counts.to_host()[:4].copy()
            if reduce_counts is not None:
                c4 = reduce_counts(c4.reshape(1, 4))[0]
        else:
            c4 = count([th])[0]
        with np.errstate(divide='ignore', invalid='ignore'):
            return {"mn": float(mn), "mx": float(mx), "threshold": th, "counts": c4,
                    "correctness": correctness(c4), "fit": fit(c4)}
    finally:
        ext.free()
        counts.free()
        if own_desc is not None:
            own_desc.free()
