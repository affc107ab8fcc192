"""Flood-map evaluation -- replacement of descriptools/evaluation.py.

minMaxScale / binary_map / avaliacao return full host rasters and keep the reference's numpy
semantics literally (dtype rules, in-place remap of the benchmark map, the value at [0,0] as nodata).
`calibration` -- 61 x (binary_map + avaliacao) in the reference -- runs as 5 multi-threshold
confusion-count passes on the GPU over rasters uploaded once; counts are exact integers, so the
returned threshold is identical."""
import numpy as np

from . import _lib
from ._lib import C, c_f64p, check, ptr
from .device import Context


def minMaxScale(mat, mn, mx, nodata):
    """evaluation.py:5-9."""
    scaled = np.where(mat == nodata, np.nan, mat)
    scaled = np.where(np.isnan(mat), scaled, (scaled - mn) / (mx - mn))
    return scaled


def binary_map(descriptor_matrix, threshold, under):
    """evaluation.py:90-123."""
    descriptor_matrix = np.where(descriptor_matrix == descriptor_matrix[0, 0], np.nan, descriptor_matrix)
    if under == 'under':
        return np.where(np.isnan(descriptor_matrix), 0, np.where(descriptor_matrix <= threshold, 1, 0))
    return np.where(np.isnan(descriptor_matrix), 0, np.where(descriptor_matrix >= threshold, 1, 0))


def correctness(count):
    """evaluation.py:174-191."""
    return ((count[3]) / (count[2] + count[3]))


def fit(count):
    """evaluation.py:194-211."""
    return ((count[3]) / (count[3] + count[2] + count[1]))


def avaliacao(descriptor_flood_map, comparison_flood_map):
    """evaluation.py:126-171 (mutates comparison_flood_map in place like the reference)."""
    comparison_flood_map[comparison_flood_map == 1] = 2
    comparison_flood_map[comparison_flood_map == -100] = 0
    result = descriptor_flood_map + comparison_flood_map
    elements, count = np.unique(result, return_counts=True)
    for v in range(4):
        if not np.any(elements == v):
            count = np.insert(count, v, 0)
            elements = np.insert(elements, v, v)
    return correctness(count), fit(count), result


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
        # the reference's first avaliacao call remaps the caller's benchmark map in place
        comparison_matrix[comparison_matrix == 1] = 2
        comparison_matrix[comparison_matrix == -100] = 0
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
                      reduce_extremes=None, reduce_counts=None, nodata_first=None):
    """Example/example.py:113-147 on rasters that are already in HBM (net-new):
    np.unique extremes -> minMaxScale -> calibration -> confusion counts at the calibrated threshold.
    x is a float32 descriptor raster (e.g. the chain's HAND), flood an int8 benchmark map, both flat
    device pointers of n cells.  reduce_extremes(np.float32[3]) / reduce_counts(np.int64[k, 4]) hook in
    the extremes combination (combine_extremes over an all-gather) and the sum all-reduce of a multi-GPU
    run; nodata_first = the scaled value at global cell [0, 0] (binary_map's nodata, evaluation.py:111)
    when this rank does not own that cell.  Returns a dict."""
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
        check(L.dt_dev_minmax_scale_f32(ctx.h, x_ptr, n, mn, mx, np.float32(nodata), desc_ptr))
        # binary_map treats the value at [0, 0] as nodata (evaluation.py:111); after minMaxScale that is
        # NaN wherever the first cell is nodata -- NaN never compares equal, so pass NaN
        first = np.empty(1, np.float64)
        check(L.dt_dev_d2h(ctx.h, first.ctypes.data_as(C.c_void_p), desc_ptr, 8))
        nod_val = float(first[0]) if nodata_first is None else nodata_first
        under_i = 1 if under == 'under' else 0

        def count(ths):
            # the descriptor is a float32 raster for numpy: thresholds compare in float32
            th = np.ascontiguousarray(np.asarray(ths, np.float64).astype(np.float32).astype(np.float64))
            check(L.dt_dev_confusion_multi(ctx.h, desc_ptr, flood_ptr, n, nod_val, ptr(th, c_f64p), len(th),
                                           under_i, counts.ptr))
            c = counts.to_host()[:4 * len(th)].reshape(len(th), 4).copy()
            return reduce_counts(c) if reduce_counts is not None else c

        def fits(ths):
            with np.errstate(divide='ignore', invalid='ignore'):
                return [fit(c) for c in count(ths)]

        th = _grid_search(fits)
        c4 = count([th])[0]
        with np.errstate(divide='ignore', invalid='ignore'):
            return {"mn": float(mn), "mx": float(mx), "threshold": th, "counts": c4,
                    "correctness": correctness(c4), "fit": fit(c4)}
    finally:
        ext.free()
        counts.free()
        if own_desc is not None:
            own_desc.free()
