"""Flow distance to the nearest drainage, drained-to river index, HAND -- HIP replacement of
descriptools/flowhand.py (the reference's per-cell pointer chase becomes pointer doubling)."""
import numpy as np

from . import _lib
from ._lib import c_f32p, c_i8p, c_i64p, c_u8p, check, dem_f32, ptr


def _hand_dtype(dem):
    dt = np.asarray(dem).dtype
    return dt if dt.kind in "iuf" else np.dtype(np.float32)


def flow_hand_index(dem_raster, flow_direction_matrix, river_matrix, px, division_column=0,
                    division_row=0):
    """flowhand.py:242-411 -> (flow_distance float32, indices int64, hand in the DEM's dtype).
    One tile (division_* accepted, ignored: tiled == untiled is the reference's contract)."""
    dem32 = dem_f32(dem_raster)
    fdr = np.ascontiguousarray(flow_direction_matrix, np.uint8)
    river = np.ascontiguousarray(river_matrix, np.int8)
    H, W = fdr.shape
    fd = np.empty((H, W), np.float32)
    idx = np.empty((H, W), np.int64)
    hand = np.empty((H, W), np.float32)
    check(_lib.lib().dt_flowhand(ptr(dem32, c_f32p), ptr(fdr, c_u8p), ptr(river, c_i8p), H, W, float(px),
                                 ptr(fd, c_f32p), ptr(idx, c_i64p), ptr(hand, c_f32p)))
    return fd, idx, hand.astype(_hand_dtype(dem_raster))


def hand_calculator(dem, indices):
    """flowhand.py:414-442."""
    dem32 = dem_f32(dem)
    idx = np.ascontiguousarray(indices, np.int64)
    hand = np.empty(dem32.shape, np.float32)
    check(_lib.lib().dt_hand_f32(ptr(dem32, c_f32p), ptr(idx, c_i64p), dem32.size, ptr(hand, c_f32p)))
    return hand.astype(_hand_dtype(dem))


def index_calculator(river_indices, row_start, column_start, column_size):
    """flowhand.py:445-472 (unused by the reference itself): tile-local flat river indices ->
    indices of the whole raster; -100 stays -100.  Pure index arithmetic (host)."""
    river_indices = np.asarray(river_indices)
    col = river_indices.shape[1]
    return np.where(river_indices == -100, -100,
                    (np.floor(river_indices / col) + row_start) * column_size + river_indices % col + column_start)


def flow_distance_index_cpu(dem, flow_direction, river_matrix, px, boundary_distance, boundary_index,
                            out, row_start, col_start, matrix_columns, blocks=0, threads=0):
    """flowhand.py:476-562 for a tile without neighbouring tiles (out == 0 on all four sides, the
    only way the reference calls it when division_* == 0).  Indices are GLOBAL flat indices
    (row_start + r) * matrix_columns + col_start + c, returned as float64 like the reference."""
    if np.any(np.asarray(out) != 0):
        raise NotImplementedError("tile-exit boundary vectors: rasters are processed as one tile on "
                                  "MI355X; use flow_hand_index or descriptools_amd.tiling")
    fdr = np.ascontiguousarray(flow_direction, np.uint8)
    river = np.ascontiguousarray(river_matrix, np.int8)
    H, W = fdr.shape
    fd = np.empty((H, W), np.float32)
    idx = np.empty((H, W), np.int64)
    check(_lib.lib().dt_flowhand(None, ptr(fdr, c_u8p), ptr(river, c_i8p), H, W, float(px),
                                 ptr(fd, c_f32p), ptr(idx, c_i64p), None))
    r, c = np.divmod(idx, W)
    g = np.where(idx == -100, -100, (row_start + r) * matrix_columns + col_start + c)
    return fd, g.astype(np.float64)


def flow_distance_indexes_sequential(flow_direction, river_matrix, px):
    """Name kept for importers of flowhand.py:8; HIP path, normative kernel semantics."""
    fd, idx = flow_distance_index_cpu(None, flow_direction, river_matrix, px, None, None, np.zeros(4),
                                      0, 0, np.asarray(flow_direction).shape[1])
    return fd, idx.astype(np.int64)


def fdist_indexes_sequential_jit(fdr, river, px, fdist=None):
    """flowhand.py:128-239 name; the separator pre-solve it exists for is not needed (one tile)."""
    return flow_distance_indexes_sequential(fdr, river, px)
