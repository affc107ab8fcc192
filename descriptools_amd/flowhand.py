"""Flow distance to the nearest drainage, drained-to river index, HAND -- HIP replacement of
descriptools/flowhand.py (the reference's per-cell pointer chase becomes pointer doubling)."""
import numpy as np

from . import _lib
from ._lib import c_f32p, c_f64p, c_i8p, c_i64p, c_u8p, check, heights, ptr
from .device import host_empty


def _hand_dtype(dem):
    dt = np.asarray(dem).dtype
    return dt if dt.kind in "iuf" else np.dtype(np.float32)


def flow_hand_index(dem_raster, flow_direction_matrix, river_matrix, px, division_column=0,
                    division_row=0):
    """flowhand.py:242-411 -> (flow_distance float32, indices int64, hand in the DEM's dtype).
    One tile (division_* accepted, ignored: tiled == untiled is the reference's contract)."""
    dem, wide = heights(dem_raster)
    fdr = np.ascontiguousarray(flow_direction_matrix, np.uint8)
    river = np.ascontiguousarray(river_matrix, np.int8)
    H, W = fdr.shape
    fd = host_empty((H, W), np.float32)
    idx = host_empty((H, W), np.int64)
    ht = _hand_dtype(dem_raster)
    if wide:
        # heights that float32 cannot hold: flow distance and river index do not read them; HAND = dem - dem[idx] in
        # float64 (flowhand.py:436-438 in the DEM's own dtype)
        check(_lib.lib().dt_flowhand(None, ptr(fdr, c_u8p), ptr(river, c_i8p), H, W, float(px),
                                     ptr(fd, c_f32p), ptr(idx, c_i64p), None))
        hand = host_empty((H, W), np.float64)
        check(_lib.lib().dt_hand_f64(ptr(dem, c_f64p), ptr(idx, c_i64p), dem.size, ptr(hand, c_f64p)))
        return fd, idx, (hand if ht == np.float64 else hand.astype(ht))
    hand = host_empty((H, W), np.float32)
    check(_lib.lib().dt_flowhand(ptr(dem, c_f32p), ptr(fdr, c_u8p), ptr(river, c_i8p), H, W, float(px),
                                 ptr(fd, c_f32p), ptr(idx, c_i64p), ptr(hand, c_f32p)))
    return fd, idx, (hand if ht == np.float32 else hand.astype(ht))


def hand_calculator(dem, indices):
    """flowhand.py:414-442."""
    d, wide = heights(dem)
    idx = np.ascontiguousarray(indices, np.int64)
    ht = _hand_dtype(dem)
    if wide:
        hand = host_empty(d.shape, np.float64)
        check(_lib.lib().dt_hand_f64(ptr(d, c_f64p), ptr(idx, c_i64p), d.size, ptr(hand, c_f64p)))
        return hand if ht == np.float64 else hand.astype(ht)
    hand = host_empty(d.shape, np.float32)
    check(_lib.lib().dt_hand_f32(ptr(d, c_f32p), ptr(idx, c_i64p), d.size, ptr(hand, c_f32p)))
    return hand if ht == np.float32 else hand.astype(ht)


def index_calculator(river_indices, row_start, column_start, column_size):
    """flowhand.py:445-472 (unused by the reference itself): flat indices local to a tile -> flat indices of
    the whole raster, float64 like the reference's expression; -100 stays -100.  Host index arithmetic."""
    local = np.asarray(river_indices)
    r, c = np.divmod(local.astype(np.float64), float(local.shape[1]))
    whole = (r + row_start) * column_size + (c + column_start)
    whole[local == -100] = -100
    return whole


def _ring_payload(shape, boundary_distance, boundary_index, out):
    """The separator values a tile is handed (flowhand.py:313-390) laid out on the one-cell ring around it:
    (dist, index, usable) arrays of the extended shape (H + 2, W + 2).  Vector k = 0 top, 1 left, 2 right,
    3 bottom; a vector starts one cell early when the tile has a neighbour on the perpendicular low side
    (`irow += 1`, flowhand.py:632-633, 678-679, 726-727, 769-770).  Corners: the two top ones come from the
    top vector, the bottom ones from the left / right vectors (the order of the kernel's tests)."""
    H, W = shape
    up, left, right, down = (int(v) == 1 for v in np.asarray(out).reshape(-1)[:4])
    bd = np.asarray(boundary_distance, np.float64)
    bi = np.asarray(boundary_index, np.float64)
    dist = np.full((H + 2, W + 2), -100.0)
    index = np.full((H + 2, W + 2), -100.0)
    usable = np.zeros((H + 2, W + 2), bool)

    def put(ys, xs, k, first):
        n = len(ys)
        pos = first + np.arange(n)
        dist[ys, xs], index[ys, xs] = bd[k][pos], bi[k][pos]
        usable[ys, xs] = True

    if up:
        x0, x1 = (0 if left else 1), (W + 2 if right else W + 1)  # extended columns of the top ring row
        put(np.zeros(x1 - x0, int), np.arange(x0, x1), 0, 0)
    if left:
        y1 = H + 2 if down else H + 1
        put(np.arange(1, y1), np.zeros(y1 - 1, int), 1, 1 if up else 0)
    if right:
        y1 = H + 2 if down else H + 1
        put(np.arange(1, y1), np.full(y1 - 1, W + 1), 2, 1 if up else 0)
    if down:
        put(np.full(W, H + 1), np.arange(1, W + 1), 3, 1 if left else 0)
    usable &= dist != -100  # a separator cell without a drainage path ends the walk (flowhand.py:646, 662, ...)
    return dist, index, usable


def flow_distance_index_cpu(dem, flow_direction, river_matrix, px, boundary_distance, boundary_index,
                            out, row_start, col_start, matrix_columns, blocks=0, threads=0):
    """flowhand.py:476-562 + the kernel's tile-exit rules (:622-797): flow distance and GLOBAL flat river index
    (row_start + r) * matrix_columns + col_start + c of a tile whose paths may leave through the sides that have a
    neighbouring tile (out[k] == 1), where they pick up the pre-solved separator value (distance added, index
    taken over).  The separator ring becomes a ring of terminal cells around the tile and the whole thing is ONE
    dt_flowhand call.  Returns (float32 distances, float64 indices) like the reference.  A path that leaves after
    exactly 20000 in-tile moves is cut here and not in the reference (the exit step counts towards the cap)."""
    fdr = np.ascontiguousarray(flow_direction, np.uint8)
    river = np.ascontiguousarray(river_matrix, np.int8)
    H, W = fdr.shape
    has_ring = out is not None and np.any(np.asarray(out) != 0)
    if has_ring:
        rd, ri, ok = _ring_payload((H, W), boundary_distance, boundary_index, out)
        fe = np.zeros((H + 2, W + 2), np.uint8)
        re = np.zeros((H + 2, W + 2), np.int8)
        fe[1:-1, 1:-1], re[1:-1, 1:-1] = fdr, river
        fe[ok], re[ok] = 1, 1  # terminal: "river" cells of the extended raster; the rest of the ring has fdr 0
        fdr, river = fe, re
    h, w = fdr.shape
    fd = np.empty((h, w), np.float32)
    idx = np.empty((h, w), np.int64)
    check(_lib.lib().dt_flowhand(None, ptr(fdr, c_u8p), ptr(river, c_i8p), h, w, float(px),
                                 ptr(fd, c_f32p), ptr(idx, c_i64p), None))
    if has_ring:
        fd, idx = fd[1:-1, 1:-1], idx[1:-1, 1:-1]
    r, c = np.divmod(idx, w)
    if has_ring:
        on_ring = (idx != -100) & ((r == 0) | (r == h - 1) | (c == 0) | (c == w - 1))
        r, c = r - 1, c - 1
    g = ((row_start + r) * matrix_columns + col_start + c).astype(np.float64)
    g[idx == -100] = -100
    if has_ring:
        ty, tx = np.divmod(idx[on_ring], w)
        fd = fd.copy()
        fd[on_ring] = (fd[on_ring].astype(np.float64) + rd[ty, tx]).astype(np.float32)
        g[on_ring] = ri[ty, tx]
    return np.ascontiguousarray(fd), g


def flow_distance_indexes_sequential(flow_direction, river_matrix, px):
    """Name kept for importers of flowhand.py:8; HIP path, normative kernel semantics."""
    fd, idx = flow_distance_index_cpu(None, flow_direction, river_matrix, px, None, None, None,
                                      0, 0, np.asarray(flow_direction).shape[1])
    return fd, idx.astype(np.int64)


def fdist_indexes_sequential_jit(fdr, river, px, fdist=None):
    """flowhand.py:128-239: the separator pre-solve of the reference's host tiling.  Without `fdist` every cell
    is solved; with it only the cells marked -50 are (the separator lines, flowhand.py:283-286), the others keep
    their values and index 0.  Returns (float32 distances, int32 indices).  Runs the HIP kernels: their move cap
    is 20000 with the cycle test, where this twin of the reference stops at 5000 moves."""
    fd, idx = flow_distance_indexes_sequential(fdr, river, px)
    idx = idx.astype(np.int32)
    if fdist is None or np.size(fdist) == 0:
        return fd, idx
    todo = np.asarray(fdist) == -50
    fdist[todo] = fd[todo]
    return fdist, np.where(todo, idx, 0).astype(np.int32)
