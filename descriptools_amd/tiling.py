"""Multi-GPU tiling of the descriptor chain: one process per GPU, one core tile per rank.

The reference "tiles" on the host to fit a small GPU (division_* arguments) and stitches with CPU
pre-solves / repairs (flowhand.py:282-286, downslope.py:373-374).  Here a large DEM is tiled across
the GPUs of a node; what crosses a tile border is exchanged once per descriptor:

  slope / D8      a halo of the DEM (point-to-point with the <= 8 neighbouring ranks; synthetic DEMs
                  generate their halo locally)
  flow accum.     phase 1 per rank (dt_dev_flowacc_local_w) -> ALL-GATHER of one summary row per cell of
                  the core ring -> every rank solves the small rank-level graph redundantly
                  (solve_flowacc) -> phase 2 injects the inflow from other ranks
  HAND            same shape: phase 1 -> all-gather ring summaries -> solve_flowhand (pointer doubling
                  over ring entries) -> phase 2 with the resolved river cell (global index, height,
                  accumulation carried as payload: no remote gathers)
  downslope       walks are short: a 64-cell halo of dem + fdr; a walk that leaves it is counted and
                  reported (none on the benchmark DEMs)
  TI/MTI, GFI ... local

Messages are tiny (a ring row is 13-25 bytes per border cell), so the exchanges are latency-bound:
one all-gather each, no ring all-reduce.  The rank-level solves are plain numpy on every rank and are
shared with the single-process simulation (`simulate`) that proves tiled == untiled on one GPU, and
with the gloo CPU tests.
"""
import ctypes as C

import numpy as np

FA_CYCLE = np.uint64(1 << 63)
CAP = 20000
K_RIVER, K_DEAD, K_REXIT = 1, 2, 4
HALO = 64
WALKER_WORDS, WALKER_BYTES = 12, 48   # a walker record (include/descriptools_hip.h, dt_dev_downslope_emit_w)
W_SEQ, W_DONE = 1, 2

_DY = {1: 0, 2: 1, 4: 1, 8: 1, 16: 0, 32: -1, 64: -1, 128: -1}
_DX = {1: 1, 2: 1, 4: 0, 8: -1, 16: -1, 32: -1, 64: 0, 128: 1}
_DY_LUT = np.zeros(256, np.int64)
_DX_LUT = np.zeros(256, np.int64)
for _c in _DY:
    _DY_LUT[_c], _DX_LUT[_c] = _DY[_c], _DX[_c]


# ---------------------------------------------------------------------------------------------------
# geometry
# ---------------------------------------------------------------------------------------------------
class Layout:
    """ty x tx grid of rank tiles; rank r owns rows [ys[r//tx], ys[r//tx+1]) x cols [xs[r%tx], ...)."""

    def __init__(self, heights, widths):
        self.heights, self.widths = [int(h) for h in heights], [int(w) for w in widths]
        self.ty, self.tx = len(self.heights), len(self.widths)
        self.ys = np.concatenate([[0], np.cumsum(self.heights)]).astype(np.int64)
        self.xs = np.concatenate([[0], np.cumsum(self.widths)]).astype(np.int64)
        self.Hg, self.Wg = int(self.ys[-1]), int(self.xs[-1])
        self.size = self.ty * self.tx
        # rank borders must fall on the 64-cell tile grid of each rank (the in-LDS tiles are anchored
        # at the core origin); only the last row / column of ranks may be ragged
        assert all(h % 64 == 0 for h in self.heights[:-1]) and all(w % 64 == 0 for w in self.widths[:-1]), \
            "rank tile heights / widths must be multiples of 64 (except the last row / column)"
        # a rank tile must stay below 2^31 cells (32-bit in-rank indices).  A GLOBAL raster beyond 2^31 cells is
        # allowed: its ranks keep the flow accumulation as int64 rasters (RankTile(acc64=...), the reference's own
        # dtype) -- with int32 rasters a basin that reaches 2^31 cells is detected at run time
        # (DT_STATUS_ACC_OVERFLOW, RankTile.check_status)
        assert all(h * w < 2 ** 31 for h in self.heights for w in self.widths), "a rank tile must have < 2^31 cells"

    @staticmethod
    def uniform(world, H, W):
        """the most nearly square ty x tx grid with ty * tx == world and ty <= tx (1 x 2, 2 x 2, 2 x 3, 2 x 4, ...)"""
        ty = max(d for d in range(1, int(world ** 0.5) + 1) if world % d == 0)
        return Layout([H] * ty, [W] * (world // ty))

    def origin(self, r):
        return int(self.ys[r // self.tx]), int(self.xs[r % self.tx])

    def shape(self, r):
        return self.heights[r // self.tx], self.widths[r % self.tx]

    def owner(self, gy, gx):
        """rank owning global cells (vectorised)."""
        ry = np.searchsorted(self.ys, gy, side="right") - 1
        rx = np.searchsorted(self.xs, gx, side="right") - 1
        return ry * self.tx + rx


def perim_count(H, W):
    if H <= 0 or W <= 0:
        return 0
    if H == 1:
        return W
    if W == 1:
        return H
    return 2 * W + 2 * (H - 2)


def ring_coords(H, W):
    """core-local (y, x) of the ring cells in the library's order (dt_perim_cell)."""
    if H == 1:
        return np.zeros(W, np.int64), np.arange(W, dtype=np.int64)
    if W == 1:
        return np.arange(H, dtype=np.int64), np.zeros(H, np.int64)
    y = np.concatenate([np.zeros(W), np.full(W, H - 1), np.arange(1, H - 1), np.arange(1, H - 1)])
    x = np.concatenate([np.arange(W), np.arange(W), np.zeros(H - 2), np.full(H - 2, W - 1)])
    return y.astype(np.int64), x.astype(np.int64)


def ring_index(H, W, y, x):
    """inverse of ring_coords (vectorised); -1 for interior cells."""
    y, x = np.asarray(y, np.int64), np.asarray(x, np.int64)
    if H == 1:
        return x.copy()
    if W == 1:
        return y.copy()
    out = np.full(y.shape, -1, np.int64)
    m = x == W - 1
    out[m] = 2 * W + (H - 2) + (y[m] - 1)
    m = x == 0
    out[m] = 2 * W + (y[m] - 1)
    m = y == H - 1
    out[m] = W + x[m]
    m = y == 0
    out[m] = x[m]
    return out


def _exit_targets(layout, r, codes):
    """for ring cells of rank r whose D8 step (codes, 0 = none) leaves the core into another rank:
    (ring indices i, owner rank, ring index in the owner)."""
    H, W = layout.shape(r)
    y0, x0 = layout.origin(r)
    ys, xs = ring_coords(H, W)
    codes = np.asarray(codes, np.uint8)
    ty, tx = ys + _DY_LUT[codes], xs + _DX_LUT[codes]
    leaves = (codes != 0) & ((ty < 0) | (ty >= H) | (tx < 0) | (tx >= W))
    gy, gx = y0 + ty, x0 + tx
    leaves &= (gy >= 0) & (gy < layout.Hg) & (gx >= 0) & (gx < layout.Wg)
    i = np.nonzero(leaves)[0]
    own = layout.owner(gy[i], gx[i])
    tgt = np.empty(len(i), np.int64)
    for o in np.unique(own):
        m = own == o
        oy0, ox0 = layout.origin(int(o))
        oh, ow = layout.shape(int(o))
        tgt[m] = ring_index(oh, ow, gy[i][m] - oy0, gx[i][m] - ox0)
    assert (tgt >= 0).all()
    return i, own, tgt


# ---------------------------------------------------------------------------------------------------
# rank-level solves (numpy, identical on every rank)
# ---------------------------------------------------------------------------------------------------
def solve_flowacc(layout, summaries, max_iter=4096):
    """summaries[r] = (A int64[P_r], xr int32[P_r], code uint8[P_r]) from dt_dev_flowacc_local_w.
    Returns ext[r] uint64[P_r]: inflow arriving at each ring cell from other ranks (bit 63 = fed by
    a D8 cycle spanning ranks)."""
    P = [len(s[0]) for s in summaries]
    offs = np.concatenate([[0], np.cumsum(P)]).astype(np.int64)
    n = int(offs[-1])
    a_local = np.zeros(n, np.int64)
    xr_node = np.full(n, -1, np.int64)
    ex_nodes, ex_tgt = [], []
    for r, (A, xr, code) in enumerate(summaries):
        a_local[offs[r]:offs[r + 1]] = A
        xr = np.asarray(xr, np.int64)
        xr_node[offs[r]:offs[r + 1]] = np.where(xr >= 0, offs[r] + xr, -1)
        i, own, tgt = _exit_targets(layout, r, code)
        ex_nodes.append(offs[r] + i)
        ex_tgt.append(offs[own] + tgt)
    E = np.concatenate(ex_nodes) if ex_nodes else np.zeros(0, np.int64)
    T = np.concatenate(ex_tgt) if ex_tgt else np.zeros(0, np.int64)
    has = np.nonzero(xr_node >= 0)[0]
    total = a_local.copy()
    changed = np.zeros(n, bool)
    ext = np.zeros(n, np.int64)
    for _ in range(max_iter):
        # integer scatter-adds via bincount (float64 weights are exact below 2^53)
        ext = np.bincount(T, weights=total[E].astype(np.float64), minlength=n).astype(np.int64)
        recv = np.bincount(xr_node[has], weights=ext[has].astype(np.float64), minlength=n).astype(np.int64)
        new = a_local + recv
        changed = new != total
        total = new
        if not changed.any():
            break
    out = ext.astype(np.uint64)
    if changed.any():  # exits whose total never settles sit on a cycle spanning ranks
        cyc = changed[E]
        out[T[cyc]] = out[T[cyc]] | FA_CYCLE
    return [out[offs[r]:offs[r + 1]].copy() for r in range(len(summaries))]


def solve_flowhand(layout, summaries, ring_codes):
    """summaries[r] = (kind u8, ref i32, nc i32, nd i32, zr f32, ar i64)[P_r] from
    dt_dev_flowhand_local_w; ring_codes[r] = D8 codes of rank r's ring cells.  Returns per rank
    (res_ok u8, res_nc i32, res_nd i32, gidx i64, zr f32, ar i64)[P_r] for dt_dev_flowhand_finish_w."""
    P = [len(s[0]) for s in summaries]
    offs = np.concatenate([[0], np.cumsum(P)]).astype(np.int64)
    n = int(offs[-1])
    kind = np.concatenate([np.asarray(s[0], np.uint8) for s in summaries]) if n else np.zeros(0, np.uint8)
    nc = np.concatenate([np.asarray(s[2], np.int64) for s in summaries]) if n else np.zeros(0, np.int64)
    nd = np.concatenate([np.asarray(s[3], np.int64) for s in summaries]) if n else np.zeros(0, np.int64)
    zr = np.concatenate([np.asarray(s[4], np.float32) for s in summaries]) if n else np.zeros(0, np.float32)
    ar = np.concatenate([np.asarray(s[5], np.int64) for s in summaries]) if n else np.zeros(0, np.int64)
    gidx = np.full(n, -100, np.int64)
    step_tgt = np.full(n, -1, np.int64)  # entry node a ring cell's own D8 step lands on (other rank)
    for r, s in enumerate(summaries):
        ref = np.asarray(s[1], np.int64)
        y0, x0 = layout.origin(r)
        H, W = layout.shape(r)
        riv = np.asarray(s[0]) == K_RIVER
        g = np.full(P[r], -100, np.int64)
        g[riv] = (y0 + ref[riv] // W) * layout.Wg + x0 + ref[riv] % W
        gidx[offs[r]:offs[r + 1]] = g
        i, own, tgt = _exit_targets(layout, r, ring_codes[r])
        step_tgt[offs[r] + i] = offs[own] + tgt
    # entry e of kind REXIT continues at the entry its exit ring cell `ref` steps onto
    ptr = np.full(n, -1, np.int64)
    for r, s in enumerate(summaries):
        ref = np.asarray(s[1], np.int64)
        rex = np.nonzero(np.asarray(s[0]) == K_REXIT)[0]
        ptr[offs[r] + rex] = step_tgt[offs[r] + ref[rex]]
    done = (kind == K_RIVER) | (kind == K_DEAD) | (ptr < 0)
    dead = ~(kind == K_RIVER) & done
    term = np.arange(n, dtype=np.int64)
    for _ in range(17):  # pointer doubling, Jacobi; 2^16 rank crossings >> cap of 20000 moves
        act = np.nonzero(~done)[0]
        if len(act) == 0:
            break
        t = ptr[act]
        nnc, nnd = nc[act] + nc[t], nd[act] + nd[t]
        over = nnc + nnd > CAP
        ndead = dead[t] & done[t] | over
        ndone = done[t] | over
        nterm, nptr = term[t], ptr[t]
        nc[act], nd[act] = np.where(over, 0, nnc), np.where(over, 0, nnd)
        dead[act], done[act], term[act], ptr[act] = ndead, ndone, nterm, nptr
    dead |= ~done
    ok_e = ~dead
    out = []
    for r in range(len(summaries)):
        sl = slice(int(offs[r]), int(offs[r + 1]))
        tgt = step_tgt[sl]
        has = tgt >= 0
        tt = np.where(has, tgt, 0)
        ok = has & ok_e[tt]
        tm = term[tt]
        out.append((ok.astype(np.uint8), np.where(ok, nc[tt], 0).astype(np.int32),
                    np.where(ok, nd[tt], 0).astype(np.int32), np.where(ok, gidx[tm], -100).astype(np.int64),
                    np.where(ok, zr[tm], -100).astype(np.float32), np.where(ok, ar[tm], 0).astype(np.int64)))
    return out


# ---------------------------------------------------------------------------------------------------
# one rank's tile on its GPU
# ---------------------------------------------------------------------------------------------------
# summary-row fields: (name, dtype, byte offset in units of pmax)
FA_FIELDS = (("A", "int64", 0), ("xr", "int32", 8), ("code", "uint8", 12), ("ring", "uint8", 13))
FA_ROW_BYTES = 16
FH_FIELDS = (("ref", "int32", 0), ("nc", "int32", 4), ("nd", "int32", 8), ("zr", "float32", 12),
             ("ar", "int64", 16), ("kind", "uint8", 24), ("ring", "uint8", 25))
FH_ROW_BYTES = 32


class RankTile:
    """Extended rasters ((H + 2*HALO) x (W + 2*HALO)) of one rank and the windowed library calls."""

    def __init__(self, layout, rank, device=0, stream=None, px=10.0, n_top=0.1, n_gfi=0.4, b=0.1, dz=5.0,
                 river_threshold=None, halo=HALO, idx64=None, acc64=None, rasters=None, tune_placement=True,
                 long_walks=False, emit_walkers=None):
        """acc64: flow accumulation (and the river accumulation payload) as int64 rasters, the reference's dtype --
        the default for a global raster of more than 2^31 cells, where a basin can exceed 32 bits; int32 rasters
        otherwise (exact: an accumulation is at most cells - 1; half the bytes).  rasters: names of the rasters to allocate (default: all).
        tune_placement: True -- hand the 4-byte rasters this tile allocates anyway to their roles by measured
        write-conflict class (placement.py; only rasters of >= 64 MiB are measured); "search" -- also look for blocks
        of other classes when those are all alike (bounded transient allocations on THIS rank's device, seconds of
        set-up: bench.py); False -- allocation order.
        emit_walkers: downslope emits the walks that leave this rank's memory as walker records on the device (48 bytes
        each, room for one per 64 core cells; finish_downslope sends them on from where they are) -- default: with
        long_walks, i.e. on real terrain; without it such walks are found by their -50 marks and start again.
        long_walks: downslope with the long-walk workspace (dt_dev_downslope_lift_w: 8 bytes per core cell + 24 per
        cell of core + halo, allocated at the first downslope call) -- for real terrain, where flats and valley floors
        make walks thousands of moves long; same results."""
        import torch
        from . import _lib
        from .device import Context
        self.torch, self._lib, self.L = torch, _lib, _lib.lib()
        self.layout, self.rank, self.halo = layout, rank, halo
        self.long_walks, self._lift_work = bool(long_walks), None
        self.emit_walkers = bool(long_walks) if emit_walkers is None else bool(emit_walkers)
        self._walkers = None
        self.H, self.W = layout.shape(rank)
        self.gy0, self.gx0 = layout.origin(rank)
        self.He, self.We = self.H + 2 * halo, self.W + 2 * halo
        self.dev = torch.device("cuda", device)
        self.ctx = Context(device=device, stream=stream)
        self.side_ctx = Context(device=device)  # the downslope branch (own stream), see run_rank
        # every torch op of this tile (fills, copies, the ring gather, the all-gathers' events) is issued on the
        # CONTEXT's stream, wrapped as a torch stream: library kernels and torch ops are ordered by the stream
        # itself, whatever torch's current stream is
        self.ts = torch.cuda.ExternalStream(int(self.ctx.stream), device=self.dev)
        self.px, self.n_top, self.n_gfi, self.b, self.dz = px, n_top, n_gfi, b, dz
        self.river_threshold = (layout.Hg * layout.Wg) // 512 if river_threshold is None else int(river_threshold)
        self.P = perim_count(self.H, self.W)
        # the river index is the GLOBAL raster's flat index: int32 while that raster has <= 2^31 cells (eight
        # ranks of 16384^2), int64 beyond -- the widest output raster of the step
        if idx64 is None:
            idx64 = layout.Hg * layout.Wg > 2 ** 31
        assert idx64 or layout.Hg * layout.Wg <= 2 ** 31
        self.idx_dtype = torch.int64 if idx64 else torch.int32
        if acc64 is None:
            # an accumulation counts the upstream cells without the cell itself: <= cells - 1, which fits int32 up to
            # and including a raster of exactly 2^31 cells (eight ranks of 16384^2)
            acc64 = layout.Hg * layout.Wg > 2 ** 31
        self.acc64 = bool(acc64)
        self.acc_dtype = torch.int64 if self.acc64 else torch.int32
        sfx = "_a64" if self.acc64 else ""
        L = self.L
        self._f_fa_finish = getattr(L, "dt_dev_flowacc_finish_w" + sfx)
        self._f_fa_finish_fh_local = getattr(L, "dt_dev_flowacc_finish_flowhand_local_w" + sfx)
        self._f_slope_twi = getattr(L, "dt_dev_slope_twi_w" + sfx)
        self._f_fh_local = getattr(L, "dt_dev_flowhand_local_w" + sfx)
        self._f_fh_finish = getattr(L, "dt_dev_flowhand_finish_w" + sfx)
        self._f_fh_gfi_finish = getattr(L, "dt_dev_flowhand_gfi_finish_w" + sfx)
        self._f_gfi_lnhlh = getattr(L, "dt_dev_gfi_lnhlh" + sfx)
        t = {}
        self._on_ts = torch.cuda.stream(self.ts)
        self._on_ts.__enter__()
        for name, dt in (("dem", torch.float32), ("fdr", torch.uint8), ("fac", self.acc_dtype),
                         ("river", torch.int8), ("fdist", torch.float32), ("idx", self.idx_dtype),
                         ("hand", torch.float32), ("a_river", self.acc_dtype), ("slope", torch.float32),
                         ("ti", torch.float32), ("mti", torch.float32), ("gfi", torch.float32),
                         ("lnhlh", torch.float32), ("down", torch.float32)):
            if rasters is None or name in rasters:
                t[name] = torch.zeros((self.He, self.We), dtype=dt, device=self.dev)
        self.t = t
        assert tune_placement in (False, True, "search")
        self.placement = {"tuned": False, "why": "tune_placement=False"}
        if tune_placement:
            self._tune_placement(search=tune_placement == "search")
        self.n_unres = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.win = _lib.Window(self.H, self.W, self.We, self.gy0, self.gx0, layout.Hg, layout.Wg, halo)
        ys, xs = ring_coords(self.H, self.W)
        self._ring_lin = torch.as_tensor((ys + halo) * self.We + xs + halo, device=self.dev)
        # summary rows, laid out for the all-gather and for the rank-level solves on the GPU: every field
        # has pmax entries (pmax = the largest ring of any rank, rounded up to 8 so fields stay aligned)
        pm = max(perim_count(*layout.shape(r)) for r in range(layout.size))
        self.pmax = pm = (pm + 7) // 8 * 8
        self.fa_row = torch.zeros(FA_ROW_BYTES * pm, dtype=torch.uint8, device=self.dev)
        self.fh_row = torch.zeros(FH_ROW_BYTES * pm, dtype=torch.uint8, device=self.dev)
        self._fa_v = self._row_views(self.fa_row, FA_FIELDS)
        self._fh_v = self._row_views(self.fh_row, FH_FIELDS)
        self._fa_offs = (C.c_int64 * 3)(*[o * pm for _, _, o in FA_FIELDS[:3]])
        self._fh_offs = (C.c_int64 * 7)(*[o * pm for _, _, o in FH_FIELDS])
        self._heights = (C.c_int64 * layout.ty)(*layout.heights)
        self._widths = (C.c_int64 * layout.tx)(*layout.widths)
        self._ext = torch.zeros(max(self.P, 1), dtype=torch.int64, device=self.dev)
        self._res = (torch.zeros(max(self.P, 1), dtype=torch.uint8, device=self.dev),
                     torch.zeros(max(self.P, 1), dtype=torch.int32, device=self.dev),
                     torch.zeros(max(self.P, 1), dtype=torch.int32, device=self.dev),
                     torch.zeros(max(self.P, 1), dtype=torch.int64, device=self.dev),
                     torch.zeros(max(self.P, 1), dtype=torch.float32, device=self.dev),
                     torch.zeros(max(self.P, 1), dtype=torch.int64, device=self.dev))
        self._on_ts.__exit__(None, None, None)
        self.ctx.sync()  # the buffers exist and are zero before anybody (any stream) touches them

    def _tune_placement(self, search=False):
        """placement.assign over this tile's 4-byte rasters (chain.Chain._tune_placement's counterpart): the rasters
        one kernel writes together -- slope / TI / MTI; fdist / idx / hand / GFI / ln(hl/H) -- must not all lie in
        one conflict class of the device's memory; further candidate blocks are tried when the first ones are alike"""
        from . import placement
        tc = self.torch
        four = [n for n, x in self.t.items() if x.element_size() == 4]
        if not four:
            return
        dts = {n: self.t[n].dtype for n in four}
        objs = {self.t[n].data_ptr(): self.t[n] for n in four}
        together = [[n for n in g if n in four] for g in (("slope", "ti", "mti"), ("fdist", "idx", "hand", "gfi", "lnhlh"))]
        groups = [g for g in together if g] + [[n] for n in four if not any(n in g for g in together)]

        def extra_alloc():
            x = tc.zeros((self.He, self.We), dtype=tc.float32, device=self.dev)
            objs[x.data_ptr()] = x
            return x.data_ptr()

        def extra_release(q):
            objs.pop(q)
        def spacer_alloc(nbytes):
            return tc.empty(int(nbytes), dtype=tc.uint8, device=self.dev)

        def spacer_release(x):
            del x
        self.ctx.sync()
        roles, info = placement.assign(self.ctx, self.He * self.We * 4, list(objs), groups, extra_alloc, extra_release,
                                       spacer_alloc=spacer_alloc, spacer_release=spacer_release, search=search)
        self.placement = info
        if roles is None:
            return
        for n, q in roles.items():
            self.t[n] = objs[q].view(dts[n])
            self.t[n].zero_()  # the measurement wrote into the blocks; the halos must read as zeros
        objs.clear()
        self.ctx.sync()
        tc.cuda.empty_cache()
        if info.get("spacer_GiB"):  # the runtime defers the release of the spacers: take the wait here (placement.assign)
            try:
                self.ctx.empty((2 << 30,), np.uint8).free()
            except MemoryError:
                pass

    def on_stream(self):
        """context manager: torch ops inside run on this tile's context stream"""
        return self.torch.cuda.stream(self.ts)

    def _row_views(self, row, fields):
        tc, pm = self.torch, self.pmax
        v = {}
        for name, dt, off in fields:
            tdt = getattr(tc, dt)
            es = tc.empty(0, dtype=tdt).element_size()
            v[name] = row[off * pm:(off + es) * pm].view(tdt)[:self.P]
        return v

    # pointer of raster `name` at the core origin
    def p(self, name):
        t = self.t.get(name)
        if t is None:  # a raster this tile was built without (RankTile(rasters=...))
            return None
        return t.data_ptr() + (self.halo * self.We + self.halo) * t.element_size()

    def core(self, name):
        h = self.halo
        return self.t[name][h:h + self.H, h:h + self.W]

    def _chk(self, rc):
        self._lib.check(rc)

    # ---- DEM ---------------------------------------------------------------------------------
    def synth_dem(self, seed, nodata_pct=0):
        """generate the core AND its halo from the global generator (clipped to the global raster)."""
        h = self.halo
        y0, x0 = max(self.gy0 - h, 0), max(self.gx0 - h, 0)
        y1, x1 = min(self.gy0 + self.H + h, self.layout.Hg), min(self.gx0 + self.W + h, self.layout.Wg)
        with self.on_stream():
            tmp = self.torch.empty((y1 - y0, x1 - x0), dtype=self.torch.float32, device=self.dev)
            self._chk(self.L.dt_dev_synth_dem(self.ctx.h, seed, self.layout.Hg, self.layout.Wg, y0, x0, y1 - y0,
                                              x1 - x0, nodata_pct, tmp.data_ptr()))
            oy, ox = y0 - (self.gy0 - h), x0 - (self.gx0 - h)
            self.t["dem"][oy:oy + (y1 - y0), ox:ox + (x1 - x0)] = tmp
        self.ctx.sync()

    def set_dem_ext(self, dem_ext):
        """host array of the extended window (He x We); cells outside the global raster are ignored."""
        with self.on_stream():
            self.t["dem"].copy_(self.torch.as_tensor(np.ascontiguousarray(dem_ext, np.float32)))
        self.ctx.sync()

    # ---- local stages ------------------------------------------------------------------------------
    def d8(self):
        """D8 over the core and the halo minus its outermost ring (needed by HAND's 1-cell look-ahead
        and by downslope's margin)."""
        h = self.halo
        m = h - 1
        y0, x0 = max(self.gy0 - m, 0), max(self.gx0 - m, 0)
        y1, x1 = min(self.gy0 + self.H + m, self.layout.Hg), min(self.gx0 + self.W + m, self.layout.Wg)
        win = self._lib.Window(y1 - y0, x1 - x0, self.We, y0, x0, self.layout.Hg, self.layout.Wg, 1)
        off = ((y0 - (self.gy0 - h)) * self.We + (x0 - (self.gx0 - h)))
        dem = self.t["dem"].data_ptr() + off * 4
        fdr = self.t["fdr"].data_ptr() + off
        self._chk(self.L.dt_dev_slope_d8_w(self.ctx.h, C.byref(win), dem, self.px, None, fdr, None))

    # ---- hydrological conditioning over ranks: see condition_ranks ------------------------------
    def cond_alloc(self):
        """the filled surface, the flat distances (extended rasters like the others) and the iteration flag"""
        tc = self.torch
        if "filled" not in self.t:
            with self.on_stream():
                self.t["filled"] = tc.zeros((self.He, self.We), dtype=tc.float32, device=self.dev)
                self.t["dist"] = tc.zeros((self.He, self.We), dtype=tc.int32, device=self.dev)
                # one byte per cell: which neighbours have another filled height (what the flat stages read instead of
                # the surface: dt_dev_condition_stage_m_w)
                self.t["nsame"] = tc.zeros((self.He, self.We), dtype=tc.uint8, device=self.dev)
                self._cflag = tc.zeros(1, dtype=tc.int32, device=self.dev)
            self.ctx.sync()

    def cond_stage(self, stage, rounds=1):
        """dt_dev_condition_stage_m_w on the core window; stages 1 / 3 / 4 start from a zeroed flag"""
        if stage in (1, 3, 4):
            with self.on_stream():
                self._cflag.zero_()
        self._chk(self.L.dt_dev_condition_stage_m_w(self.ctx.h, C.byref(self.win), stage, rounds, self.p("dem"),
                                                    self.p("filled"), self.p("fdr"), self.p("dist"),
                                                    self._cflag.data_ptr(), self.p("nsame")))

    def cond_flag(self):
        self.ctx.sync()
        return int(self._cflag.item())

    def cond_d8(self):
        """D8 codes of the CORE from the filled surface (its halo comes from the neighbours' exchange); the codes of
        the halo are exchanged afterwards, not recomputed: they are the neighbours' conditioned codes"""
        self._chk(self.L.dt_dev_slope_d8_w(self.ctx.h, C.byref(self.win), self.p("filled"), self.px, None,
                                           self.p("fdr"), None))

    def ring_codes_dev(self):
        """D8 codes of the ring cells (device tensor, no synchronisation)."""
        return self.t["fdr"].reshape(-1)[self._ring_lin]

    def ring_codes(self):
        with self.on_stream():
            rc = self.ring_codes_dev()
        self.ctx.sync()
        return rc.cpu().numpy()

    def fa_local(self, sync=True):
        v = self._fa_v
        A, xr, code = v["A"], v["xr"], v["code"]
        self._chk(self.L.dt_dev_flowacc_local_w(self.ctx.h, C.byref(self.win), self.p("fdr"), self.p("fac"),
                                                A.data_ptr(), xr.data_ptr(), code.data_ptr()))
        if sync:
            self.ctx.sync()
        return A, xr, code

    def fa_finish(self, ext):
        tc = self.torch
        with self.on_stream():
            e = tc.as_tensor(ext.view(np.int64), device=self.dev) if ext is not None else None
        self._keep = e
        self._chk(self._f_fa_finish(self.ctx.h, C.byref(self.win), self.p("fdr"), self.p("dem"),
                                    e.data_ptr() if e is not None else None,
                                    self.river_threshold, self.p("fac"), self.p("river")))

    def fill_ring_codes(self):
        """D8 codes of the ring cells into both summary rows (the rank-level solves step across ranks with them)."""
        with self.on_stream():
            rc = self.ring_codes_dev()
            self._fa_v["ring"].copy_(rc)
            self._fh_v["ring"].copy_(rc)

    def fa_solve_finish(self, rows):
        """rank-level inflow solve on the GPU from the all-gathered rows (size x FA_ROW_BYTES*pmax bytes,
        device), then pass 3.  No host synchronisation."""
        self._keep_rows = rows
        self._chk(self.L.dt_dev_rank_solve_flowacc(self.ctx.h, self.layout.ty, self.layout.tx, self._heights,
                                                   self._widths, self.pmax, rows.data_ptr(),
                                                   FA_ROW_BYTES * self.pmax, self._fa_offs, self.rank, self.P,
                                                   self._ext.data_ptr()))
        self._chk(self._f_fa_finish(self.ctx.h, C.byref(self.win), self.p("fdr"), self.p("dem"),
                                    self._ext.data_ptr(), self.river_threshold, self.p("fac"), self.p("river")))

    def fa_solve_finish_fh_local(self, rows):
        """fa_solve_finish + fh_local in one call: the last accumulation tile pass and HAND's first share the tile's
        direction codes and the river mask (one kernel in the common form, dt_dev_flowacc_finish_flowhand_local_w)"""
        self._keep_rows = rows
        self._chk(self.L.dt_dev_rank_solve_flowacc(self.ctx.h, self.layout.ty, self.layout.tx, self._heights,
                                                   self._widths, self.pmax, rows.data_ptr(),
                                                   FA_ROW_BYTES * self.pmax, self._fa_offs, self.rank, self.P,
                                                   self._ext.data_ptr()))
        v = self._fh_v
        self._chk(self._f_fa_finish_fh_local(self.ctx.h, C.byref(self.win), self.p("fdr"), self.p("dem"),
                                             self._ext.data_ptr(), self.river_threshold, self.p("fac"),
                                             self.p("river"), v["kind"].data_ptr(), v["ref"].data_ptr(),
                                             v["nc"].data_ptr(), v["nd"].data_ptr(), v["zr"].data_ptr(),
                                             v["ar"].data_ptr()))

    def _idx_args(self):
        """(idx32, idx64) of the windowed HAND calls: the one of the two that matches the idx raster's dtype"""
        return (self.p("idx"), None) if self.idx_dtype == self.torch.int32 else (None, self.p("idx"))

    def free(self):
        """release the rasters and both contexts"""
        self.side_ctx.sync()
        self.ctx.sync()
        self.t, self._keep, self._keep2, self._keep_rows, self._keep_rows2 = {}, None, None, None, None
        self.fa_row = self.fh_row = self._ext = self._res = self._fa_v = self._fh_v = self._lift_work = None
        self._walkers = None
        self.side_ctx.close()
        self.ctx.close()

    def fh_solve_finish(self, rows, fuse_gfi=False, want_a_river=True):
        """rank-level HAND solve on the GPU, then pass 3; fuse_gfi: GFI and ln(hl/H) in the same pass (the
        river-accumulation raster is then only an optional by-product)."""
        self._keep_rows2 = rows
        r = self._res
        self._chk(self.L.dt_dev_rank_solve_flowhand(self.ctx.h, self.layout.ty, self.layout.tx, self._heights,
                                                    self._widths, self.pmax, rows.data_ptr(),
                                                    FH_ROW_BYTES * self.pmax, self._fh_offs, self.rank, self.P,
                                                    *[a.data_ptr() for a in r]))
        if fuse_gfi:
            self._chk(self._f_fh_gfi_finish(
                self.ctx.h, C.byref(self.win), self.p("dem"), self.p("fdr"), self.p("river"), self.p("fac"),
                self.px, self.n_gfi, self.b, *[a.data_ptr() for a in r], self.p("fdist"), *self._idx_args(),
                self.p("hand"), self.p("a_river") if want_a_river else None, self.p("gfi"), self.p("lnhlh")))
        else:
            self._chk(self._f_fh_finish(self.ctx.h, C.byref(self.win), self.p("dem"), self.p("fdr"),
                                        self.p("river"), self.p("fac"), self.px,
                                        *[a.data_ptr() for a in r], self.p("fdist"),
                                        *self._idx_args(), self.p("hand"), self.p("a_river")))

    def fh_local(self, sync=True):
        v = self._fh_v
        kind, ref, nc, nd, zr, ar = v["kind"], v["ref"], v["nc"], v["nd"], v["zr"], v["ar"]
        self._chk(self._f_fh_local(self.ctx.h, C.byref(self.win), self.p("dem"), self.p("fdr"),
                                   self.p("river"), self.p("fac"), kind.data_ptr(),
                                   ref.data_ptr(), nc.data_ptr(), nd.data_ptr(), zr.data_ptr(),
                                   ar.data_ptr()))
        if sync:
            self.ctx.sync()
        return kind, ref, nc, nd, zr, ar

    def fh_finish(self, res):
        tc = self.torch
        ptrs = [None] * 6
        if res is not None:
            with self.on_stream():
                self._keep2 = [tc.as_tensor(np.ascontiguousarray(a), device=self.dev) for a in res]
            ptrs = [a.data_ptr() for a in self._keep2]
        self._chk(self._f_fh_finish(self.ctx.h, C.byref(self.win), self.p("dem"), self.p("fdr"),
                                    self.p("river"), self.p("fac"), self.px, ptrs[0], ptrs[1],
                                    ptrs[2], ptrs[3], ptrs[4], ptrs[5], self.p("fdist"),
                                    *self._idx_args(), self.p("hand"), self.p("a_river")))

    def slope_twi(self):
        """fused slope + TI + MTI on the core window (needs fac)."""
        self._chk(self._f_slope_twi(self.ctx.h, C.byref(self.win), self.p("dem"), self.p("fac"), self.px,
                                    self.n_top, self.p("slope"), None, self.p("ti"), self.p("mti")))

    def gfi(self):
        """GFI + ln(hl/H) over the flat extended rasters (halo cells hold zeros and are never read back)."""
        t, n = self.t, self.He * self.We
        self._chk(self._f_gfi_lnhlh(self.ctx.h, t["hand"].data_ptr(), t["a_river"].data_ptr(),
                                    t["fac"].data_ptr(), n, self.n_gfi, self.b, self.px,
                                    t["gfi"].data_ptr(), t["lnhlh"].data_ptr()))

    def downslope(self, side=False):
        """downslope on the core window (needs only dem + fdr: independent of the exchanges).  side=True:
        as a second branch on the side context's stream, forked here (after D8) and joined by join_side()."""
        ctx = self.ctx
        if side:
            self.ctx.fork(self.side_ctx)
            ctx = self.side_ctx
        if self.long_walks and self._lift_work is None:
            nb = int(self.L.dt_downslope_lift_workspace_w(C.byref(self.win)))
            with self.on_stream():
                self._lift_work = self.torch.empty(nb, dtype=self.torch.uint8, device=self.dev)
            self.ctx.sync()  # (the side stream may be the one that uses it)
        if self.emit_walkers:
            if self._walkers is None:
                cap = max(self.H * self.W // 64, 65536)
                with self.on_stream():
                    self._walkers = self.torch.zeros(256 + WALKER_BYTES * cap, dtype=self.torch.uint8, device=self.dev)
                self.ctx.sync()
            work = self._lift_work
            self._chk(self.L.dt_dev_downslope_emit_w(ctx.h, C.byref(self.win), self.p("dem"), self.p("fdr"), self.px,
                                                     self.dz, 0, self.p("down"), self.n_unres.data_ptr(),
                                                     work.data_ptr() if work is not None else None,
                                                     work.numel() if work is not None else 0,
                                                     self._walkers.data_ptr(), self._walkers.numel()))
            return
        if self.long_walks:
            self._chk(self.L.dt_dev_downslope_lift_w(ctx.h, C.byref(self.win), self.p("dem"), self.p("fdr"), self.px,
                                                     self.dz, 0, self.p("down"), self.n_unres.data_ptr(),
                                                     self._lift_work.data_ptr(), self._lift_work.numel()))
            return
        self._chk(self.L.dt_dev_downslope_w(ctx.h, C.byref(self.win), self.p("dem"), self.p("fdr"), self.px,
                                            self.dz, 0, self.p("down"), self.n_unres.data_ptr()))

    def join_side(self):
        self.ctx.join(self.side_ctx)

    def pointwise(self):
        self.slope_twi()
        self.gfi()
        self.downslope()

    def check_status(self):
        """raise if a kernel of this tile's steps flagged a condition that invalidates its rasters (a flow
        accumulation of >= 2^31 cells); synchronises"""
        self.ctx.raise_on_status()

    def unresolved_downslope(self):
        self.side_ctx.sync()
        self.ctx.sync()
        return int(self.n_unres.item())

    def host(self, name):
        self.side_ctx.sync()
        self.ctx.sync()
        return self.core(name).cpu().numpy()


# ---------------------------------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------------------------------
def _pad_rows(arrs, pmax):
    """concatenate 1-D tensors of different dtypes into one byte row of fixed length (for all_gather)."""
    import torch
    rows = []
    for a in arrs:
        b = a.contiguous().view(torch.uint8).reshape(-1)
        pad = pmax * a.element_size() - b.numel()
        rows.append(torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)]) if pad else b)
    return torch.cat(rows)


def all_gather_summaries(arrs, layout, rank, group=None):
    """all-gather a tuple of per-ring-cell tensors (one row per ring cell) across ranks with
    torch.distributed (RCCL on GPUs, gloo on CPU).  Returns summaries[r] = tuple of numpy arrays."""
    import torch
    import torch.distributed as dist
    P = [perim_count(*layout.shape(r)) for r in range(layout.size)]
    pmax = max(P)
    row = _pad_rows(arrs, pmax)
    if dist.get_backend(group) == "gloo":  # CPU rehearsal of the RCCL path
        row = row.cpu()
    out = torch.empty(layout.size * row.numel(), dtype=torch.uint8, device=row.device)
    dist.all_gather_into_tensor(out, row, group=group)
    host = out.cpu().numpy().reshape(layout.size, row.numel())
    res = []
    for r in range(layout.size):
        o, items = 0, []
        for a in arrs:
            es = a.element_size()
            np_dt = np.dtype(str(a.dtype).replace("torch.", ""))
            items.append(host[r, o:o + P[r] * es].copy().view(np_dt))
            o += pmax * es
        res.append(tuple(items))
    return res


_DIRS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def _halo_slices(H, W, h):
    """(rows, cols)(d, send): slices of an extended raster for direction component d in {-1, 0, 1}: the strip of
    the core that is SENT towards d, or the part of the halo that is RECEIVED from d."""
    def rows(d, send):
        if d == 0:
            return slice(h, h + H)
        if send:
            return slice(h, 2 * h) if d < 0 else slice(H, H + h)
        return slice(0, h) if d < 0 else slice(h + H, 2 * h + H)

    def cols(d, send):
        if d == 0:
            return slice(h, h + W)
        if send:
            return slice(h, 2 * h) if d < 0 else slice(W, W + h)
        return slice(0, h) if d < 0 else slice(h + W, 2 * h + W)
    return rows, cols


def halo_plan(layout, rank):
    """[(k, dy, dx, peer)] for the <= 8 neighbouring ranks of `rank` (k = index into _DIRS; the peer's strip that
    lands in our (dy, dx) halo is the one it sends in direction 7 - k)."""
    ry, rx = rank // layout.tx, rank % layout.tx
    plan = []
    for k, (dy, dx) in enumerate(_DIRS):
        py, px = ry + dy, rx + dx
        if 0 <= py < layout.ty and 0 <= px < layout.tx:
            plan.append((k, dy, dx, py * layout.tx + px))
    return plan


def exchange_halos(exts, layout, proc_of=None, halo=HALO, group=None):
    """Point-to-point halo exchange of extended rasters ((H + 2*halo) x (W + 2*halo) torch tensors, CPU for gloo /
    GPU for RCCL) with the <= 8 neighbouring ranks: each rank sends the border strips of its core and receives its
    halo.  xGMI is point-to-point, the strips are KBs: eight small sends per rank, no collective.  Halo cells outside
    the global raster are left untouched.

    exts: {logical rank: its extended raster} -- the logical ranks that live in THIS process (one, for one process per
    GPU); proc_of[r]: the process (group rank) holding logical rank r (default: r itself).  A neighbour that lives in
    the same process is still served by an isend / irecv pair to oneself (torch allows it; RCCL turns it into a copy):
    that is how a 1-GPU box exercises this very code path over RCCL (tests/test_gpu_run_rank.py).  RCCL matches the
    sends and receives between two processes by ORDER (tags are a gloo notion), so both lists are posted in the order
    of the key (sending logical rank, direction of the send).  gloo has no connection to oneself: there a neighbour
    in the same process is served by a plain copy of the same strips."""
    import torch
    import torch.distributed as dist
    h = halo
    me = dist.get_rank(group)
    self_p2p = dist.get_backend(group) != "gloo"
    assert all(hh >= h for hh in layout.heights) and all(ww >= h for ww in layout.widths), \
        "rank tiles must be at least `halo` cells in both directions"
    if proc_of is None:
        proc_of = list(range(layout.size))
    sends, recvs, local = [], [], []
    for r in sorted(exts):
        ext = exts[r]
        H, W = layout.shape(r)
        assert tuple(ext.shape) == (H + 2 * h, W + 2 * h), (r, tuple(ext.shape), (H, W))
        rows, cols = _halo_slices(H, W, h)
        for k, dy, dx, peer in halo_plan(layout, r):
            if not self_p2p and int(proc_of[peer]) == me and peer in exts:
                pH, pW = layout.shape(peer)
                prow, pcol = _halo_slices(pH, pW, h)
                local.append((r, dy, dx, exts[peer][prow(-dy, True), pcol(-dx, True)].clone()))
                continue
            sbuf = ext[rows(dy, True), cols(dx, True)].contiguous()
            rbuf = torch.empty_like(ext[rows(dy, False), cols(dx, False)]).contiguous()
            # the peer sends towards us in the opposite direction: its key is (peer, 7 - k)
            sends.append(((r, k), dist.P2POp(dist.isend, sbuf, int(proc_of[peer]), group=group, tag=r * 8 + k)))
            recvs.append(((peer, 7 - k), dist.P2POp(dist.irecv, rbuf, int(proc_of[peer]), group=group,
                                                    tag=peer * 8 + 7 - k), r, dy, dx, rbuf))
    sends.sort(key=lambda e: e[0])
    recvs.sort(key=lambda e: e[0])
    ops = [e[1] for e in sends] + [e[1] for e in recvs]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for r, dy, dx, rbuf in [e[2:] for e in recvs] + local:
        H, W = layout.shape(r)
        rows, cols = _halo_slices(H, W, h)
        exts[r][rows(dy, False), cols(dx, False)] = rbuf
    return exts


def exchange_halo(ext, layout, rank, halo=HALO, group=None):
    """exchange_halos for one process per rank: `ext` is rank `rank`'s extended raster"""
    return exchange_halos({rank: ext}, layout, None, halo, group)[rank]


def exchange_halo_local(exts, layout, halo=HALO):
    """The same exchange between N LOGICAL ranks living in one process (one device): exts[r] is rank r's extended
    raster; every send / receive pair of exchange_halo becomes one device-to-device copy of the same strips
    (what simulate() is to the all-gathers: the slicing, packing and placement are the product's, only the
    transport differs)."""
    h = halo
    for r in range(layout.size):
        H, W = layout.shape(r)
        rows, cols = _halo_slices(H, W, h)
        for k, dy, dx, peer in halo_plan(layout, r):
            pH, pW = layout.shape(peer)
            prow, pcol = _halo_slices(pH, pW, h)
            # what the peer sends in direction (-dy, -dx) is what we receive from (dy, dx)
            exts[r][rows(dy, False), cols(dx, False)] = exts[peer][prow(-dy, True), pcol(-dx, True)]
    return exts


class Exchange:
    """The two all-gathers of a step, issued on a SIDE stream behind an event recorded on the tile's context
    stream, so that kernels queued on that stream after the summaries keep the GPU busy while the ring rows travel
    (RCCL).  The gathered rows stay on the device: the rank-level graphs are solved there (dt_dev_rank_solve_*), the
    context stream only waits on an event, the host never blocks."""

    def __init__(self, tile, layout, world, group=None):
        self.tile, self.layout, self.world, self.group = tile, layout, world, group
        tc = self.torch = tile.torch
        self.side = tc.cuda.Stream(device=tile.dev)
        n = layout.size
        with tile.on_stream():
            self.fa_all = tc.zeros(n * tile.fa_row.numel(), dtype=tc.uint8, device=tile.dev)
            self.fh_all = tc.zeros(n * tile.fh_row.numel(), dtype=tc.uint8, device=tile.dev)
        tile.ctx.sync()
        self._done = None

    def gather(self, row, out):
        """all-gather `row` into `out` behind the work queued so far on the tile's context stream (the kernels that
        write the row); the returned tensor is valid on that stream after wait().  Work queued on the context
        stream between gather() and wait() overlaps the transfer."""
        tc = self.torch
        if self.world == 1:
            return row
        import torch.distributed as dist
        ev = tc.cuda.Event()
        ev.record(self.tile.ts)
        if dist.get_backend(self.group) == "gloo":  # CPU rehearsal of the RCCL path
            ev.synchronize()
            h = tc.empty(out.numel(), dtype=tc.uint8)
            dist.all_gather_into_tensor(h, row.cpu(), group=self.group)
            with self.tile.on_stream():
                out.copy_(h)
            return out
        with tc.cuda.stream(self.side):
            self.side.wait_event(ev)
            dist.all_gather_into_tensor(out, row, group=self.group)
            done = tc.cuda.Event()
            done.record(self.side)
        self._done = done
        return out

    def wait(self):
        if self._done is not None:
            self.tile.ts.wait_event(self._done)
            self._done = None


class LocalExchange:
    """Exchange for N LOGICAL ranks living in one process (one device), driven in lock-step by run_ranks_local: a
    gather concatenates the ranks' rows, exactly what the all-gather delivers.  Same interface as Exchange, so the
    product's schedule (rank_ops, the stages bench.py times) runs unchanged on a 1-GPU box."""

    def __init__(self, tiles):
        self.tiles = tiles
        self.fa_all = self.fh_all = None  # (Exchange's receive buffers: a gather here returns a fresh concatenation)

    def gather(self, row, out):
        tc = self.tiles[0].torch
        which = "fa_row" if any(row is t.fa_row for t in self.tiles) else "fh_row"
        for t in self.tiles:
            t.ctx.sync()
        out = tc.cat([getattr(t, which) for t in self.tiles])
        tc.cuda.synchronize()
        return out

    def wait(self):
        pass


def run_ranks_local(tiles, layout, d8=True):
    """rank_ops() of every logical rank, stage by stage in lock-step (every rank's rows exist before anybody gathers)"""
    ex = LocalExchange(tiles)
    ops = [rank_ops(t, layout, ex, d8=d8) for t in tiles]
    for i in range(len(RANK_OPS)):
        for o in ops:
            o[i][1]()
    for t in tiles:
        t.ctx.sync()


# one rank's step as named stages in launch order: (name, algorithmic bytes per cell -- chain.OPS' definitions; 0 for
# the exchanges --, what it covers)
RANK_OPS = (
    ("d8", 5), ("flowacc_local", 0), ("flowacc_gather", 0), ("downslope", 9), ("flowacc_finish_flowhand_local", 8),
    ("flowhand_gather", 0), ("slope_twi", 20), ("flowhand_gfi_solve_finish", 28),
)


def rank_ops(tile, layout, exchange, d8=True):
    """The serial schedule of one rank's step as [(name, call)] in launch order -- what run_rank(overlap=False)
    executes and bench.py times stage by stage.  The two all-gathers are the only communication; the independent
    kernels (downslope; slope+TI+MTI) are queued between each gather's launch and the wait on it, so they overlap
    the transfer.  Nothing synchronises with the host.  d8=False: the tile already holds its D8 codes, core and halo
    (condition_rank / condition_ranks: the conditioned codes; or a D8 raster from a GIS tool, Example/example.py:36) --
    the "d8" stage is then a no-op instead of overwriting them with the plain steepest descent."""
    st = {}

    def fa_local():
        tile.fa_local(sync=False)
        tile.fill_ring_codes()

    def fa_gather():
        st["fa"] = exchange.gather(tile.fa_row, exchange.fa_all)

    def fa_finish_fh_local():
        exchange.wait()
        tile.fa_solve_finish_fh_local(st["fa"])

    def fh_gather():
        st["fh"] = exchange.gather(tile.fh_row, exchange.fh_all)

    def fh_finish():
        exchange.wait()
        tile.fh_solve_finish(st["fh"], fuse_gfi=True, want_a_river=False)

    calls = (tile.d8 if d8 else (lambda: None), fa_local, fa_gather, tile.downslope, fa_finish_fh_local, fh_gather,
             tile.slope_twi, fh_finish)
    return [(name, fn) for (name, _), fn in zip(RANK_OPS, calls)]


def run_rank(tile, layout, exchange, overlap=True, d8=True):
    """One step of one rank.  overlap=True (the default, as chain.Chain's): downslope runs as a second compute
    branch on its own stream from the D8 kernel to the end of the step, beside the flow kernels as well as the
    exchanges; overlap=False: the serial schedule of rank_ops().  Nothing in the step synchronises with the host.
    d8=False: start from the D8 codes the tile holds (see rank_ops)."""
    if not overlap:
        for _, fn in rank_ops(tile, layout, exchange, d8=d8):
            fn()
        return
    if d8:
        tile.d8()
    tile.downslope(side=True)
    tile.fa_local(sync=False)
    tile.fill_ring_codes()
    rows = exchange.gather(tile.fa_row, exchange.fa_all)
    exchange.wait()
    tile.fa_solve_finish_fh_local(rows)
    rows = exchange.gather(tile.fh_row, exchange.fh_all)
    tile.slope_twi()
    exchange.wait()
    tile.fh_solve_finish(rows, fuse_gfi=True, want_a_river=False)
    tile.join_side()


def simulate(tiles, layout):
    """N logical ranks on ONE device, lock-step, with the all-gathers replaced by list collection and the
    rank-level solves done by the numpy restatement (solve_flowacc / solve_flowhand): what proves
    tiled == untiled without a multi-GPU node."""
    for t in tiles:
        t.d8()
    fa = [tuple(a.cpu().numpy() for a in t.fa_local()) for t in tiles]
    ext = solve_flowacc(layout, fa)
    for t in tiles:
        t.fa_finish(ext[t.rank])
    codes = [t.ring_codes() for t in tiles]
    fh = [tuple(a.cpu().numpy() for a in t.fh_local()) for t in tiles]
    res = solve_flowhand(layout, fh, codes)
    for t in tiles:
        t.fh_finish(res[t.rank])
        t.pointwise()


def simulate_dev(tiles, layout, d8=True):
    """Same as simulate(), but with the product's rank-level solves on the GPU: the gathered buffer is
    the concatenation of the logical ranks' summary rows, exactly what the RCCL all-gather delivers.
    d8=False: the tiles already hold their D8 codes (condition_local / condition_ranks)."""
    import torch

    def gather(rows):  # the logical ranks own one stream each: order them around the copy by hand
        for t in tiles:
            t.ctx.sync()
        out = torch.cat(rows)
        torch.cuda.synchronize()
        return out
    for t in tiles:
        if d8:
            t.d8()
        t.fa_local(sync=False)
        t.fill_ring_codes()
    rows = gather([t.fa_row for t in tiles])
    for t in tiles:
        t.fa_solve_finish_fh_local(rows)
    rows = gather([t.fh_row for t in tiles])
    for t in tiles:
        t.fh_solve_finish(rows, fuse_gfi=True)
        t.slope_twi()
        t.downslope()


# ---------------------------------------------------------------------------------------------------
# evaluation of a tiled descriptor (BASELINE.json configs[4]: "full chain + evaluation.py flood-map classifier")
# ---------------------------------------------------------------------------------------------------
class DistComm:
    """what a rank exchanges with the others outside its step, over torch.distributed: small host arrays
    (all_gather: the classifier's extremes and counts, 3-96 numbers per calibration stage) and walker records
    (exchange_rows: an all-to-all of DEVICE buffers over RCCL; through the CPU on gloo, the rehearsal backend)"""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.cpu = dist.get_backend(group) == "gloo"

    def all_gather(self, value):
        out = [None] * self.size
        self.dist.all_gather_object(out, np.asarray(value), group=self.group)
        return out

    def all_gather_ints(self, vec):
        """vec: int64 device tensor [k] -> host numpy [size, k] (one small collective, one synchronisation)"""
        import torch
        v = vec.to(torch.int64)
        if self.cpu:
            v = v.cpu()
        out = torch.empty(self.size * v.numel(), dtype=torch.int64, device=v.device)
        self.dist.all_gather_into_tensor(out, v.contiguous(), group=self.group)
        return out.cpu().numpy().reshape(self.size, -1)

    def all_gather_host_ints(self, values, device=None):
        """values: a few host integers -> host numpy [size, k] (device: where the collective's buffer lives on RCCL --
        the caller's GPU; default: torch's current device)"""
        import torch
        if self.cpu:
            dev = "cpu"
        else:
            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        return self.all_gather_ints(torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev))

    def exchange_rows(self, rows, send_counts, recv_counts):
        """rows: [n, w] int32 device tensor sorted by destination rank; send_counts / recv_counts: rows per peer (host
        ints).  Returns the rows the peers sent here, a device tensor [sum(recv_counts), w]."""
        import torch
        w = rows.shape[1]
        src = rows.reshape(-1)
        if self.cpu:
            src = src.cpu()
        out = torch.empty(int(sum(recv_counts)) * w, dtype=rows.dtype, device=src.device)
        self.dist.all_to_all_single(out, src.contiguous(), output_split_sizes=[int(c) * w for c in recv_counts],
                                    input_split_sizes=[int(c) * w for c in send_counts], group=self.group)
        return out.to(rows.device).reshape(-1, w)


class LocalComm:
    """the same between N logical ranks of ONE process, each running in its own thread (one-GPU rehearsals and
    tests): barrier-synchronised exchanges through shared slots; device tensors are handed over as they are"""

    class _Shared:
        def __init__(self, n):
            import threading
            # two sets of slots, used in turn: every exchange has a barrier, so no thread is more than one exchange
            # ahead of another, and the slots of exchange k are not written again before exchange k + 2 -- after the
            # barrier of k + 1, which every reader of k has passed.  One barrier per exchange instead of two.
            self.n, self.slots, self.barrier = n, [[None] * n, [None] * n], threading.Barrier(n)

    def __init__(self, shared, rank):
        self.sh, self.rank, self.size, self._turn = shared, rank, shared.n, 0

    @staticmethod
    def create(n):
        sh = LocalComm._Shared(n)
        return [LocalComm(sh, r) for r in range(n)]

    def _swap(self, mine):
        slots = self.sh.slots[self._turn]
        self._turn ^= 1
        slots[self.rank] = mine
        self.sh.barrier.wait()
        return list(slots)

    def all_gather(self, value):
        return self._swap(np.asarray(value).copy())

    def all_gather_ints(self, vec):
        return np.stack(self._swap(vec.cpu().numpy().astype(np.int64)))

    def all_gather_host_ints(self, values, device=None):
        return np.stack(self._swap(np.asarray([int(v) for v in values], np.int64)))

    def exchange_rows(self, rows, send_counts, recv_counts):
        import torch
        offs = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
        st = torch.cuda.current_stream()
        st.synchronize()  # the other threads' streams read these rows
        pieces = [s[self.rank] for s in self._swap([rows[int(offs[d]):int(offs[d + 1])] for d in range(self.size)])]
        out = torch.cat(pieces)
        # copied before this thread reaches the next barrier: the senders' rows stay referenced by the slots until the
        # exchange after next overwrites them, and that one starts behind that barrier.  (Not record_stream: the tiles'
        # streams are destroyed with the tiles, and the allocator would still hold them.)
        st.synchronize()
        return out


def finish_downslope(tile, comm, max_iters=200, stats=None):
    """Downslope walks that left a rank's memory (none on the synthetic benchmark terrain, thousands along every border
    on real terrain, whose walks run for kilometres through flats and along valley floors; the reference's analogue is
    the CPU repair downslope.py:373-374).  Every rank calls this after its step.  Such a walk travels on as a WALKER
    record (include/descriptools_hip.h, dt_dev_downslope_emit_w): start cell, the cell it stands on, moves and diagonal
    moves made, the start height -- emitted by the downslope kernel where the walk left the rank (RankTile(emit_walkers),
    the default with long_walks), or seeded at the start cell from the -50 marks.  Each iteration a rank advances the
    walkers standing in its memory (dt_dev_downslope_walk_w: across the rank in skips of 64 moves when it has the
    long-walk tables), then the records go where they belong in ONE all-to-all of device buffers (comm.exchange_rows:
    RCCL; DistComm on gloo and LocalComm rehearse it) -- a finished walker to the owner of its start cell, which writes
    the value, the others to the owner of the cell they stand on -- preceded by one tiny all-gather of the counts, the
    iteration's only synchronisation with the host.  The result is the reference's float32 whatever the route: counts
    give it through the rounding-safety test of the count form, and the rare walk that fails the test starts again
    carrying the reference's own sequential float64 sum.  Returns the number of cells resolved (over all ranks); 0
    without any exchange of records when no rank had any.  stats (a dict, optional): filled with the number of
    iterations and this rank's wall-clock seconds per phase."""
    import time
    tc, L, layout = tile.torch, tile.L, tile.layout
    t_begin = time.perf_counter()
    phase = {"setup": 0.0, "kernels": 0.0, "counts": 0.0, "exchange": 0.0}
    n_local = tile.unresolved_downslope()
    i32 = tc.int32
    with tile.on_stream():
        total = int(np.sum(comm.all_gather_host_ints([n_local], device=tile.dev)))
    if total == 0:
        return 0
    with tile.on_stream():
        rec = None
        if tile._walkers is not None:
            emitted = int(tile._walkers[:4].view(i32).item())
            if emitted == n_local and n_local <= (tile._walkers.numel() - 256) // WALKER_BYTES:
                # (advanced in place: the buffer is the next step's to overwrite)
                rec = tile._walkers[256:256 + WALKER_BYTES * n_local].view(i32).reshape(n_local, WALKER_WORDS)
        if rec is None:  # no records (or more walks than the buffer holds): from the -50 marks, at the start cells
            ys, xs = (tile.core("down") == -50.0).nonzero(as_tuple=True)
            ys, xs = ys.to(i32).contiguous(), xs.to(i32).contiguous()
            rec = tc.empty((int(ys.numel()), WALKER_WORDS), dtype=i32, device=tile.dev)
            tile._chk(L.dt_dev_downslope_walk_seed_w(tile.ctx.h, C.byref(tile.win), tile.p("dem"), int(ys.numel()),
                                                     ys.data_ptr(), xs.data_ptr(), rec.data_ptr()))
        if getattr(tile, "_route", None) is None:  # the layout's bands on the device, once per tile
            tile._route = (tc.as_tensor(np.asarray(layout.ys, np.int32), device=tile.dev),
                           tc.as_tensor(np.asarray(layout.xs, np.int32), device=tile.dev),
                           tc.zeros(comm.size + 1, dtype=i32, device=tile.dev))
        ys_t, xs_t, counts = tile._route
    work = tile._lift_work
    phase["setup"] = time.perf_counter() - t_begin
    iters = 0
    for _ in range(max_iters):
        iters += 1
        t0 = time.perf_counter()
        with tile.on_stream():
            n = int(rec.shape[0])
            rec = rec.contiguous()
            send = tc.empty_like(rec)
            scratch = tc.empty(n + comm.size, dtype=i32, device=tile.dev)
            # arrivals that are finished are written home, the others advance; then the records are grouped by where
            # they go next and counted -- all on the device (dt_dev_downslope_walk_route_w)
            tile._chk(L.dt_dev_downslope_walk_route_w(
                tile.ctx.h, C.byref(tile.win), tile.p("dem"), tile.p("fdr"), tile.px, tile.dz, n, rec.data_ptr(),
                work.data_ptr() if work is not None else None, work.numel() if work is not None else 0,
                tile.p("down"), ys_t.data_ptr(), layout.ty, xs_t.data_ptr(), layout.tx, send.data_ptr(),
                counts.data_ptr(), scratch.data_ptr()))
            t1 = time.perf_counter()
            m = comm.all_gather_ints(counts)                    # [size, size + 1] on the host: the one synchronisation
            t2 = time.perf_counter()
            phase["kernels"] += t1 - t0
            phase["counts"] += t2 - t1
            if int(m[:, :comm.size].sum()) == 0:                # nobody sends anything: every walker is home
                break
            rec = comm.exchange_rows(send[:int(m[comm.rank, :comm.size].sum())], m[comm.rank, :comm.size],
                                     m[:, comm.rank])
            phase["exchange"] += time.perf_counter() - t2
    else:
        raise RuntimeError("finish_downslope: walkers still on their way after %d exchanges" % max_iters)
    with tile.on_stream():
        tile.n_unres.zero_()
    tile.ctx.sync()
    if stats is not None:
        stats.update(iterations=iters, seconds=time.perf_counter() - t_begin, **{"s_" + k: v for k, v in phase.items()})
    return total


def evaluate_rank(tile, flood_core, comm, name="hand", under="under", class_map=False):
    """Example/example.py:113-147 on ONE rank's core window of a tiled descriptor raster (default: the HAND this
    tile's step left in tile.t["hand"]), every rank calling it at the same time: the np.unique extremes are
    all-gathered and combined (evaluation.combine_extremes), the confusion counts of every calibration stage summed
    over ranks -- exact integers, so every rank finds the threshold evaluation.calibration finds on the whole raster.
    flood_core: this rank's window of the benchmark flood map (int8 H x W device tensor, contiguous).  Returns
    evaluation.evaluate_resident's dict (+ "class_map": int32 H x W device tensor when asked for)."""
    from . import evaluation
    tc = tile.torch
    with tile.on_stream():
        x = tile.core(name).contiguous()
        klass = tc.empty((tile.H, tile.W), dtype=tc.int32, device=tile.dev) if class_map else None
    tile.ctx.sync()
    assert flood_core.is_contiguous() and flood_core.dtype == tc.int8 and tuple(flood_core.shape) == (tile.H, tile.W)
    # binary_map's nodata is the scaled value at GLOBAL cell [0, 0] (evaluation.py:111): its owner passes it on
    own00 = tile.gy0 == 0 and tile.gx0 == 0
    h00 = float(x[0, 0].item()) if own00 else float("nan")
    state = {}

    def reduce_extremes(e):
        rows = comm.all_gather(np.concatenate([np.asarray(e, np.float64), [h00, 1.0 if own00 else 0.0]]))
        g = evaluation.combine_extremes([r[:3] for r in rows])
        v00 = [r[3] for r in rows if r[4] == 1.0][0]
        mn, mx = np.float32(g[1]), np.float32(g[2])
        state["first"] = float("nan") if v00 == -100.0 or v00 != v00 else float((np.float32(v00) - mn) / (mx - mn))
        return g

    def reduce_counts(c):
        return np.sum(comm.all_gather(np.asarray(c, np.int64)), axis=0)

    res = evaluation.evaluate_resident(tile.ctx, x.data_ptr(), flood_core.data_ptr(), tile.H * tile.W, under,
                                       reduce_extremes=reduce_extremes, reduce_counts=reduce_counts,
                                       nodata_first=lambda: state["first"],
                                       class_ptr=klass.data_ptr() if class_map else None)
    if class_map:
        res["class_map"] = klass
    return res


# ---------------------------------------------------------------------------------------------------
# hydrological conditioning over ranks (single raster: dt_dev_condition_d8)
# ---------------------------------------------------------------------------------------------------
def condition_ranks(tiles, exchange, any_flag, rounds=4, max_iter=1 << 30):
    """Depression filling + flat routing of a tiled DEM: the D8 codes every rank's step then starts from (instead of
    RankTile.d8()).  Both fixed points are monotone relaxations, so each rank relaxes its own core against the halo it
    has, the halos are exchanged, and the loop ends with the first iteration in which no rank changed anything -- the
    result is the single raster's (tests/test_gpu_hydro.py: bit-identical).  One host synchronisation per iteration
    (the flag), as in the single raster's synchronous form.

    tiles: the RankTiles of THIS process with their DEM halos in place (one for a real rank, all of them for logical
    ranks on one device); exchange(name): brings the halo of raster `name` up to date on every tile (exchange_halo
    over torch.distributed, or exchange_halo_local); any_flag(values) -> bool: OR over ALL ranks of the per-tile
    flags (an all-reduce for real ranks).  Returns (cells left without a code -- 0 --, fill iterations, flat
    iterations)."""
    for t in tiles:
        t.cond_alloc()
        t.cond_stage(0)
    exchange("filled")
    it_fill = it_flat = 0
    for it_fill in range(1, max_iter + 1):
        for t in tiles:
            t.cond_stage(1, rounds)
        changed = [t.cond_flag() for t in tiles]
        exchange("filled")
        if not any_flag(changed):
            break
    for t in tiles:
        t.cond_d8()
        t.cond_stage(2)
    exchange("dist")
    for it_flat in range(1, max_iter + 1):
        for t in tiles:
            t.cond_stage(3, rounds)
        changed = [t.cond_flag() for t in tiles]
        exchange("dist")
        if not any_flag(changed):
            break
    for t in tiles:
        t.cond_stage(4)
    left = [t.cond_flag() for t in tiles]
    exchange("fdr")
    return sum(left), it_fill, it_flat


def condition_local(tiles, layout, rounds=4):
    """condition_ranks for N logical ranks living in one process (one device)"""
    def exchange(name):
        for t in tiles:
            t.ctx.sync()
        exchange_halo_local([t.t[name] for t in tiles], layout, tiles[0].halo)
        tiles[0].torch.cuda.synchronize()
    return condition_ranks(tiles, exchange, any, rounds)


def condition_rank(tile, layout, group=None, rounds=4):
    """condition_ranks for one real rank: point-to-point halo exchanges with the <= 8 neighbours, an all-reduce (MAX) of
    the flag per iteration"""
    import torch
    import torch.distributed as dist
    cpu = dist.get_backend(group) == "gloo"

    def exchange(name):
        tile.ctx.sync()
        x = tile.t[name]
        if cpu:
            h = exchange_halo(x.cpu(), layout, tile.rank, tile.halo, group)
            with tile.on_stream():
                x.copy_(h)
            tile.ctx.sync()
        else:
            with tile.on_stream():
                exchange_halo(x, layout, tile.rank, tile.halo, group)
            tile.ctx.sync()

    def any_flag(vals):
        f = torch.tensor([1 if any(vals) else 0], dtype=torch.int32, device="cpu" if cpu else tile.dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX, group=group)
        return bool(f.item())
    return condition_ranks([tile], exchange, any_flag, rounds)
