"""Device-resident descriptor chain (net-new, additive): slope -> D8 -> flow accumulation -> river
mask -> flow distance / river index / HAND -> TI / MTI -> GFI -> ln(hl/H) -> downslope, every raster
staying in HBM between steps (the reference round-trips each descriptor through the host,
Example/example.py:59-91).  Parameters default to the example's (n_top 0.1, n_gfi 0.4, b 0.1, dz 5)."""
import numpy as np

from . import _lib
from ._lib import check
from .device import Context

F32, U8, I8, I32 = np.float32, np.uint8, np.int8, np.int32

OUTPUTS = (("slope", F32), ("fdr", U8), ("fac", I32), ("river", I8), ("fdist", F32), ("idx", I32),
           ("hand", F32), ("a_river", I32), ("slope_rad", F32), ("ti", F32), ("mti", F32), ("gfi", F32),
           ("lnhlh", F32), ("down", F32))

# Unfused algorithmic bytes per cell of the chain: slope 8, D8 5, flow-acc 5, river
# mask 5, HAND 18, TI+MTI 16, GFI 12, ln(hl/H) 12, downslope 9.
ALGO_BYTES_PER_CELL = 90


class Chain:
    """Owns the output rasters of one H x W tile on one device."""

    def __init__(self, H, W, ctx=None, px=10.0, n_top=0.1, n_gfi=0.4, b=0.1, dz=5.0,
                 river_threshold=None, alloc=None, want_slope_rad=True, side_ctx=None, overlap=False):
        """overlap: run downslope as a second branch on its own stream (side_ctx, created on demand) beside
        the flow-accumulation / HAND kernels: ~5 % faster end to end at 16384^2, at the price of per-kernel
        timings that are no longer attributable (the branch is stretched over the whole step).  Off by
        default: one stream, kernels back to back."""
        self.want_slope_rad = want_slope_rad
        self.H, self.W, self.N = int(H), int(W), int(H) * int(W)
        self.ctx = ctx or Context()
        self.side = None
        if overlap:
            self.side = side_ctx if side_ctx is not None else Context(device=self.ctx.device)
        self._own_side = overlap and side_ctx is None
        self.px, self.n_top, self.n_gfi, self.b, self.dz = px, n_top, n_gfi, b, dz
        self.river_threshold = self.N // 512 if river_threshold is None else int(river_threshold)
        self.buf = {}
        for name, dt in OUTPUTS:
            self.buf[name] = alloc((H, W), dt) if alloc else self.ctx.empty((H, W), dt)

    def p(self, name):
        b = self.buf[name]
        return b.ptr if hasattr(b, "ptr") else b

    def run(self, dem_ptr):
        """Enqueue the whole chain (asynchronous).  Downslope needs only the DEM and the D8 codes, and the
        flow-accumulation / HAND kernels are latency chains that leave most of the GPU idle, so it runs as a
        second branch on the side context's stream between the D8 kernel and the end of the chain."""
        L, c, H, W, N = _lib.lib(), self.ctx.h, self.H, self.W, self.N
        p = self.p
        check(L.dt_dev_slope_d8(c, dem_ptr, H, W, self.px, None, p("fdr"), None))
        if self.side is not None:
            self.ctx.fork(self.side)
            check(L.dt_dev_downslope(self.side.h, dem_ptr, p("fdr"), H, W, self.px, self.dz, 0, p("down")))
        check(L.dt_dev_flowacc_river(c, p("fdr"), dem_ptr, H, W, self.river_threshold, p("fac"), p("river")))
        # HAND with GFI and ln(hl/H) evaluated in its last tile pass (one pass over the rasters less)
        check(L.dt_dev_flowhand_gfi(c, dem_ptr, p("fdr"), p("river"), p("fac"), H, W, self.px, self.n_gfi, self.b,
                                    p("fdist"), p("idx"), p("hand"), p("a_river"), p("gfi"), p("lnhlh")))
        check(L.dt_dev_slope_twi(c, dem_ptr, p("fac"), H, W, self.px, self.n_top, p("slope"),
                                 p("slope_rad") if self.want_slope_rad else None, p("ti"), p("mti")))
        if self.side is not None:
            self.ctx.join(self.side)
        else:
            check(L.dt_dev_downslope(c, dem_ptr, p("fdr"), H, W, self.px, self.dz, 0, p("down")))

    def free(self):
        if self._own_side:
            self.side.close()
            self.side, self._own_side = None, False
        for b in self.buf.values():
            if hasattr(b, "free"):
                b.free()
        self.buf = {}


def run_host(dem, px, **kw):
    """Convenience: host DEM in, dict of host rasters out (fac / idx widened to int64)."""
    dem32 = _lib.dem_f32(dem)
    H, W = dem32.shape
    ctx = Context()
    ch = Chain(H, W, ctx=ctx, px=px, **kw)
    d_dem = ctx.to_device(dem32)
    try:
        ch.run(d_dem.ptr)
        ctx.sync()
        out = {k: ch.buf[k].to_host() for k, _ in OUTPUTS}
    finally:
        d_dem.free()
        ch.free()
        ctx.close()
    out["fac"] = out["fac"].astype(np.int64)
    out["idx"] = out["idx"].astype(np.int64)
    return out
