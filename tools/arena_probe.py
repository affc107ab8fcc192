"""Placement BY CONSTRUCTION (round 4): the chain's rasters carved from ONE arena at chosen byte offsets.

profiles/r4/placement_map_delta.txt: inside one large allocation the slowdown of two lock-step write streams is a
function of their address DIFFERENCE with a period of 4 GiB (worst at 0 .. +1.25 GiB, mild around +2 .. +3 GiB), and
vanishes between regions >= ~30 GiB apart.  This tool measures what that means for the chain's two multi-output kernels
at 16384^2: the fused slope + TI + MTI stencil with its three outputs at (0, a, b) GiB, and HAND's last pass with its
five outputs on a few spacings, everything else unchanged (tune_placement off).

    python tools/arena_probe.py [stencil|hand|all] [size]
"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from descriptools_amd import _lib, chain  # noqa: E402
from descriptools_amd.device import Context  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
GIB = 1 << 30
L = _lib.lib()
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
N = S * S
RB = N * 4                      # bytes of a 4-byte raster
UNIT = RB                       # offsets below are in units of one raster (1 GiB at 16384^2)
ARENA = int(26 * UNIT)
arena = torch.empty(ARENA, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
base += (-base) % (2 << 20)     # 2 MiB aligned
dem = torch.empty((S, S), dtype=torch.float32, device="cuda")
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
small = {}                      # the byte rasters and the ones not under test live outside the arena


def run_layout(offsets):
    """offsets: {name: offset in raster units} for 4-byte rasters placed in the arena; others allocated normally.
    Returns per-op ms of the serial schedule."""
    names = [n for n, _ in chain.OUTPUTS]
    calls = iter(names)

    def alloc(shape, dt):
        n = next(calls)
        if n in offsets:
            return base + int(offsets[n] * UNIT) // 256 * 256
        key = (n, np.dtype(dt).itemsize)
        if key not in small:
            small[key] = torch.empty(shape, dtype={4: torch.float32, 1: torch.uint8}[np.dtype(dt).itemsize], device="cuda")
        return small[key].data_ptr()
    ch = chain.Chain(S, S, ctx=ctx, px=10.0, river_threshold=N // 512, alloc=alloc, overlap=False,
                     want_slope_rad=False, tune_placement=False)
    ops = ch.ops(dem.data_ptr(), want_a_river=False, serial=True)
    for _, _, fn in ops:
        _lib.check(fn())
    reps = 5
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ops] for _ in range(reps)]
    for k in range(reps):
        for i, (_, _, fn) in enumerate(ops):
            ev[k][i][0].record(st)
            _lib.check(fn())
            ev[k][i][1].record(st)
    torch.cuda.synchronize()
    return {name: float(np.median([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(reps)]))
            for i, (name, _, _) in enumerate(ops)}


if what in ("stencil", "all"):
    print("fused slope + TI + MTI stencil (ms) with slope / TI / MTI at (0, a, b) raster units of %d MiB; rows a, columns b"
          % (UNIT >> 20), flush=True)
    grid = [1, 1.25, 1.5, 2, 2.5, 3, 3.5, 4, 5, 6, 7]
    print("   a\\b " + " ".join("%6.2f" % b for b in grid))
    for a in grid:
        row = []
        for b in grid:
            if abs(a - b) < 1:
                row.append("   -  ")
                continue
            t = run_layout({"slope": 0, "ti": a, "mti": b})
            row.append("%6.3f" % t["slope_twi"])
        print("%6.2f " % a + " ".join(row), flush=True)
    # order matters? (which stream leads)
    for perm in itertools.permutations(("slope", "ti", "mti")):
        t = run_layout(dict(zip(perm, (0, 3, 6))))
        print("order %s at (0, 3, 6): %.3f ms" % (perm, t["slope_twi"]), flush=True)
if what in ("hand", "all"):
    five = ("fdist", "idx", "hand", "gfi", "lnhlh")
    print("HAND's last pass (fdist, idx, hand, gfi, lnhlh written together), ms, by spacing of the five rasters:", flush=True)
    for step in (1, 1.25, 1.5, 2, 2.5, 3, 3.5, 4.5, 5):
        t = run_layout({n: 8 * 0 + k * step for k, n in enumerate(five)})
        print("  step %.2f: flowhand_gfi_finish %.3f ms   (whole serial step %.3f)" % (step, t["flowhand_gfi_finish"], sum(t.values())), flush=True)
    for offs in ((0, 3, 6, 1.5, 4.5), (0, 2.5, 5, 7.5, 10), (0, 3, 6, 9.5, 12.5), (0, 2, 3, 5, 6), (0, 3, 5, 8, 10)):
        t = run_layout(dict(zip(five, offs)))
        print("  offsets %s: %.3f ms" % (offs, t["flowhand_gfi_finish"]), flush=True)
