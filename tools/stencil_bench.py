"""Ablation timings of the stencil kernel variants at 16384^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from descriptools_amd import _lib
from descriptools_amd.device import Context
L = _lib.lib()
S = 16384
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = Context(0, st.cuda_stream)
f32 = lambda: torch.empty((S, S), dtype=torch.float32, device="cuda")
dem, slope, rad, ti, mti = f32(), f32(), f32(), f32(), f32()
fdr = torch.empty((S, S), dtype=torch.uint8, device="cuda")
fac = torch.randint(0, 5000, (S, S), dtype=torch.int32, device="cuda")
_lib.check(L.dt_dev_synth_dem(ctx.h, 1, S, S, 0, 0, S, S, 0, dem.data_ptr()))
def t(name, fn, bytes_per_cell):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5): fn()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-28s %.3f ms  %.0f GB/s" % (name, ms, S * S * bytes_per_cell / ms / 1e6))
t("copy (torch, 8 B/cell)", lambda: slope.copy_(dem), 8)
for blocks in (1024, 2048, 4096, 8192, 65536):
    t("hip float4 copy, %d wg" % blocks, lambda: L.dt_dev_membench_copy(ctx.h, dem.data_ptr(), slope.data_ptr(), S * S, blocks), 8)
t("slope only (8 B)", lambda: L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, slope.data_ptr(), None, None), 8)
t("d8 only (5 B)", lambda: L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, None, fdr.data_ptr(), None), 5)
t("slope + rad (12 B)", lambda: L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, slope.data_ptr(), None, rad.data_ptr()), 12)
t("slope+rad+ti+mti (24 B)", lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), rad.data_ptr(), ti.data_ptr(), mti.data_ptr()), 24)
t("slope+ti+mti (20 B)", lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr()), 20)
t("ti+mti only (16 B)", lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, None, None, ti.data_ptr(), mti.data_ptr()), 16)
t("twi pointwise (16 B)", lambda: L.dt_dev_twi(ctx.h, fac.data_ptr(), rad.data_ptr(), S * S, 10.0, 0.1, ti.data_ptr(), mti.data_ptr()), 16)

# ---- the fused kernel on the chain's own accumulation raster, store-policy A/B (interleaved) ----
river = torch.empty((S, S), dtype=torch.int8, device="cuda")
_lib.check(L.dt_dev_slope_d8(ctx.h, dem.data_ptr(), S, S, 10.0, None, fdr.data_ptr(), None))
_lib.check(L.dt_dev_flowacc_river(ctx.h, fdr.data_ptr(), dem.data_ptr(), S, S, S * S // 512, fac.data_ptr(), river.data_ptr()))
torch.cuda.synchronize()
for rep in range(3):
    for nt in (0, 1):
        _lib.check(L.dt_debug_set(1, nt))
        t("real fac, plain=%d slope+ti+mti" % nt, lambda: L.dt_dev_slope_twi(ctx.h, dem.data_ptr(), fac.data_ptr(), S, S, 10.0, 0.1, slope.data_ptr(), None, ti.data_ptr(), mti.data_ptr()), 20)
_lib.check(L.dt_debug_set(1, 0))
