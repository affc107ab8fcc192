#!/usr/bin/env python3
"""bench.py -- full descriptor chain on a device-resident synthetic DEM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S]

A step = one pass of the whole chain (slope -> D8 -> flow accumulation -> river mask -> flow
distance / river index / HAND -> fused slope+TI+MTI -> GFI -> ln(hl/H) -> downslope) over one
S x S "tilted integer fBm" DEM that is already in HBM when the timed region starts.
N = 1 runs BASELINE.json configs[2], the 16384^2 DEM its metric is quoted on.  For N > 1 (launched
by torch.distributed.run, one rank per GPU) every rank owns one S x S tile of a larger DEM (weak
scaling).  Rank 0 prints ONE JSON line.

Timing: K steps bracketed by barrier + synchronize on both sides, max over ranks.  Per-op times
come from HIP events recorded on the stream the kernels run on, inside the same timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

# op -> (algorithmic bytes per cell).  "d8" writes fdr only (slope comes out of the
# fused slope+TWI stencil: dem 4 + fac 4 read, slope 4 + TI 4 + MTI 4 written = the north_star's
# 20 B/cell; the slope-in-radians raster is an optional extra output that the chain does not need).
# Algorithmic (compulsory) bytes per cell of each op as it is fused here (SURVEY.md 8d lists the unfused
# definitions, which add up to the chain's 90 B): HAND + GFI + ln(hl/H) in one go reads fdr 1 + river 1 (pass 1)
# and dem 4 + fac 4 (last pass) and writes fdist, idx, hand, gfi, lnhlh (20).  The op is timed as its two
# phases -- the windowed entry points with the whole raster as the window, the same kernels as
# dt_dev_flowhand_gfi -- so that its last pass, one kernel, has a duration of its own.
OPS = [("d8", 5), ("downslope", 9), ("flowacc_river", 5 + 1), ("flowhand_local", 2), ("flowhand_gfi_finish", 28),
       ("slope_twi", 20)]


# kernels behind each op (names as rocprofv3 prints them) -- used to attach the PMC-measured HBM
# traffic (profiles/r1/v2_pmc_traffic.json, collected in separate --pmc runs of this script) to an op
OP_KERNELS = {
    "d8": ["k_stencil<false, true, false, false>"],
    "flowacc_river": ["k_fa_tile1", "k_fa_reduce", "k_fa_poison", "k_fa_tile3<true, true>"],
    "flowhand_local": ["k_fh_tile1n", "k_fh_tile1", "k_fh_ghost_init", "k_fh_node_jump", "k_fh_rank_summary"],
    "flowhand_gfi_finish": ["k_fh_tile3"],
    "slope_twi": ["k_stencil<true, false, false, true>"],
    "downslope": ["k_downslope_win"],
}
PMC_FILE = os.path.join(ROOT, "profiles", "r1", "v10_pmc_traffic.json")


def pmc_traffic(op, size):
    """HBM bytes per op from the committed PMC run (None when not measured for this size)."""
    try:
        d = json.load(open(PMC_FILE))
        if d["size"] != size:
            return None
        tot = 0.0
        for k in OP_KERNELS[op]:
            e = d["kernels"][k]
            tot += (2.0 * e["fetch_kb_step"] + e["write_kb_step"]) * 1024.0
        return tot
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=16384, help="tile edge per GPU")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="run downslope as a second branch on its own stream "
                    "beside flow accumulation / HAND (Chain(overlap=True)): faster end to end, but the per-kernel "
                    "timings stop being attributable; off by default")
    ap.add_argument("--tiled", action="store_true", help="N = 1 through the multi-rank path (1 x 1 layout): "
                                                         "measures what the tiling machinery costs")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and use the "
                    "collective path even with one rank (rehearsal of the RCCL calls on a 1-GPU box)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; default) or gloo (rehearsal of N > 1 "
                                                        "with several ranks sharing one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "gloo":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from descriptools_amd import _lib, chain
    from descriptools_amd.device import Context
    L = _lib.lib()

    if world > 1 or args.tiled or args.force_dist:
        return main_tiled(args, torch, dist, world, rank, local_rank, dev)

    S = args.size
    H = W = S
    # a torch-owned NON-default stream: the library launches on it and torch events see it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = Context(device=local_rank, stream=stream.cuda_stream)
    # second branch of the chain (downslope beside flow accumulation / HAND, chain.Chain.run): its own stream
    stream2 = torch.cuda.Stream(device=dev) if args.overlap else stream
    ctx2 = Context(device=local_rank, stream=stream2.cuda_stream) if args.overlap else ctx

    # ---- tile layout: ranks tile a (ty*S) x (tx*S) DEM; rank r owns tile (r // tx, r % tx) ----
    tx = 1
    while tx * tx < world:
        tx *= 2
    ty = (world + tx - 1) // tx
    Hg, Wg = ty * S, tx * S
    y0, x0 = (rank // tx) * S, (rank % tx) * S

    def alloc(shape, dt):
        tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8,
               np.int32: torch.int32}[dt]
        return torch.empty(shape, dtype=tdt, device=dev)

    bufs = {}

    def alloc_ptr(shape, dt):
        t = alloc(shape, dt)
        bufs[len(bufs)] = t
        return t.data_ptr()

    dem = alloc((H, W), np.float32)
    _lib.check(L.dt_dev_synth_dem(ctx.h, args.seed, Hg, Wg, y0, x0, H, W, 0, dem.data_ptr()))
    ch = chain.Chain(H, W, ctx=ctx, px=10.0, river_threshold=(H * W) // 512, alloc=alloc_ptr, side_ctx=ctx2,
                     overlap=args.overlap)
    p = ch.p
    c, c2 = ctx.h, ctx2.h
    N = H * W

    # HAND as its two phases: the window is the whole raster; the ring summary of phase 1 (for other ranks) is
    # written and ignored
    import ctypes as C
    full = _lib.Window(H, W, W, 0, 0, H, W, 0)
    P = int(L.dt_perim_cells(H, W))
    ring = [torch.empty(max(P, 1), dtype=dt_, device=dev) for dt_ in (torch.uint8, torch.int32, torch.int32,
                                                                        torch.int32, torch.float32, torch.int32)]
    # the ops of chain.Chain.run, in its order and on its stream(s): (name, stream, call); without --overlap
    # ctx2 is ctx and downslope simply runs after D8
    def op_calls():
        return [
            ("d8", stream, lambda: L.dt_dev_slope_d8(c, dem.data_ptr(), H, W, ch.px, None, p("fdr"), None)),
            ("downslope", stream2, lambda: L.dt_dev_downslope(c2, dem.data_ptr(), p("fdr"), H, W, ch.px, ch.dz, 0,
                                                              p("down"))),
            ("flowacc_river", stream, lambda: L.dt_dev_flowacc_river(c, p("fdr"), dem.data_ptr(), H, W,
                                                                     ch.river_threshold, p("fac"), p("river"))),
            ("flowhand_local", stream, lambda: L.dt_dev_flowhand_local_w(
                c, C.byref(full), dem.data_ptr(), p("fdr"), p("river"), p("fac"), *[t.data_ptr() for t in ring])),
            ("flowhand_gfi_finish", stream, lambda: L.dt_dev_flowhand_gfi_finish_w(
                c, C.byref(full), dem.data_ptr(), p("fdr"), p("river"), p("fac"), ch.px, ch.n_gfi, ch.b, None, None,
                None, None, None, None, p("fdist"), p("idx"), None, p("hand"), None, p("gfi"), p("lnhlh"))),
            ("slope_twi", stream, lambda: L.dt_dev_slope_twi(c, dem.data_ptr(), p("fac"), H, W, ch.px, ch.n_top,
                                                             p("slope"), None, p("ti"), p("mti"))),
        ]

    calls = op_calls()
    assert [n for n, _, _ in calls] == [n for n, _ in OPS]

    def step(events=None):
        for i, (name, st, fn) in enumerate(calls):
            if name == "downslope" and args.overlap:
                ctx.fork(ctx2)  # after D8
            if events is not None:
                events[i][0].record(st)
            _lib.check(fn())
            if events is not None:
                events[i][1].record(st)
        if args.overlap:
            ctx.join(ctx2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in calls]
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-op device times (HIP events on the launch stream) ----
    per_op = {}
    for i, (name, bpc) in enumerate(OPS):
        ms = float(np.mean([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(args.steps)]))
        gbs = N * bpc / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        per_op[name] = {"ms": round(ms, 4), "algo_bytes_per_cell": bpc, "achieved_GBs": round(gbs, 1),
                        "frac": round(gbs / HBM_PEAK_GBS, 4)}
    for name in per_op:
        tr = pmc_traffic(name, S)
        per_op[name]["traffic_bytes"] = None if tr is None else int(tr)
        per_op[name]["kernels"] = [k for k in OP_KERNELS[name] if not k.startswith("__amd")]
    # the dominant KERNEL: ops that are one kernel are timed exactly by their events; the kernels of the
    # multi-kernel ops (flow accumulation, HAND's first phase) are <= 1.4 ms each (profiles/)
    single = [k for k in per_op if len(per_op[k]["kernels"]) == 1]
    dom = max(single, key=lambda k: per_op[k]["ms"])
    roof = {"kernel": per_op[dom]["kernels"][0], "op": dom, "bound": "hbm",
            "achieved": per_op[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": per_op[dom]["frac"], "traffic": per_op[dom]["traffic_bytes"],
            "note": "dominant single kernel; achieved = algorithmic bytes/cell x cells / mean kernel time (HIP "
                    "events on the launch stream, timed region); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                    "from separate rocprofv3 --pmc runs committed under profiles/; per_op lists every op "
                    "(flowacc_river and flowhand_local are multi-kernel ops: their frac is of the op as a whole)"
                    + (".  --overlap: downslope runs on a second stream beside flow accumulation / HAND, so the "
                       "per-op times overlap and add up to more than ms_per_step" if args.overlap else "")}

    # practical HBM ceiling of this device: the better of two copies through the same library (float4
    # grid-stride; 1024 x 4 patches, whose pieces spread a workgroup over more HBM channels)
    src, dst = alloc((H, W), np.float32), alloc((H, W), np.float32)
    copy_gbs = 0.0
    for blocks in (65536, -1):
        if blocks < 0 and N % 65536 != 0:
            continue
        for _ in range(2):
            _lib.check(L.dt_dev_membench_copy(c, src.data_ptr(), dst.data_ptr(), N, blocks))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            _lib.check(L.dt_dev_membench_copy(c, src.data_ptr(), dst.data_ptr(), N, blocks))
        e1.record(stream)
        torch.cuda.synchronize()
        copy_gbs = max(copy_gbs, N * 8 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst

    cells = N * world
    value = cells * args.steps / dt / 1e6
    out = {
        "metric": "Mcells/s full descriptor chain", "value": round(value, 1), "unit": "Mcells/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d synthetic tilted-integer-fBm DEM per GPU, full chain "
                               "(d8, flowacc, river mask, flowhand/HAND with fused GFI + ln(hl/H), fused slope+TI+MTI, "
                               "downslope), device-resident" % (S, S),
                   "global_dem": "%dx%d" % (Hg, Wg), "px": 10.0, "river_threshold_cells": ch.river_threshold,
                   "parallelism": ("1 tile per GPU" if world > 1 else "single GPU")
                   + (", downslope on a second stream (--overlap)" if args.overlap else "")},
        "roofline": roof,
        "per_op": per_op,
        "hbm_copy_ceiling_GBs": round(copy_gbs, 1),
        "chain_algo_bytes_per_cell": chain.ALGO_BYTES_PER_CELL,
        "chain_frac_of_hbm_peak": round(cells * args.steps * chain.ALGO_BYTES_PER_CELL / dt / 1e9 / world
                                        / HBM_PEAK_GBS, 4),
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.seed)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main_tiled(args, torch, dist, world, rank, local_rank, dev):
    """N > 1: one S x S core tile per rank of a (ty*S) x (tx*S) DEM; halo generated locally from the
    global generator; two RCCL all-gathers of ring summaries per step (flow accumulation, HAND)."""
    from descriptools_amd import chain, tiling
    S = args.size
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    layout = tiling.Layout.uniform(world, S, S)
    tile = tiling.RankTile(layout, rank, device=local_rank, stream=stream.cuda_stream, px=10.0,
                           river_threshold=(layout.Hg * layout.Wg) // 512)
    tile.synth_dem(args.seed)

    use_dist = world > 1 or args.force_dist
    exchange = tiling.Exchange(tile, layout, world if not args.force_dist else max(world, 2))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    cpu_red = use_dist and dist.get_backend() == "gloo"

    for _ in range(args.warmup):
        tiling.run_rank(tile, layout, exchange, overlap=args.overlap)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tiling.run_rank(tile, layout, exchange, overlap=args.overlap)
    barrier()
    dt = time.perf_counter() - t0
    unres = torch.tensor([tile.unresolved_downslope()], dtype=torch.int64, device="cpu" if cpu_red else dev)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if cpu_red else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.all_reduce(unres)
    cells = S * S * world
    out = {
        "metric": "Mcells/s full descriptor chain", "value": round(cells * args.steps / dt / 1e6, 1),
        "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d synthetic tilted-integer-fBm DEM per GPU, full chain, device-resident"
                               % (S, S),
                   "global_dem": "%dx%d" % (layout.Hg, layout.Wg), "px": 10.0,
                   "river_threshold_cells": tile.river_threshold,
                   "parallelism": "%dx%d rank tiles, 64-cell halo, 2 RCCL all-gathers of ring summaries per "
                                  "step (flow accumulation inflow, HAND rank exits)" % (layout.ty, layout.tx)},
        "roofline": {"bound": "hbm", "achieved": round(cells * args.steps * chain.ALGO_BYTES_PER_CELL / dt / 1e9
                                                        / world, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(cells * args.steps * chain.ALGO_BYTES_PER_CELL / dt / 1e9 / world
                                   / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": "whole chain, per GPU (per-kernel figures: N = 1 run)"},
        "downslope_walks_beyond_halo": int(unres.item()),
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(seed, n=3072):
    """The oracle (the reference's per-cell algorithms restated in C, single thread) timed on a
    bounded sample: the same chain on an n x n DEM from the same generator.  Baseline only."""
    import oracle
    px = 10.0
    dem = oracle.synth_dem(seed, n, n)
    t0 = time.perf_counter()
    sl, fdr = oracle.slope_d8(dem, px)
    acc = oracle.flowacc(fdr, dem)
    river = (acc > (n * n) // 512).astype(np.int8)
    fd, idx, hand = oracle.flowhand(dem, fdr, river, px)
    slr = np.where(dem == -100, -100, np.arctan(sl / 100)).astype(np.float32)
    oracle.twi(acc, slr, px, 0.1)
    oracle.gfi(hand, acc, idx, 0.4, 0.1, px)
    oracle.lnhlh(hand, acc, 0.4, 0.1, px)
    oracle.downslope(dem, fdr, px, 5.0)
    dt = time.perf_counter() - t0
    return {"value": round(n * n / dt / 1e6, 3), "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": "full chain on a %dx%d DEM of the same generator (%.1f s, oracle/dt_oracle.c, "
                      "gcc -O2, 1 thread; reference-algorithm restatement, not Numba)" % (n, n, dt)}


if __name__ == "__main__":
    main()
