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
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

PMC_FILE = os.path.join(ROOT, "profiles", "r4", "pmc_traffic.json")


def pmc_traffic(kernels, size):
    """HBM bytes of an op (the kernels behind it) from the committed PMC run (None when not measured for this
    size): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md's gfx950 correction."""
    try:
        d = json.load(open(PMC_FILE))
        if d["size"] != size:
            return None
        tot = 0.0
        for k in kernels:
            e = d["kernels"][k]
            tot += (2.0 * e["fetch_kb_step"] + e["write_kb_step"]) * 1024.0
        return tot
    except Exception:
        return None


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (a child process, before
    anything here has touched the GPU) and relay its output."""
    import subprocess
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def per_op_table(events, ops, cells, size, steps):
    """events[k][i] = (start, end) HIP events of op i in step k, recorded on the stream the op was launched on;
    ops = [(name, algorithmic bytes per cell, kernels)] -> per-op dict + the roofline block of the dominant
    single-kernel op"""
    per_op = {}
    for i, (name, bpc, kernels) in enumerate(ops):
        ms = float(np.mean([events[k][i][0].elapsed_time(events[k][i][1]) for k in range(steps)]))
        gbs = cells * bpc / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = pmc_traffic(kernels, size) if kernels else None
        per_op[name] = {"ms": round(ms, 4), "algo_bytes_per_cell": bpc, "achieved_GBs": round(gbs, 1),
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic_bytes": None if tr is None else int(tr),
                        "kernels": kernels}
    # the dominant KERNEL: ops that are one (hot) kernel are timed by their events; the kernels of the multi-kernel
    # ops (flow accumulation, HAND's first phase) are <= 1.4 ms each (profiles/)
    single = [k for k in per_op if per_op[k]["algo_bytes_per_cell"] > 0 and
              len([x for x in per_op[k]["kernels"] if "_fix" not in x]) == 1]
    dom = max(single, key=lambda k: per_op[k]["ms"])
    roof = {"kernel": per_op[dom]["kernels"][0], "op": dom, "bound": "hbm",
            "achieved": per_op[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": per_op[dom]["frac"], "traffic": per_op[dom]["traffic_bytes"],
            "note": "dominant single kernel; achieved = algorithmic bytes/cell x cells / mean kernel time (HIP "
                    "events on the launch stream, in the SERIAL timed loop of this process: ms_per_step_serial; "
                    "the headline ms_per_step is the overlapped schedule, where per-kernel durations are not "
                    "attributable); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc runs "
                    "committed under profiles/; per_op lists every op (multi-kernel ops: frac of the op as a whole)"}
    return per_op, roof


def default_workload(gpus, size, global_dem):
    """(size, global_dem) after the defaults: --gpus 8 with neither given is BASELINE.json configs[4] -- the
    north_star's "65536^2 DEM tiled over 8 MI355X" (2 x 4 rank tiles of 32768 x 16384; int64 accumulation and river
    index, since the raster has 2^32 cells); everything else is the weak-scaling series of 16384^2 tiles per GPU
    (N = 1: configs[2], the 16384^2 DEM the metric is quoted on; N = 4: configs[3], 32768^2 as 2 x 2)."""
    if global_dem is None and size is None and gpus == 8:
        global_dem = "65536x65536"
    return (16384 if size is None else size), global_dem


def tiled_layout(world, size, global_dem):
    """the rank grid of an N > 1 run: (tiling.Layout, description).  global_dem "HxW": that raster cut into the most
    nearly square ty x tx grid of equal rank tiles; otherwise one size x size tile per rank."""
    from descriptools_amd import tiling
    if global_dem:
        Hg, Wg = (int(v) for v in global_dem.lower().split("x"))
        grid = tiling.Layout.uniform(world, 64, 64)
        if Hg % (64 * grid.ty) or Wg % (64 * grid.tx):
            sys.exit("bench.py: --global %s does not cut into %d x %d rank tiles on the 64-cell grid"
                     % (global_dem, grid.ty, grid.tx))
        layout = tiling.Layout.uniform(world, Hg // grid.ty, Wg // grid.tx)
        series = "global raster %dx%d%s" % (Hg, Wg, " = BASELINE.json configs[4]" if (Hg, Wg, world) == (65536, 65536, 8)
                                            else " (--global)")
        return layout, series
    return (tiling.Layout.uniform(world, size, size),
            "weak-scaling series: one %dx%d tile per GPU (--size)" % (size, size))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=None, help="tile edge per GPU: every rank owns one SIZE x SIZE tile (the "
                    "weak-scaling series; default 16384 -- except --gpus 8, whose default is BASELINE.json configs[4]: "
                    "see --global)")
    ap.add_argument("--global", dest="global_dem", default=None, metavar="HxW", help="N > 1: the GLOBAL raster, cut into "
                    "the most nearly square ty x tx grid of equal rank tiles.  Default for --gpus 8 (when --size is not "
                    "given): 65536x65536 = BASELINE.json configs[4], 2 x 4 rank tiles of 32768 x 16384")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=3584, help="edge of the CPU baseline's DEM (a bounded sample: ~20 s "
                    "on one thread)")
    ap.add_argument("--no-verify", action="store_true", help="skip the cross-check of the timed step's rasters")
    ap.add_argument("--no-placement", action="store_true", help="rasters in allocation order (no placement tuning)")
    ap.add_argument("--placement", default="search", choices=["off", "own", "search"], help="which block of device memory "
                    "serves which raster (descriptools_amd/placement.py): search (default here; an opt-in set-up option "
                    "of a long-lived chain: bounded transient allocations until blocks of several write-conflict classes "
                    "are found, seconds reported in config.placement.setup_s), own (the library's default: only the "
                    "blocks the chain allocates anyway), off")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end_to_end block (host-tier API, PCIe included)")
    ap.add_argument("--e2e-size", type=int, default=16384, help="edge of the host DEM of the end_to_end block")
    ap.add_argument("--real-rep", type=int, default=8, help="end_to_end.real_terrain: the bundled Example tiled REP x REP "
                    "(8: 214 M cells)")
    ap.add_argument("--graph", action="store_true", help="N = 1: the headline loop replays the step as one HIP graph "
                    "launch (chain.Chain.capture) instead of ~45 kernel launches")
    ap.add_argument("--no-overlap", action="store_true", help="headline loop on ONE stream, kernels back to back "
                    "(default: downslope as a second branch on its own stream beside flow accumulation / HAND, "
                    "Chain(overlap=True), the fastest correct schedule; per-op times always come from a second, "
                    "serial timed loop)")
    ap.add_argument("--overlap", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--tiled", action="store_true", help="N = 1 through the multi-rank path (1 x 1 layout): "
                                                         "measures what the tiling machinery costs")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and use the "
                    "collective path even with one rank (rehearsal of the RCCL calls on a 1-GPU box)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; default) or gloo (rehearsal of N > 1 "
                                                        "with several ranks sharing one GPU)")
    args = ap.parse_args()
    args.overlap = not args.no_overlap
    args.size, args.global_dem = default_workload(args.gpus, args.size, args.global_dem)
    args.tune = False if (args.no_placement or args.placement == "off") else (True if args.placement == "own" else "search")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or "
                 "without a launcher)" % (args.gpus, world, args.gpus))
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
    stream = torch.cuda.Stream(device=dev, priority=-1 if args.overlap else 0)  # overlap: the main branch first
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = Context(device=local_rank, stream=stream.cuda_stream)
    # second branch of the chain (downslope beside flow accumulation / HAND, chain.Chain.run): its own stream
    stream2 = torch.cuda.Stream(device=dev) if args.overlap else stream
    ctx2 = Context(device=local_rank, stream=stream2.cuda_stream) if args.overlap else ctx

    def alloc(shape, dt):
        tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8,
               np.int32: torch.int32}[dt]
        return torch.empty(shape, dtype=tdt, device=dev)

    bufs = {}  # device pointer -> tensor

    def alloc_ptr(shape, dt):
        t = alloc(shape, dt)
        bufs[t.data_ptr()] = t
        return t.data_ptr()

    def release_ptr(q):
        del bufs[q]

    dem = alloc((H, W), np.float32)
    _lib.check(L.dt_dev_synth_dem(ctx.h, args.seed, H, W, 0, 0, H, W, 0, dem.data_ptr()))
    # the chain hands its blocks to the rasters by measured write-conflict class (descriptools_amd/placement.py): the
    # rasters one kernel writes together must not all lie in one class of the device's memory
    ch = chain.Chain(H, W, ctx=ctx, px=10.0, river_threshold=(H * W) // 512, alloc=alloc_ptr, release=release_ptr,
                     side_ctx=ctx2 if args.overlap else None, overlap=args.overlap, want_slope_rad=False,
                     tune_placement=args.tune)
    torch.cuda.empty_cache()  # the candidates the chain did not keep go back to the device
    if (ch.placement or {}).get("spacer_GiB"):
        # the runtime defers the release of what the search allocated: the next allocation of >= 2 GiB would wait for it
        # (seconds) -- take that wait here, as part of the chain's set-up, not inside a timed region further down
        ctx.empty((2 << 30,), np.uint8).free()
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int8: torch.int8, np.int32: torch.int32}
    rasters = {name: bufs[ch.p(name)].view(tdt[dt]) for name, dt in chain.OUTPUTS}
    N = H * W

    def barrier():
        torch.cuda.synchronize()

    # ---- headline: K steps of THE step -- chain.Chain.run, the schedule the tests validate (downslope as a second
    # branch unless --no-overlap), or its HIP graph replay ----
    def step():
        ch.run(dem.data_ptr(), want_a_river=False)

    for _ in range(args.warmup):
        step()
    barrier()
    graph = None
    if args.graph:
        graph = ch.capture(dem.data_ptr(), want_a_river=False)
        for _ in range(args.warmup):
            graph.launch()
        barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if graph is not None:
            graph.launch()
        else:
            step()
    barrier()
    dt = time.perf_counter() - t0

    # ---- second timed loop, same process: the same ops on ONE stream, back to back, each bracketed by HIP events on
    # that stream -> per-op / per-kernel durations that mean something (chain.Chain.ops(serial=True)) ----
    calls = ch.ops(dem.data_ptr(), want_a_river=False, serial=True)
    assert [n for n, _, _ in calls] == [n for n, _, _ in chain.OPS]
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in calls]
          for _ in range(args.steps)]
    for _, _, fn in calls:
        _lib.check(fn())
    barrier()
    t1 = time.perf_counter()
    for k in range(args.steps):
        for i, (name, octx, fn) in enumerate(calls):
            ev[k][i][0].record(stream)
            _lib.check(fn())
            ev[k][i][1].record(stream)
    barrier()
    dt_serial = time.perf_counter() - t1
    per_op, roof = per_op_table(ev, chain.OPS, N, S, args.steps)

    verified = None if args.no_verify else verify_step(torch, L, _lib, ctx, dem, rasters, ch, H, W)

    # practical HBM ceiling of this device: the better of two copies through the same library (float4
    # grid-stride; 1024 x 4 patches, whose pieces spread a workgroup over more HBM channels)
    src, dst = alloc((H, W), np.float32), alloc((H, W), np.float32)
    copy_gbs = 0.0
    for blocks in (65536, -1):
        if blocks < 0 and N % 65536 != 0:
            continue
        for _ in range(2):
            _lib.check(L.dt_dev_membench_copy(ctx.h, src.data_ptr(), dst.data_ptr(), N, blocks))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            _lib.check(L.dt_dev_membench_copy(ctx.h, src.data_ptr(), dst.data_ptr(), N, blocks))
        e1.record(stream)
        torch.cuda.synchronize()
        copy_gbs = max(copy_gbs, N * 8 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst

    sched = ("downslope on a second stream beside flow accumulation / HAND (Chain(overlap=True))" if args.overlap
             else "one stream, kernels back to back (--no-overlap)")
    value = N * args.steps / dt / 1e6
    out = {
        "metric": "Mcells/s full descriptor chain", "value": round(value, 1), "unit": "Mcells/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d synthetic tilted-integer-fBm DEM per GPU, full chain "
                               "(d8, flowacc, river mask, flowhand/HAND with fused GFI + ln(hl/H), fused slope+TI+MTI, "
                               "downslope), device-resident" % (S, S),
                   "global_dem": "%dx%d" % (H, W), "px": 10.0, "river_threshold_cells": ch.river_threshold,
                   "parallelism": "single GPU; " + sched
                                  + ("; the step replayed as one HIP graph (--graph)" if args.graph else ""),
                   "placement": ch.placement},
        "ms_per_step_serial": round(dt_serial / args.steps * 1e3, 3),
        "roofline": roof,
        "per_op": per_op,
        "verified": verified,
        "hbm_copy_ceiling_GBs": round(copy_gbs, 1),
        "chain_algo_bytes_per_cell": chain.ALGO_BYTES_PER_CELL,
        "chain_frac_of_hbm_peak": round(N * args.steps * chain.ALGO_BYTES_PER_CELL / dt / 1e9 / HBM_PEAK_GBS, 4),
    }
    if graph is not None:
        graph.free()
    # free the chain before the host-side blocks (they need host and device memory of their own)
    rasters.clear()
    bufs.clear()
    ch.free()
    torch.cuda.empty_cache()
    if not args.no_e2e:
        out["end_to_end"] = end_to_end(args.e2e_size, args.seed, args.real_rep)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n)
    print(json.dumps(out), flush=True)


def verify_step(torch, L, _lib, ctx, dem, rasters, ch, H, W):
    """After the timed loop: the rasters the timed step left behind against INDEPENDENT computations of the same
    quantities -- the first-generation global kernels (dt_set_flow_impl(1): in-degree countdown over all cells,
    raster-wide pointer doubling; different algorithms, same definitions) for flow accumulation / river index /
    flow distance / HAND, the unfused kernels for slope, TI / MTI, GFI / ln(hl/H), conservation of the
    accumulation -- and one checksum (sum of the raster's 32-bit words) per raster.  Raises on a mismatch."""
    N = H * W
    dev = dem.device
    f32 = lambda: torch.empty((H, W), dtype=torch.float32, device=dev)  # noqa: E731
    i32 = lambda: torch.empty((H, W), dtype=torch.int32, device=dev)  # noqa: E731
    c = ctx.h
    res = {}
    # flow accumulation + HAND by the global kernels
    fac2, fd2, idx2, hand2 = i32(), f32(), i32(), f32()
    _lib.check(L.dt_set_flow_impl(1))
    try:
        # downslope by the one-thread-per-cell walk on global memory (fd2 as the temporary)
        _lib.check(L.dt_dev_downslope(c, dem.data_ptr(), rasters["fdr"].data_ptr(), H, W, ch.px, ch.dz, 0,
                                      fd2.data_ptr()))
        ctx.sync()
        if not torch.equal(rasters["down"].view(torch.int32), fd2.view(torch.int32)):
            raise SystemExit("bench.py: timed step's downslope differs from the global walk")
        _lib.check(L.dt_dev_flowacc(c, rasters["fdr"].data_ptr(), dem.data_ptr(), H, W, fac2.data_ptr()))
        _lib.check(L.dt_dev_flowhand(c, dem.data_ptr(), rasters["fdr"].data_ptr(), rasters["river"].data_ptr(), None,
                                     H, W, ch.px, fd2.data_ptr(), idx2.data_ptr(), hand2.data_ptr(), None))
    finally:
        _lib.check(L.dt_set_flow_impl(2))
    ctx.sync()
    for name, other in (("fac", fac2), ("fdist", fd2), ("idx", idx2), ("hand", hand2)):
        if not torch.equal(rasters[name], other):
            raise SystemExit("bench.py: timed step's %s differs from the global-kernel computation" % name)
    res["fac_idx_fdist_hand_downslope_vs_global_kernels"] = "equal"
    if not torch.equal(rasters["river"], (rasters["fac"] > ch.river_threshold).to(torch.int8)):
        raise SystemExit("bench.py: river mask != fac > threshold")
    # unfused slope / radians / TI / MTI and GFI / ln(hl/H)
    del fac2, fd2, hand2
    sl2, rad2, ti2, mti2 = f32(), f32(), f32(), f32()
    _lib.check(L.dt_dev_slope_d8(c, dem.data_ptr(), H, W, ch.px, sl2.data_ptr(), None, rad2.data_ptr()))
    _lib.check(L.dt_dev_twi(c, rasters["fac"].data_ptr(), rad2.data_ptr(), N, ch.px, ch.n_top, ti2.data_ptr(),
                            mti2.data_ptr()))
    ctx.sync()
    if not torch.equal(rasters["slope"], sl2):
        raise SystemExit("bench.py: fused slope differs from the slope-only kernel")

    def close(a, b, what):
        err = (a.double() - b.double()).abs()
        if not bool((err <= 1e-5 * b.double().abs() + 1e-6).all()):
            raise SystemExit("bench.py: %s differs from the unfused kernel beyond 1e-5 relative" % what)
        return float(err.max())
    res["ti_mti_max_abs_diff_vs_unfused"] = [close(rasters["ti"], ti2, "TI"), close(rasters["mti"], mti2, "MTI")]
    del sl2, rad2
    ar = i32()
    flat_idx = rasters["idx"].reshape(-1).long().clamp(min=0)
    ar.reshape(-1).copy_(torch.where(rasters["idx"].reshape(-1) >= 0, rasters["fac"].reshape(-1)[flat_idx],
                                     torch.full_like(rasters["fac"].reshape(-1), -100)))
    del flat_idx
    _lib.check(L.dt_dev_gfi_lnhlh(c, rasters["hand"].data_ptr(), ar.data_ptr(), rasters["fac"].data_ptr(), N, ch.n_gfi,
                                  ch.b, ch.px, ti2.data_ptr(), mti2.data_ptr()))
    ctx.sync()
    res["gfi_lnhlh_max_abs_diff_vs_unfused"] = [close(rasters["gfi"], ti2, "GFI"), close(rasters["lnhlh"], mti2, "ln(hl/H)")]
    del ar, ti2, mti2, idx2
    # conservation: every cell drains to exactly one outlet (pit-free DEM without nodata)
    yy = torch.arange(H, device=dev, dtype=torch.int32).view(-1, 1)
    xx = torch.arange(W, device=dev, dtype=torch.int32).view(1, -1)
    dy = torch.zeros(256, dtype=torch.int32, device=dev)
    dx = torch.zeros(256, dtype=torch.int32, device=dev)
    for code, (a, b) in {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0),
                         128: (-1, 1)}.items():
        dy[code], dx[code] = a, b
    f = rasters["fdr"].long()
    ty, tx = yy + dy[f], xx + dx[f]
    outlet = (ty < 0) | (ty >= H) | (tx < 0) | (tx >= W) | (f == 0)
    del f, ty, tx
    drained = int((rasters["fac"][outlet].long() + 1).sum())
    if drained != N:
        raise SystemExit("bench.py: flow accumulation does not conserve cells (%d != %d)" % (drained, N))
    res["cells_drained_through_outlets"] = drained
    del outlet
    sums = {}
    for name, t in rasters.items():
        if name in ("a_river", "slope_rad"):
            continue  # not written by the timed step
        words = t.reshape(-1).view(torch.int32) if t.element_size() == 4 else t.reshape(-1).view(torch.uint8)
        sums[name] = int(words.sum(dtype=torch.int64))
    res["checksums"] = sums
    return res


def main_tiled(args, torch, dist, world, rank, local_rank, dev):
    """N > 1: one S x S core tile per rank of a (ty*S) x (tx*S) DEM; halo generated locally from the
    global generator; two RCCL all-gathers of ring summaries per step (flow accumulation, HAND).  Same line as N = 1:
    headline = tiling.run_rank (overlapped unless --no-overlap), then a serial loop on rank 0's stream for per_op /
    roofline, cross-checks of the rasters, cpu_baseline on rank 0."""
    from descriptools_amd import chain, tiling
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    layout, series = tiled_layout(world, args.size, args.global_dem)
    TH_, TW_ = layout.shape(rank)
    tile = tiling.RankTile(layout, rank, device=local_rank, stream=stream.cuda_stream, px=10.0,
                           river_threshold=(layout.Hg * layout.Wg) // 512, tune_placement=args.tune)
    tile.synth_dem(args.seed)

    use_dist = world > 1 or args.force_dist
    exchange = tiling.Exchange(tile, layout, world if not args.force_dist else max(world, 2))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    cpu_red = use_dist and dist.get_backend() == "gloo"
    red_dev = "cpu" if cpu_red else dev

    def allreduce(vals, op=None):
        t = torch.tensor(vals, dtype=torch.float64 if isinstance(vals[0], float) else torch.int64, device=red_dev)
        if use_dist:
            dist.all_reduce(t, op=op if op is not None else dist.ReduceOp.SUM)
        return t.cpu().tolist()

    for _ in range(args.warmup):
        tiling.run_rank(tile, layout, exchange, overlap=args.overlap)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tiling.run_rank(tile, layout, exchange, overlap=args.overlap)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        dt = allreduce([dt], dist.ReduceOp.MAX)[0]
    status = tile.ctx.status() & 1
    unres = allreduce([tile.unresolved_downslope(), status])

    # ---- serial loop: the stages of rank_ops() back to back on the context's stream, HIP events around each ----
    ops = tiling.rank_ops(tile, layout, exchange)
    kern = {n: k for n, _, k in chain.OPS}
    op_defs = []
    for name, bpc in tiling.RANK_OPS:
        k = {"d8": kern["d8"], "downslope": ["k_downslope_win_r"], "slope_twi": kern["slope_twi"],
             "flowacc_local": ["k_fa_tile1", "k_fa_reduce", "k_fa_nxt_init", "k_fa_rank_summary"],
             "flowacc_finish_flowhand_local": ["k_rk_fa_*", "k_fa_propagate", "k_fa_poison", "k_fa3fh1", "k_fh_tile1",
                                               "k_fh_ghost_init", "k_fh_node_jump", "k_fh_rank_summary"],
             "flowhand_gfi_solve_finish": ["k_rk_fh_*", "k_fh_ghost_set", "k_fh_tile3"]}.get(name, [])
        op_defs.append((name, bpc, k))
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ops]
          for _ in range(args.steps)]
    for _, fn in ops:
        fn()
    barrier()
    t1 = time.perf_counter()
    for k in range(args.steps):
        for i, (_, fn) in enumerate(ops):
            ev[k][i][0].record(tile.ts)
            fn()
            ev[k][i][1].record(tile.ts)
    barrier()
    dt_serial = time.perf_counter() - t1
    if use_dist:
        dt_serial = allreduce([dt_serial], dist.ReduceOp.MAX)[0]
    per_op, roof = per_op_table(ev, op_defs, TH_ * TW_, -1, args.steps)  # rank 0's own stream and tile
    roof["note"] += "; N > 1: rank 0's tile, the exchange stages (0 bytes) are the all-gathers' launch and wait"

    verified = None if args.no_verify else verify_tiled(torch, tile, layout, allreduce)
    cells = layout.Hg * layout.Wg
    sched = ("downslope as a second branch on its own stream" if args.overlap else "one stream per rank (--no-overlap)")
    out = {
        "metric": "Mcells/s full descriptor chain", "value": round(cells * args.steps / dt / 1e6, 1),
        "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%dx%d synthetic tilted-integer-fBm DEM per GPU, full chain (d8, flowacc, river mask, "
                               "flowhand/HAND with fused GFI + ln(hl/H), fused slope+TI+MTI, downslope), "
                               "device-resident" % (TH_, TW_),
                   "global_dem": "%dx%d" % (layout.Hg, layout.Wg), "series": series,
                   "rank_tile": "%dx%d" % (TH_, TW_), "px": 10.0,
                   "river_index_dtype": "int64" if tile.idx_dtype == torch.int64 else "int32",
                   "river_threshold_cells": tile.river_threshold,
                   "accumulation_dtype": "int64" if tile.acc64 else "int32",
                   "placement": tile.placement,
                   "parallelism": "%dx%d rank tiles, 64-cell halo, 2 %s all-gathers of ring summaries per "
                                  "step (flow accumulation inflow, HAND rank exits); %s"
                                  % (layout.ty, layout.tx, "gloo (CPU rehearsal)" if cpu_red else "RCCL", sched)},
        "ms_per_step_serial": round(dt_serial / args.steps * 1e3, 3),
        "roofline": roof,
        "per_op": per_op,
        "verified": verified,
        "chain_algo_bytes_per_cell": chain.ALGO_BYTES_PER_CELL,
        "chain_frac_of_hbm_peak": round(cells * args.steps * chain.ALGO_BYTES_PER_CELL / dt / 1e9 / world
                                        / HBM_PEAK_GBS, 4),
        "downslope_walks_beyond_halo": int(unres[0]),
        "accumulation_overflow_ranks": int(unres[1]),  # ranks whose int32 accumulation may have reached 2^31
        "backend": (dist.get_backend() if use_dist else "none"),
        "ranks": (dist.get_world_size() if use_dist else 1),
        "distinct_gpus": min(world, torch.cuda.device_count()) if not cpu_red else 1,
    }
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_n)
    if use_dist:
        dist.barrier()  # the other ranks wait for rank 0's CPU baseline
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def verify_tiled(torch, tile, layout, allreduce):
    """the rasters the last step left on this rank: conservation over the GLOBAL raster (every cell drains through
    exactly one outlet), river mask == fac > threshold, river cells index themselves at distance 0, HAND >= 0 where
    a river cell was found, one checksum per raster summed over the ranks (for the 1 x 1 layout they are the N = 1
    run's checksums); raises on a violation"""
    Hg, Wg = layout.Hg, layout.Wg
    dev = tile.dev
    fdr, fac, river, idx, fdist, hand = (tile.core(n) for n in ("fdr", "fac", "river", "idx", "fdist", "hand"))
    tile.side_ctx.sync()
    tile.ctx.sync()
    dy = torch.zeros(256, dtype=torch.int64, device=dev)
    dx = torch.zeros(256, dtype=torch.int64, device=dev)
    for code, (a, b) in {1: (0, 1), 2: (1, 1), 4: (1, 0), 8: (1, -1), 16: (0, -1), 32: (-1, -1), 64: (-1, 0),
                         128: (-1, 1)}.items():
        dy[code], dx[code] = a, b
    gy = torch.arange(tile.gy0, tile.gy0 + tile.H, device=dev, dtype=torch.int64).view(-1, 1)
    gx = torch.arange(tile.gx0, tile.gx0 + tile.W, device=dev, dtype=torch.int64).view(1, -1)
    f = fdr.long()
    ty, tx = gy + dy[f], gx + dx[f]
    outlet = (ty < 0) | (ty >= Hg) | (tx < 0) | (tx >= Wg) | (f == 0)
    drained = int((fac[outlet].long() + 1).sum())
    del f, ty, tx, outlet
    ok = bool(torch.equal(river, (fac > tile.river_threshold).to(torch.int8)))
    rv = river == 1
    lin = gy * Wg + gx
    ok = ok and bool(torch.equal(idx.long()[rv], lin[rv])) and (not bool(rv.any()) or float(fdist[rv].abs().max()) == 0.0)
    found = idx >= 0
    ok = ok and bool((hand[found] >= 0).all()) and bool((hand[~found] == -100).all())
    del rv, lin, found
    sums = {}
    for name in ("slope", "fdr", "fac", "river", "fdist", "idx", "hand", "ti", "mti", "gfi", "lnhlh", "down"):
        t = tile.core(name)
        if t.dtype == torch.int64:   # value sums: independent of the raster's width
            sums[name] = int(t.sum())
        elif t.element_size() == 4:
            sums[name] = int(t.view(torch.int32).sum(dtype=torch.int64))
        else:
            sums[name] = int(t.sum(dtype=torch.int64))
    names = sorted(sums)
    tot = allreduce([drained, 0 if ok else 1] + [sums[n] for n in names])
    if int(tot[0]) != Hg * Wg:
        raise SystemExit("bench.py: flow accumulation does not conserve cells over the ranks (%d != %d)" % (tot[0], Hg * Wg))
    if int(tot[1]) != 0:
        raise SystemExit("bench.py: river mask / river index / HAND consistency violated on %d ranks" % tot[1])
    return {"cells_drained_through_outlets": int(tot[0]), "river_idx_hand_consistent": True,
            "checksums": {n: int(v) for n, v in zip(names, tot[2:])}}


def end_to_end(n, seed, real_rep=8):
    """What a user of the reference's API sees (SURVEY.md 8d "end-to-end incl. H2D/D2H, reported separately"; never
    `value`): (a) chain.run_host on an n x n HOST DEM -- one H2D of the DEM, the resident chain, 13 rasters back over
    PCIe into page-locked memory -- with the split; (b) the reference's call sequence (Example/example.py:59-91) through
    the drop-in functions, pageable numpy in and out, the dtypes of the reference (float64 / int64 containers);
    (c) examples/example.py on the bundled rasters (config #1)."""
    import torch
    from descriptools_amd import _lib, chain, downslope, flowhand, gfi, slope, topoindexes
    from descriptools_amd.device import Context
    L = _lib.lib()
    px, thr = 10.0, (n * n) // 512
    ctx = Context()
    d = ctx.empty((n, n), np.float32)
    _lib.check(L.dt_dev_synth_dem(ctx.h, seed, n, n, 0, 0, n, n, 0, d.ptr))
    dem = d.to_host()
    # (a) split of run_host: H2D, kernels, D2H
    ch = chain.Chain(n, n, ctx=ctx, px=px, river_threshold=thr, want_slope_rad=False)
    t0 = time.perf_counter()
    d.copy_from(dem)
    t1 = time.perf_counter()
    ch.run(d.ptr, want_a_river=False)
    ctx.sync()
    t2 = time.perf_counter()
    outs = {k: ch.buf[k].to_host_async() for k, _ in chain.OUTPUTS if k not in ("a_river", "slope_rad")}
    ctx.sync()
    t3 = time.perf_counter()
    nbytes_out = sum(a.nbytes for a in outs.values())
    fdr, fac = outs["fdr"].copy(), outs["fac"].astype(np.int64)
    del outs
    ch.free()
    d.free()
    ctx.close()
    t4 = time.perf_counter()
    rh_t = {}
    rh = chain.run_host(dem, px, timings=rh_t, river_threshold=thr, want_slope_rad=False)  # warm pools: steady state
    t5 = time.perf_counter()
    del rh
    # (b) the reference's call sequence through the drop-in API
    river = (fac > thr).astype(np.int8)
    lib_s = 0.0

    def timed(fn, *a):
        nonlocal lib_s
        ta = time.perf_counter()
        r = fn(*a)
        lib_s += time.perf_counter() - ta
        return r
    t6 = time.perf_counter()
    sl = timed(slope.sloper, dem, px).astype("float32")
    slr = np.arctan(sl / 100).astype("float32")
    slr = np.where(dem == -100, -100, slr)
    ti, mti = timed(topoindexes.topographic_index, fac, slr, px, 0.1)
    down = timed(downslope.downsloper, dem, fdr, px, 5)
    flow, indices, hand = timed(flowhand.flow_hand_index, dem, fdr, river, px)
    geofi = timed(gfi.gfi_calculator, hand, fac, indices, 0.4, 0.1, px)
    lnhlh = timed(gfi.ln_hl_H_calculator, hand, fac, 0.4, 0.1, px)
    t7 = time.perf_counter()
    del sl, slr, ti, mti, down, flow, indices, hand, geofi, lnhlh, river, fdr, fac, dem
    from descriptools_amd import device
    device.trim()
    torch.cuda.empty_cache()
    # (c) config #1: the headless example on the bundled rasters
    ex = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "examples"))
        import contextlib
        import io
        import example as ex_mod
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ex = ex_mod.main()
    except Exception as e:  # the example's rasters are test fixtures: report, do not fail the bench
        ex = {"error": repr(e)}
    cells = n * n
    try:
        real = real_terrain(real_rep, 4096 if real_rep >= 8 else 1024)
    except Exception as e:  # fixtures / memory: report, do not fail the bench
        real = {"error": repr(e)}
    return {
        "real_terrain": real,
        "size": "%dx%d host DEM" % (n, n),
        "run_host_split": {"h2d_dem_pageable_ms": round((t1 - t0) * 1e3, 1), "kernels_ms": round((t2 - t1) * 1e3, 1),
                           "d2h_11_rasters_pinned_ms": round((t3 - t2) * 1e3, 1), "d2h_GBs": round(nbytes_out / (t3 - t2) / 1e9, 1),
                           "note": "first call of the process: includes locking the host pages of the 11 outputs"},
        "run_host": {"seconds": round(t5 - t4, 3), "Mcells_s": round(cells / (t5 - t4) / 1e6, 1), "phases": rh_t,
                     "note": "chain.run_host, numpy DEM in, 13 numpy rasters out (fac / idx widened to int64), warm "
                             "page-locked pool"},
        "dropin_api": {"seconds": round(t7 - t6, 3), "Mcells_s": round(cells / (t7 - t6) / 1e6, 1),
                       "library_calls_s": round(lib_s, 3), "caller_numpy_s": round(t7 - t6 - lib_s, 3),
                       "note": "sloper, arctan (host numpy, example.py:63), topographic_index, downsloper, "
                               "flow_hand_index, gfi_calculator, ln_hl_H_calculator: pageable numpy in / out in the "
                               "reference's dtypes, one H2D / D2H round trip per call as the reference's *_cpu shims; "
                               "library_calls_s = inside the six library functions, caller_numpy_s = the script's own "
                               "numpy lines between them (astype / arctan / where, example.py:63)"},
        "example": ex,
    }


def real_terrain(rep=8, rough_n=4096):
    """The reference's only workload is real terrain with a GIS D8 raster (Example/example.py:33-39); the headline's
    synthetic DEM has no flats.  (a) the bundled Example tiled rep x rep (214 M cells at rep = 8) with its GIS D8 codes
    through the resident chain (Chain(external_fdr=True, long_walks="auto")): per-op ms on one stream, the number of
    long downslope walks queued, the step with the long walks finished.  (b) a rough synthetic rough_n^2 DEM (noise,
    pits, integer plateaus, nodata) through the CONDITIONED chain (depression filling + flat routing + D8, then
    everything else; Chain(condition=True, long_walks=True)) beside its unconditioned chain."""
    import torch
    import descriptools_amd.rasterio_lite as rio
    from descriptools_amd import _lib, chain
    from descriptools_amd.device import Context
    L = _lib.lib()
    ex = os.path.join(ROOT, "tests", "golden", "example")
    dem0, _ = rio.read_masked(os.path.join(ex, "12_dem.tif"), -100, "int16")
    fdr0 = np.ascontiguousarray(rio.read(os.path.join(ex, "12_fdr.tif"))[0], np.uint8)
    dem = np.tile(dem0.astype(np.float32), (rep, rep))
    fdr = np.tile(fdr0, (rep, rep))
    H, W = dem.shape
    N, px = H * W, 12.5
    st = torch.cuda.Stream()
    ctx = Context(0, st.cuda_stream)
    out = {}
    with torch.cuda.stream(st):
        d = ctx.to_device(dem)
        ch = chain.Chain(H, W, ctx=ctx, px=px, river_threshold=N // 512, want_slope_rad=False, overlap=False,
                         tune_placement=False, long_walks="auto", external_fdr=True)
        ch.buf["fdr"].copy_from(fdr)
        ops = ch.ops(d.ptr, want_a_river=False, serial=True)

        def step():
            for _, _, fn in ops:
                _lib.check(fn())
            return ch.finish_long_walks()
        step()
        ctx.sync()
        reps = 3
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(len(ops) + 1)]
              for _ in range(reps)]
        t0 = time.perf_counter()
        queued = 0
        for k in range(reps):
            for i, (_, _, fn) in enumerate(ops):
                ev[k][i][0].record(st)
                _lib.check(fn())
                ev[k][i][1].record(st)
            ev[k][-1][0].record(st)
            queued = ch.finish_long_walks()   # a synchronisation point: looks at the queue, builds the skip tables
            ev[k][-1][1].record(st)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        names = [n for n, _, _ in ops] + ["downslope_long_walks_finish"]
        per_op = {n: round(float(np.mean([ev[k][i][0].elapsed_time(ev[k][i][1]) for k in range(reps)])), 3)
                  for i, n in enumerate(names)}
        valid = int((dem != -100).sum())
        out["example_tiled"] = {
            "raster": "%dx%d = bundled Example x %d x %d, GIS D8 codes (Example/input/12_fdr.tif)" % (H, W, rep, rep),
            "cells": N, "valid_cells": valid, "ms_per_step": round(dt * 1e3, 3), "Mcells_s": round(N / dt / 1e6, 1),
            "per_op_ms": per_op, "downslope_walks_queued": int(queued),
            "note": "resident chain without a D8 op (codes given), one stream; downslope queues its long walks "
                    "(valley floors, flats: thousands of moves) and finish_long_walks() crosses them with skip tables"}
        ch.free()
        d.free()
        del dem, fdr
        # (b) the conditioned chain on rough terrain
        g = ctx.empty((rough_n, rough_n), np.float32)
        _lib.check(L.dt_dev_synth_dem(ctx.h, 3, rough_n, rough_n, 0, 0, rough_n, rough_n, 2, g.ptr))
        base = g.to_host()
        rng = np.random.default_rng(3)
        nod = base == -100
        rough = np.floor(base + rng.normal(0, 6.0, base.shape).astype(np.float32)).astype(np.float32)
        rough[rng.random(rough.shape) < 0.02] -= 40
        rough[nod] = -100
        g.copy_from(rough)
        times = {}
        for cond in (False, True):
            c2 = chain.Chain(rough_n, rough_n, ctx=ctx, px=10.0, condition=cond, condition_rounds=96, tune_placement=False,
                             long_walks=cond, overlap=False)
            for _ in range(2):
                c2.run(g.ptr)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                c2.run(g.ptr)
            ctx.sync()
            times[cond] = (time.perf_counter() - t0) / 5
            c2.check_status()
            c2.free()
        out["rough_conditioned"] = {
            "raster": "%dx%d synthetic DEM + noise, pits, integer plateaus, 2 %% nodata" % (rough_n, rough_n),
            "chain_ms": round(times[False] * 1e3, 3), "conditioned_chain_ms": round(times[True] * 1e3, 3),
            "note": "conditioned = depression filling + flat routing + D8 on the filled surface (dt_dev_condition_d8_async), "
                    "then the flow kernels, descriptors and downslope with the long-walk workspace"}
        g.free()
    ctx.close()
    torch.cuda.empty_cache()
    return out


def cpu_baseline(n=3584, seeds=(1,)):
    """the reference's CPU path is its single-threaded `*_sequential_jit` family (Numba is not in
    this image), so the baseline is the oracle -- the same per-cell algorithms restated in C -- built with
    gcc -O3 -march=native, timed on the whole chain over an n x n DEM of the same generator: ONE thread (`value`),
    and with OpenMP over the per-cell loops on all cores (`all_cores`, median of three; the in-degree-countdown flow
    accumulation stays sequential).  A bounded sample (~20 + ~10 s), baseline only."""
    import oracle
    px = 10.0

    def run(seed):
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
        return time.perf_counter() - t0

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        oracle.use_bench_build(1)
        t1 = sorted(run(s) for s in seeds)
        oracle.use_bench_build(ncpu)
        tn = sorted(run(s) for s in (1, 2, 3))
    finally:
        oracle.use_bench_build(None)
    med1, medn = t1[len(t1) // 2], tn[len(tn) // 2]
    return {"value": round(n * n / med1 / 1e6, 3), "unit": "Mcells/s", "cores": 1, "kind": "port",
            "all_cores": {"value": round(n * n / medn / 1e6, 3), "unit": "Mcells/s", "cores": ncpu,
                          "nproc": os.cpu_count()},
            "sample": "full chain on a %dx%d DEM of the same generator (the 16384^2 workload's generator, a bounded "
                      "sample): seed %s on 1 thread (%.1f s), median of seeds 1-3 on %d threads (%.1f s); "
                      "oracle/dt_oracle.c, gcc -O3 -march=native (+ OpenMP over the per-cell loops for all_cores); "
                      "reference-algorithm restatement, not Numba"
                      % (n, n, list(seeds), med1, ncpu, medn)}


if __name__ == "__main__":
    main()
