"""GPU (-m gpu): tiling.run_rank -- the step the N > 1 bench and the 8-GPU driver run execute (side-stream
all-gathers behind events on the context stream, dt_dev_rank_solve_*, fused finish) -- driven by real
torch.distributed ranks in child processes that share the one GPU, and compared raster by raster with the untiled
chain.  gloo carries the collectives between the ranks (RCCL refuses two ranks on one device); the RCCL calls
themselves run in the one-rank case."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["fdr", "fac", "river", "fdist", "idx", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(tmp_path, world, h, w, nparts=None, **kw):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_rank_child.py"), "--out", str(tmp_path), "--h", str(h), "--w", str(w)]
    for k, v in kw.items():
        cmd += ["--" + k.replace("_", "-"), str(v)]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(nparts or world)]


def _check(parts, Hg, Wg, seed, nodata, dem=None, **chain_kw):
    from descriptools_amd import chain
    if dem is None:
        dem = oracle.synth_dem(seed, Hg, Wg, 0, 0, Hg, Wg, nodata)
    thr = (Hg * Wg) // 512
    ref = chain.run_host(dem, 10.0, river_threshold=thr, **chain_kw)
    if Hg * Wg <= 1 << 20 and not chain_kw.get("condition"):
        # ... and the untiled chain against the ORACLE on this very DEM, so that "the ranks equal the chain" means "the
        # ranks equal the reference's algorithms" in this test too (the conditioned chain: tests/test_gpu_hydro.py)
        sl_o, fdr_o = oracle.slope_d8(dem, 10.0)
        acc_o = oracle.flowacc(fdr_o, dem)
        fd_o, idx_o, hand_o = oracle.flowhand(dem, fdr_o, (acc_o > thr).astype(np.int8), 10.0)
        assert np.array_equal(ref["fdr"], fdr_o) and np.array_equal(ref["slope"], sl_o) and np.array_equal(ref["fac"], acc_o)
        assert np.array_equal(ref["idx"], idx_o) and np.array_equal(ref["hand"], hand_o)
        assert np.array_equal(ref["fdist"] == -100, fd_o == -100) and np.allclose(ref["fdist"], fd_o, rtol=1e-6, atol=0)
        assert np.array_equal(ref["down"], oracle.downslope(dem, fdr_o, 10.0, 5.0))
    for r, p in enumerate(parts):
        y0, x0 = (int(v) for v in p["origin"])
        H, W = p["fdr"].shape
        sl = (slice(y0, y0 + H), slice(x0, x0 + W))
        assert np.array_equal(p["dem"], dem[sl])
        for n in NAMES:
            got, want = p[n], ref[n][sl]
            assert np.array_equal(got, want.astype(got.dtype), equal_nan=True), \
                "rank %d %s: %d cells differ" % (r, n, int((got != want).sum()))


@pytest.mark.parametrize("world,h,w,nodata,halo,overlap", [
    (2, 256, 192, 0, "synth", 0),        # 1 x 2
    (4, 192, 256, 2, "exchange", 0),     # 2 x 2, DEM halo exchanged point-to-point, nodata blobs
    (4, 128, 128, 0, "synth", 1),        # 2 x 2, downslope as a second branch
    (3, 128, 192, 3, "exchange", 0),     # 1 x 3 (at most 4 ranks: the box admits 6 GPU processes, pytest is one)
    (2, 2048, 1536, 1, "exchange", 1),   # 1 x 2 of 32 x 24 tiles each: rivers and HAND paths across a process border
])
def test_run_rank_gloo_ranks_sharing_the_gpu(tmp_path, world, h, w, nodata, halo, overlap):
    parts = _launch(tmp_path, world, h, w, seed=5, nodata=nodata, halo=halo, overlap=overlap)
    from descriptools_amd import tiling
    layout = tiling.Layout.uniform(world, h, w)
    _check(parts, layout.Hg, layout.Wg, 5, nodata)


def test_long_walks_travel_between_real_processes(tmp_path):
    """a 1 per mille plane: every downslope walk leaves its rank; tiling.finish_downslope with DistComm (gloo) between
    two real processes sends the walkers on -- all rasters equal the untiled chain's (chain.run_host finishes its own
    long walks with the skip tables)"""
    parts = _launch(tmp_path, 2, 256, 384, seed=1, nodata=0, terrain="plane")
    from descriptools_amd import tiling
    layout = tiling.Layout.uniform(2, 256, 384)
    yy, xx = np.mgrid[0:layout.Hg, 0:layout.Wg]
    _check(parts, layout.Hg, layout.Wg, 1, 0, dem=(200.0 - 0.001 * xx - 0.0002 * yy).astype(np.float32))


def test_conditioning_over_real_processes_then_the_rank_step_on_its_codes(tmp_path):
    """tiling.condition_rank over torch.distributed (halo exchanges of the filled surface / the flat
    distances / the codes between two processes, an all-reduce of the flag per iteration), then run_rank(d8=False) -- the
    step starts from the conditioned codes instead of overwriting them -- and the walkers; every raster equals the
    single raster's conditioned chain"""
    parts = _launch(tmp_path, 2, 320, 256, seed=6, terrain="rough", condition=1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _rank_child import rough_dem
    from descriptools_amd import tiling
    layout = tiling.Layout.uniform(2, 320, 256)
    dem = rough_dem(layout.Hg, layout.Wg, 6)
    _check(parts, layout.Hg, layout.Wg, 6, 0, dem=dem, condition=True, condition_rounds=256, long_walks=True)


def test_run_rank_rccl_single_rank(tmp_path):
    """the RCCL all-gathers on device tensors (side stream, events) with one rank: the collective path of the
    8-GPU run, minus the peers"""
    parts = _launch(tmp_path, 1, 320, 448, seed=3, nodata=2, backend="nccl", force_world=2)
    _check(parts, 320, 448, 3, 2)


def test_exchange_halo_over_rccl_to_itself_then_rank_ops(tmp_path):
    """tiling.exchange_halos on DEVICE tensors over RCCL: one process, 2 x 3 logical ranks with nodata, every send /
    receive pair of the 8-GPU run's point-to-point halo exchange posted to the process itself (torch allows isend to
    oneself; RCCL matches by posting order -- the order exchange_halos establishes); then the ranks run rank_ops(),
    the serial schedule bench.py times, in lock-step: every raster equals the untiled chain's"""
    parts = _launch(tmp_path, 1, 192, 128, nparts=6, seed=4, nodata=2, backend="nccl", logical=6)
    from descriptools_amd import tiling
    layout = tiling.Layout.uniform(6, 192, 128)
    assert (layout.ty, layout.tx) == (2, 3)
    _check(parts, layout.Hg, layout.Wg, 4, 2)


def test_exchange_halo_on_device_tensors():
    """the halo exchange's strip slicing / placement on DEVICE tensors, N logical ranks on one device (the
    transport is a device-to-device copy instead of isend / irecv, like simulate() for the all-gathers): every
    halo cell inside the global raster must equal the global DEM, ragged last row / column included."""
    import torch
    from descriptools_amd import tiling
    for heights, widths in (([128, 192], [192, 128, 64]), ([256], [128, 128]), ([128, 70], [64, 200])):
        layout = tiling.Layout(heights, widths)
        Hg, Wg = layout.Hg, layout.Wg
        dem = torch.as_tensor(oracle.synth_dem(9, Hg, Wg, 0, 0, Hg, Wg, 0), device="cuda")
        h = tiling.HALO
        exts = []
        for r in range(layout.size):
            y0, x0 = layout.origin(r)
            H, W = layout.shape(r)
            e = torch.full((H + 2 * h, W + 2 * h), float("nan"), device="cuda")
            e[h:h + H, h:h + W] = dem[y0:y0 + H, x0:x0 + W]
            exts.append(e)
        tiling.exchange_halo_local(exts, layout)
        pad = torch.full((Hg + 2 * h, Wg + 2 * h), float("nan"), device="cuda")
        pad[h:h + Hg, h:h + Wg] = dem
        for r in range(layout.size):
            y0, x0 = layout.origin(r)
            H, W = layout.shape(r)
            want = pad[y0:y0 + H + 2 * h, x0:x0 + W + 2 * h]
            assert torch.equal(torch.nan_to_num(exts[r], nan=-1.0), torch.nan_to_num(want, nan=-1.0)), (heights, widths, r)
