"""GPU (-m gpu): bench.py's one-line JSON contract at a small size -- the keys the driver reads, the roofline
and cpu_baseline objects, and the N = 1 run through the multi-rank path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "2048", "--steps", "2",
                          "--warmup", "1", *extra], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def _check_common(d, cells_per_gpu):
    """keys every bench line carries, N = 1 or N > 1"""
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_serial",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "per_op", "verified"):
        assert k in d, k
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"].startswith("k_") and r["op"] in d["per_op"]
    op = d["per_op"][r["op"]]
    assert op["ms"] > 0 and abs(op["achieved_GBs"] - cells_per_gpu * op["algo_bytes_per_cell"] / (op["ms"] * 1e-3) / 1e9) \
        <= 0.01 * op["achieved_GBs"] + 0.1
    # the serial loop is what the per-op events come from: they add up to (about) its step time
    assert sum(v["ms"] for v in d["per_op"].values()) <= 1.2 * d["ms_per_step_serial"] + 0.5
    assert set(d["verified"]["checksums"]) >= {"fdr", "fac", "river", "fdist", "idx", "hand", "slope", "ti", "mti", "gfi",
                                               "lnhlh", "down"}


def test_bench_line_contract():
    d = _run("--cpu-n", "512", "--e2e-size", "1024", "--real-rep", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "end_to_end"):
        assert k in d, k
    _check_common(d, 2048 * 2048)
    assert "second stream" in d["config"]["parallelism"]  # the headline is the overlapped schedule
    e = d["end_to_end"]
    assert e["run_host"]["Mcells_s"] > 0 and e["dropin_api"]["Mcells_s"] > 0 and e["run_host_split"]["kernels_ms"] > 0
    assert e["example"]["class_map_mismatches"] == 0 and e["example"]["threshold"] == 0.012
    rt = e["real_terrain"]  # real terrain (GIS D8 codes, long walks) and the conditioned chain
    assert "error" not in rt, rt
    assert rt["example_tiled"]["cells"] == 4 * 2178 * 1534 and rt["example_tiled"]["downslope_walks_queued"] > 0
    assert set(rt["example_tiled"]["per_op_ms"]) == {"downslope", "flowacc_flowhand_local", "slope_twi",
                                                      "flowhand_gfi_finish", "downslope_long_walks_finish"}
    assert rt["rough_conditioned"]["conditioned_chain_ms"] > rt["rough_conditioned"]["chain_ms"] > 0
    p = d["config"]["placement"]
    assert p["tuned"] is False or (p["setup_s"] >= 0 and p["spacer_GiB"] <= p["spacer_budget_GiB"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mcells/s" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2048 * 2048 / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"].startswith("k_")
    assert r["traffic"] is None  # the committed PMC summary is for 16384^2 only
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "Mcells/s" and c["value"] > 0 and c["sample"]
    assert c["all_cores"]["cores"] >= 1 and c["all_cores"]["value"] > 0 and "-O3 -march=native" in c["sample"]
    v = d["verified"]  # the timed step's rasters were cross-checked after the timed loop
    assert v["fac_idx_fdist_hand_downslope_vs_global_kernels"] == "equal" and v["cells_drained_through_outlets"] == 2048 * 2048
    assert set(v["checksums"]) >= {"fdr", "fac", "river", "fdist", "idx", "hand", "slope", "ti", "mti", "gfi", "lnhlh", "down"}


def test_bench_serial_headline_and_graph_replay():
    """--no-overlap: one stream; --graph: the headline loop replays a HIP graph and per-op times still come from the
    serial loop (they used to be zeros)"""
    d = _run("--no-overlap", "--no-cpu-baseline", "--no-e2e")
    _check_common(d, 2048 * 2048)
    assert "one stream" in d["config"]["parallelism"]
    g = _run("--graph", "--no-cpu-baseline", "--no-e2e")
    _check_common(g, 2048 * 2048)
    assert all(v["ms"] > 0 for v in g["per_op"].values()) and g["roofline"]["frac"] > 0
    assert g["verified"]["checksums"] == d["verified"]["checksums"]


def test_bench_tiled_path_at_one_rank():
    """the N > 1 line's shape, at one rank: per-op events on the rank's stream, roofline of the dominant kernel,
    cpu_baseline on rank 0, cross-checks; the 1 x 1 layout's checksums are the N = 1 run's"""
    d = _run("--tiled", "--cpu-n", "512")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["downslope_walks_beyond_halo"] == 0
    assert "rank tiles" in d["config"]["parallelism"] and d["ranks"] == 1 and d["backend"] == "none"
    _check_common(d, 2048 * 2048)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["verified"]["cells_drained_through_outlets"] == 2048 * 2048
    one = _run("--no-cpu-baseline", "--no-e2e")
    assert d["verified"]["checksums"] == one["verified"]["checksums"]
    f = _run("--force-dist", "--no-cpu-baseline")  # the RCCL collectives with one rank
    _check_common(f, 2048 * 2048)
    assert f["backend"] == "nccl" and f["ranks"] == 1 and f["verified"]["checksums"] == d["verified"]["checksums"]


def test_bench_gpus_flag_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun around it must not quietly run one rank: it launches the ranks
    as a child torch.distributed.run (here: gloo rehearsal, two ranks sharing the one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size",
                          "1024", "--steps", "2", "--warmup", "1", "--cpu-n", "256"], capture_output=True, text=True,
                         timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["distinct_gpus"] == 1 and d["value"] > 0 and d["ranks"] == 2
    assert d["config"]["global_dem"] == "1024x2048" and d["downslope_walks_beyond_halo"] == 0
    _check_common(d, 1024 * 1024)
    assert d["verified"]["cells_drained_through_outlets"] == 1024 * 2048 and "cpu_baseline" in d
