"""CPU: the committed rocprofv3 summaries are what bench.py attaches as roofline.traffic / per_op.traffic_bytes."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R1 = os.path.join(ROOT, "profiles", "r1")


def _pmc_file():
    src = open(os.path.join(ROOT, "bench.py")).read()
    name = src.split('PMC_FILE = os.path.join(ROOT, "profiles", "r1", "')[1].split('"')[0]
    return name


def test_pmc_traffic_json_is_derived_from_the_committed_counter_csvs(tmp_path):
    name = _pmc_file()
    tag = name.split("_pmc_traffic.json")[0]
    out = str(tmp_path / "t.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"),
                    os.path.join(R1, tag + "_pmc_fetch_counter_collection.csv"),
                    os.path.join(R1, tag + "_pmc_write_counter_collection.csv"), out], check=True,
                   capture_output=True)
    assert json.load(open(out)) == json.load(open(os.path.join(R1, name)))


def test_bench_op_kernels_are_in_the_pmc_summary_and_the_kernel_stats():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    d = json.load(open(os.path.join(R1, _pmc_file())))
    tag = _pmc_file().split("_pmc_traffic.json")[0]
    stats = [r["Name"] for r in csv.DictReader(open(os.path.join(R1, tag + "_bench16384_kernel_stats.csv")))]
    for op, kernels in bench.OP_KERNELS.items():
        assert bench.pmc_traffic(op, d["size"]) is not None, op
        for k in kernels:
            assert k in d["kernels"], k
            assert any(k.split("<")[0] in n for n in stats), k
    line = json.load(open(os.path.join(R1, tag + "_bench16384.json")))
    assert line["roofline"]["traffic"] == int(bench.pmc_traffic(line["roofline"]["op"], d["size"]))
