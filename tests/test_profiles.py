"""CPU: the committed rocprofv3 summaries of this round (profiles/r4/) are what bench.py attaches as
roofline.traffic / per_op.traffic_bytes, and DESIGN.md's measurement table is generated from them."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R2 = os.path.join(ROOT, "profiles", "r4")  # this round's set
sys.path.insert(0, ROOT)

needs_profiles = pytest.mark.skipif(not os.path.exists(os.path.join(R2, "pmc_traffic.json")),
                                    reason="profiles/r3 not collected yet")


@needs_profiles
def test_pmc_traffic_json_is_derived_from_the_committed_counter_csvs(tmp_path):
    out = str(tmp_path / "t.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"),
                    os.path.join(R2, "pmc_fetch_counter_collection.csv"),
                    os.path.join(R2, "pmc_write_counter_collection.csv"), out], check=True, capture_output=True)
    assert json.load(open(out)) == json.load(open(os.path.join(R2, "pmc_traffic.json")))


@needs_profiles
def test_chain_op_kernels_are_in_the_pmc_summary_and_the_kernel_stats():
    import importlib
    bench = importlib.import_module("bench")
    from descriptools_amd import chain
    assert bench.PMC_FILE == os.path.join(R2, "pmc_traffic.json")
    d = json.load(open(bench.PMC_FILE))
    stats = [r["Name"] for r in csv.DictReader(open(os.path.join(R2, "bench16384_kernel_stats.csv")))]
    for op, _, kernels in chain.OPS:
        assert bench.pmc_traffic(kernels, d["size"]) is not None, op
        for k in kernels:
            assert k in d["kernels"], k
            assert any(k.split("<")[0] in n for n in stats), k
    line = json.load(open(os.path.join(R2, "bench16384.json")))
    kernels = dict((o, k) for o, _, k in chain.OPS)[line["roofline"]["op"]]
    assert line["roofline"]["traffic"] == int(bench.pmc_traffic(kernels, d["size"]))


@needs_profiles
def test_design_measurement_table_is_generated_from_the_profiles():
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_table.py")], check=True,
                         capture_output=True, text=True).stdout.strip()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert gen in design, "DESIGN.md section 6 table is stale: regenerate with tools/design_table.py"
