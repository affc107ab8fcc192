#!/usr/bin/env python3
"""Prints DESIGN.md section 6's measurement table from the committed profile set (profiles/r4/): the un-profiled
bench line (per-op HIP-event times), the rocprofv3 kernel-trace summary (per-kernel averages) and the PMC traffic
summary.  tests/test_profiles.py checks that DESIGN.md contains exactly this output."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from descriptools_amd import chain  # noqa: E402

R = os.path.join(ROOT, "profiles", "r4")
LABEL = {"d8": "D8", "downslope": "downslope",
         "flowacc_flowhand_local": "flow accumulation + river mask + HAND phase 1 (tile solves, node jumps)",
         "flowhand_gfi_finish": "HAND last pass: fdist, idx, hand + GFI + ln(hl/H)",
         "slope_twi": "fused slope + TI + MTI"}


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    line = json.load(open(os.path.join(R, "bench16384.json")))
    pmc = json.load(open(os.path.join(R, "pmc_traffic.json")))
    stats = {}
    for r in csv.DictReader(open(os.path.join(R, "bench16384_kernel_stats.csv"))):
        stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
    steps_profiled = 8  # --no-overlap: (1 warm-up + 3 timed steps) x (headline loop + per-op loop), all serial
    n = pmc["size"] ** 2
    out = ["| op (kernels) | algorithmic B/cell | ms (HIP events, serial loop of the un-profiled run) | achieved GB/s | % of 8 TB/s | "
           "rocprofv3 kernel ms per step | PMC HBM B/cell (2·FETCH+WRITE) |",
           "|---|---|---|---|---|---|---|"]
    tot_ms = tot_b = 0.0
    for op, bpc, kernels in chain.OPS:
        po = line["per_op"][op]
        kms = []
        traffic = 0.0
        for k in kernels:
            calls, avg = stats[k]
            kms.append("%s %.3f" % (k.split("<")[0], avg * calls / steps_profiled))
            e = pmc["kernels"][k]
            traffic += (2.0 * e["fetch_kb_step"] + e["write_kb_step"]) * 1024.0
        tot_ms += po["ms"]
        tot_b += traffic
        out.append("| %s (`%s`) | %d | %.3f | %.0f | %.1f %% | %s | %.1f |" % (
            LABEL[op], "`, `".join(k.split("<")[0] for k in kernels), bpc, po["ms"], po["achieved_GBs"], 100 * po["frac"],
            "; ".join(kms), traffic / n))
    out.append("| **chain** | %d (unfused definition) | **%.3f ms/step overlapped -> %.1f Gcells/s** (%.3f serial) | %.0f | %.1f %% | sum of the ops "
               "%.3f | %.1f (%.1f GB per step) |" % (chain.ALGO_BYTES_PER_CELL, line["ms_per_step"], line["value"] / 1e3,
                                                    line["ms_per_step_serial"],
                                                    n * chain.ALGO_BYTES_PER_CELL / (line["ms_per_step"] * 1e-3) / 1e9,
                                                    100 * line["chain_frac_of_hbm_peak"], tot_ms, tot_b / n, tot_b / 1e9))
    print("\n".join(out))


if __name__ == "__main__":
    main()
