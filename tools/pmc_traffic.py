"""Build profiles/<round>/<tag>_pmc_traffic.json from two rocprofv3 counter runs
(`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, each of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify
--no-e2e --no-overlap --no-placement`: the serial schedule).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [size]

Per kernel: KB summed over its launches in the timed step, and launches per step; bench.py turns them
into HBM bytes with the gfx950 correction of MI355X_MICROARCH.md: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
"""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def read(path, counter, steps):
    """-> {kernel: KB summed over the launches of the LAST step}, {kernel: launches per step}"""
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        vals.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    per_step = {k: len(v) // steps for k, v in vals.items()}
    total = {k: (sum(v[-per_step[k]:]) if per_step[k] else 0.0) for k, v in vals.items()}
    return total, per_step


def main():
    steps = 4  # --no-overlap --steps 1 --warmup 1: (warm-up + timed step) of the headline loop and of the per-op loop
    fetch, calls = read(sys.argv[1], "FETCH_SIZE", steps)
    write, _ = read(sys.argv[2], "WRITE_SIZE", steps)
    size = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
    out = {"size": size,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `python3 bench.py "
                   "--steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --no-overlap --no-placement`; values are KB "
                   "summed over a kernel's launches in the last step; hbm_bytes = "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request on wide "
                   "coalesced reads, MI355X_MICROARCH.md HBM section; for the 64-B-row tile loads the factor 2 is "
                   "an upper bound)",
           "kernels": {}}
    for k in fetch:
        out["kernels"][k] = {"fetch_kb_step": fetch[k], "write_kb_step": write.get(k, 0.0),
                             "launches_per_step": calls[k]}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, e in out["kernels"].items():
        print("%-44s x%-3d fetch %10.1f MB  write %10.1f MB" % (k, e["launches_per_step"],
                                                                e["fetch_kb_step"] / 1024, e["write_kb_step"] / 1024))


if __name__ == "__main__":
    main()
