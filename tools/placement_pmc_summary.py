"""Per-generation summary of tools/placement_pmc.sh's counter runs: for every rocprofv3 counter CSV, the dispatches of
the fused stencil grouped by raster generation (13 launches each in tools/placement_probe.py), mean duration and mean
counter values per generation.  usage: placement_pmc_summary.py <dir with *_counter_collection.csv> [launches/gen]"""
import csv, glob, os, sys
d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 13
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    rows = {}
    for r in csv.DictReader(open(f)):
        if "k_slope_twi<" not in r["Kernel_Name"]:
            continue
        e = rows.setdefault(int(r["Dispatch_Id"]), {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(rows)
    names = sorted(k for k in rows[ids[0]] if k != "dur")
    print(os.path.basename(f))
    print("  gen   ms    " + "  ".join("%22s" % n[-22:] for n in names))
    for g in range(len(ids) // per):
        grp = [rows[i] for i in ids[g * per:(g + 1) * per]][3:]  # drop the warm-up launches
        ms = sum(e["dur"] for e in grp) / len(grp)
        print("  %2d  %.3f  " % (g, ms) + "  ".join("%22.0f" % (sum(e[n] for e in grp) / len(grp)) for n in names))
