"""summary of tools/ds_finish_ab.sh: per build, mean duration and FETCH_SIZE of the long-walk kernels"""
import csv, glob, sys, os
root = sys.argv[1]
for d in sorted(glob.glob(root + "/*_fetch")):
    v = os.path.basename(d)[:-6]
    dur = {}
    for r in csv.DictReader(open(glob.glob(root + "/" + v + "/*kernel_trace.csv")[0])):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        dur.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    fetch = {}
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")
            fetch.setdefault(n, []).append(float(r["Counter_Value"]))
    print(v)
    for n in sorted(dur):
        if n.startswith("k_ds_") or n.startswith("k_downslope"):
            big = [x for x in dur[n] if x > 0.05]
            fb = [x for x in fetch.get(n, []) if x > 0]
            if big:
                # FETCH_SIZE: KiB on gfx950 after the guide's correction? printed raw here (units of the counter) and as GB assuming KiB
                print("   %-28s %3d launches  mean %.3f ms   FETCH_SIZE mean %.0f (x1024 = %.2f GB)"
                      % (n, len(big), sum(big) / len(big), sum(fb) / max(len(fb), 1), sum(fb) / max(len(fb), 1) * 1024 / 1e9))
