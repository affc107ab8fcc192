"""per-kernel SQ counter summary from two rocprofv3 --pmc runs; with a third argument, the
kernel-trace stats csv of an un-counted run, also each kernel's share of the chip's VALU issue slots and LDS-array
cycles: instructions x 2.5 cycles (a wave64 fp32 / int instruction at 8 waves per SIMD on gfx950, measured by
tools/micro/valu_rate.hip; float64 FMA 4.5) over 1024 SIMDs x the kernel's duration at 2.4 GHz."""
import csv, glob, re, collections, os, sys
vals = collections.defaultdict(dict)
short = lambda n: re.match(r"([A-Za-z_0-9]+(?:<[^>]*>)?)", re.sub(r"^void ", "", n)).group(1)
dur = {}
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        dur[short(r["Name"])] = (float(r["AverageNs"]), int(r["Calls"]))
for d in sys.argv[1:3]:
    f = max(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r['Kernel_Name']); k = re.match(r"([A-Za-z_0-9]+(?:<[^>]*>)?)", k).group(1)
        vals[k][r['Counter_Name']] = float(r['Counter_Value'])
for k, v in vals.items():
    if 'SQ_WAVES' not in v not in v or v['SQ_WAVES'] < 1000 or 'SQ_WAVE_CYCLES' not in v: continue
    w, wc = v['SQ_WAVES'], v['SQ_WAVE_CYCLES']
    print(k)
    print("   per wave: VALU %.0f SALU %.0f LDS %.0f VMEMRD %.0f | LDS busy cycles %.0f, conflicts %.0f%%" % (
        v['SQ_INSTS_VALU'] / w, v['SQ_INSTS_SALU'] / w, v['SQ_INSTS_LDS'] / w, v.get('SQ_INSTS_VMEM_RD', 0) / w,
        v['SQ_LDS_IDX_ACTIVE'] / w, 100 * v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1)))
    print("   wave cycles: issuing %.0f%%, parked (waitcnt/barrier) %.0f%%, issue-stalled %.0f%% | VALU %.0f%% scalar %.0f%% LDS %.0f%%" % (
        100 * v['SQ_ACTIVE_INST_ANY'] / wc, 100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc,
        100 * v['SQ_ACTIVE_INST_VALU'] / wc, 100 * v['SQ_ACTIVE_INST_SCA'] / wc, 100 * v['SQ_ACTIVE_INST_LDS'] / wc))
    if k in dur:
        ns, _ = dur[k]
        cyc = ns * 2.4  # cycles of one launch at 2.4 GHz
        # the counter rows are per dispatch; vals keeps the last launch of each kernel
        print("   chip level (%.3f ms per launch): VALU issue %.0f%% of 1024 SIMDs x 2.5 cycles, LDS array %.0f%% of 256 CUs"
              % (ns / 1e6, 100 * v['SQ_INSTS_VALU'] * 2.5 / (1024 * cyc),
                 100 * v['SQ_LDS_IDX_ACTIVE'] / (256 * cyc)))
