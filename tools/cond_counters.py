"""per-DISPATCH SQ counters of the conditioning's relaxation kernels (tools/sq_summary.py keeps a kernel's last launch,
which for these kernels is a quiet round): python tools/cond_counters.py <pmc dir A> <pmc dir B>"""
import collections
import csv
import glob
import os
import sys

rows = collections.defaultdict(dict)
for d in sys.argv[1:3]:
    f = max(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if k not in ('k_fill_relax', 'k_flat_relax', 'k_flat_assign', 'k_flat_init', 'k_fill_init'):
            continue
        rows[(k, int(r['Dispatch_Id']))][r['Counter_Name']] = float(r['Counter_Value'])
seen = collections.Counter()
for (k, did), v in sorted(rows.items(), key=lambda kv: kv[0][1]):
    if 'SQ_WAVES' not in v or v['SQ_WAVES'] < 1:
        continue
    seen[k] += 1
    if seen[k] > 12:
        continue
    w, wc = v['SQ_WAVES'], max(v.get('SQ_WAVE_CYCLES', 1), 1)
    print("%-14s #%-5d waves %8.0f | per wave: VALU %6.0f SALU %5.0f LDS %5.0f VMEM_RD %4.0f | LDS busy %7.0f conflicts %3.0f%% | "
          "wave cycles %9.0f: issuing %2.0f%% parked %2.0f%% VALU %2.0f%% LDS %2.0f%%" % (
              k, did, w, v.get('SQ_INSTS_VALU', 0) / w, v.get('SQ_INSTS_SALU', 0) / w, v.get('SQ_INSTS_LDS', 0) / w,
              v.get('SQ_INSTS_VMEM_RD', 0) / w, v.get('SQ_LDS_IDX_ACTIVE', 0) / w,
              100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1), wc / w,
              100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * v.get('SQ_WAIT_ANY', 0) / wc,
              100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * v.get('SQ_ACTIVE_INST_LDS', 0) / wc))
