"""per-rank kernel durations of tools/ds_ranks_real.py under rocprofv3 --kernel-trace (the last repetition of each rank)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0].replace('void ', ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows]
idx = [i for i, (n, d) in enumerate(seq) if n.startswith('k_ds_finish')]
for i in idx[4::5]:
    print(" | ".join("%s %.3f" % (n.replace('k_ds_', '').replace('k_downslope_', ''), d) for n, d in seq[i - 6:i + 1]))
