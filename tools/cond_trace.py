"""per-launch kernel durations of ONE conditioning run from a rocprofv3 --kernel-trace csv directory"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0].replace('void ', ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6) for r in rows]
starts = [i for i, (n, d) in enumerate(seq) if n.startswith('k_fill_relax<true>') or n == 'k_fill_init']
ends = [i for i, (n, d) in enumerate(seq) if n.startswith('k_flat_assign')]
i0, i1 = starts[0], [e for e in ends if e > starts[0]][0]
tot = {}
line = []
for n, d in seq[i0:i1 + 1]:
    if n.startswith('__amd'):
        continue
    tot[n] = tot.get(n, 0) + d
    line.append("%s %.2f" % (n.replace('k_', ''), d))
print(" | ".join(line))
print("totals (ms):", {k: round(v, 2) for k, v in tot.items()}, "sum %.2f" % sum(tot.values()))
