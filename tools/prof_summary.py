"""print the kernel_stats.csv of a rocprofv3 --kernel-trace --stats --output-format csv run (per step)"""
import csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
import os
f = max(glob.glob(d + '/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print("%-64s calls %5s avg %9.1f us  per step %8.3f ms" % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                float(r['TotalDurationNs']) / 1e6 / steps))
print("total per step %.3f ms" % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps))
