cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3n/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-e2e > gpurun_out/r3n/trace.log 2>&1
cp $(find gpurun_out/r3n/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r3n/kernel_stats.csv
rm -rf gpurun_out/r3n/trace
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r3n/kernel_stats.csv")))
for r in rows[:16]:
    print("%-60s calls %4s  avg %9.1f us  total %%%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
