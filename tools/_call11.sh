cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
for pat in plain plain pad bench bench; do timeout -k 10 200 python3 tools/placement_probe.py classes 40 $pat 2>&1 | tail -n +2 | cut -c1-700 | tee -a gpurun_out/r3k/classes.log; done
