set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout -k 10 300 python3 tools/condition_bench.py > gpurun_out/r3q/conditioning.txt 2>&1; cat gpurun_out/r3q/conditioning.txt | cut -c1-400
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3q/pytest_all.log 2>&1; rc=$?; tail -5 gpurun_out/r3q/pytest_all.log | cut -c1-300; exit $rc
