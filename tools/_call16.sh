set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
timeout -k 10 600 python -m pytest tests/test_gpu_hydro.py tests/test_gpu_tiling.py tests/test_gpu_run_rank.py -x -q -m gpu 2>&1 | grep -v "^  *+" | tail -15 | cut -c1-300
timeout -k 10 300 python3 tools/condition_bench.py > gpurun_out/r3p/conditioning.txt 2>&1; cat gpurun_out/r3p/conditioning.txt | cut -c1-400
