#!/bin/bash
# per-launch trace of one conditioning run: tools/cond_trace_run.sh <outdir> <size>
OUT=$1; N=$2
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 tools/condition_bench.py $N > $OUT/run.log 2>&1 ) && python3 tools/cond_trace.py $OUT/trace > $OUT/cond_trace.txt
