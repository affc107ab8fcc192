#!/bin/bash
# tools/gpu_call.sh <tag> <timeout-seconds> '<command>'
# One gpurun call whose shipped tree can be reconstructed afterwards: HEAD and the uncommitted diff are written to
# gpurun_out/<tag>/tree.{head,diff} before the snapshot is taken.  (Round 3 lost the cause of an abort because the
# work-in-progress tree that crashed was never recorded: DESIGN.md 2.)
set -u
TAG=$1; LIMIT=$2; CMD=$3
mkdir -p gpurun_out/$TAG
git rev-parse HEAD > gpurun_out/$TAG/tree.head
git diff HEAD > gpurun_out/$TAG/tree.diff
git status --short | grep '^??' > gpurun_out/$TAG/tree.untracked || true
exec /usr/local/graft/bin/gpurun --timeout $LIMIT -- "mkdir -p gpurun_out/$TAG && $CMD"
