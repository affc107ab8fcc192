#!/bin/bash
# A/B of two builds on the workloads with flats and nodata: the tiled Example (real terrain), the conditioned rough
# 16384^2 chain, and the headline step (serial per-op times)
#   tools/flat_moves_ab.sh <outdir> <lib> [<lib> ...]
OUT=$1; shift
mkdir -p $OUT
for lib in "$@"; do
  v=$(basename $lib .so)
  cp $lib descriptools_amd/libdescriptools_hip.so || exit 1
  python3 tools/real_terrain_run.py 8 4096 > $OUT/$v.rt.txt 2>&1 || exit 1
  python3 tools/condition_bench.py 16384 > $OUT/$v.cond.txt 2>&1 || exit 1
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-overlap --no-placement --no-verify > $OUT/$v.bench.json 2>/dev/null || exit 1
done
