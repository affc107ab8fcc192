#!/bin/bash
# A/B of two builds of the library on the real-terrain leg (long downslope walks): kernel durations and HBM fetch
# bytes of the long-walk kernels.   tools/ds_finish_ab.sh <outdir> <lib> [<lib> ...]
. tools/pmc_pass.sh
OUT=$1; shift
mkdir -p $OUT
for lib in "$@"; do
  v=$(basename $lib .so)
  cp $lib descriptools_amd/libdescriptools_hip.so || exit 1
  ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o t -- python3 tools/real_terrain_run.py 8 256 > $OUT/$v.log 2>&1 ) || exit 1
  pmc_pass $OUT/${v}_fetch FETCH_SIZE -- python3 tools/real_terrain_run.py 8 256 || exit 1
done
