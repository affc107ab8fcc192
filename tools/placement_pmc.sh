#!/bin/bash
# Round 3: hardware counters for a fast and a slow placement of the fused slope + TI + MTI stencil's five rasters.
# One process allocates several generations of the rasters (tools/placement_probe.py), so one rocprofv3 run holds
# fast and slow generations of the same build; counters are per dispatch.
set -u
: ${GRAFT_REPO_ROOT:?}
TAG=${1:-r3_placement}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ "${2:-all}" = "tlb" ] || rocprofv3 -L > $OUT/counters_avail.txt 2>&1
python3 tools/placement_probe.py 6 > $OUT/probe_plain.log 2>&1 || exit 1
run_pmc() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 ${EXTRA:-} --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/placement_probe.py 6 > $OUT/$name.log 2>&1
  local f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_counter_collection.csv
  local t=$(find $OUT/$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && cp $t $OUT/${name}_kernel_trace.csv
  rm -rf $OUT/$name
  echo "$name done"
}
if [ "${2:-all}" = "tlb" ]; then
  # second collection: address translation, and the spread over the 128 L2 channels (tools/pmc/tcc_minmax.yaml)
  run_pmc tlb_a TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
  run_pmc tlb_b GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
  EXTRA="-E tools/pmc/tcc_minmax.yaml"
  run_pmc mm_a TCC_EA0_WRREQ_max TCC_EA0_WRREQ_min TCC_EA0_WRREQ_DRAM_CREDIT_STALL_max TCC_EA0_WRREQ_DRAM_CREDIT_STALL_min
  run_pmc mm_b TCC_EA0_RDREQ_max TCC_EA0_RDREQ_min TCC_TAG_STALL_max TCC_TAG_STALL_min
else
run_pmc tcc_a TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_64B_sum
run_pmc tcc_b TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
run_pmc tcc_c TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
run_pmc tcc_d TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum
run_pmc tcc_e TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_WR_UNCACHED_32B_sum
fi
ls -la $OUT
