#!/bin/bash
# Collects this round's measurement set on the GPU box into gpurun_out/<tag>/ (copied to profiles/<round>/ afterwards):
#   pmc_{fetch,write}_counter_collection.csv  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs
#   pmc_traffic.json                          per-kernel summary of the two (tools/pmc_traffic.py)
#   sq_a / sq_b counter csv + sq_counters.txt per-kernel SQ summary (tools/sq_summary.py)
#   bench16384_kernel_stats.csv               rocprofv3 --kernel-trace --stats of the same bench command
#   bench16384.json / _overlap.json / _tiled.json   un-profiled bench lines (the first with cpu_baseline + verification)
set -o pipefail
TAG=${1:-r2_final}
ROUND=${2:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT profiles/$ROUND
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 1
cp $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $OUT/pmc_fetch_counter_collection.csv
cp $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/pmc_write_counter_collection.csv
python3 tools/pmc_traffic.py $OUT/pmc_fetch_counter_collection.csv $OUT/pmc_write_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log || exit 1
cp $OUT/pmc_traffic.json profiles/$ROUND/pmc_traffic.json
echo "pmc done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq_a -- $B > $OUT/sq_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq_b -- $B > $OUT/sq_b.log 2>&1 || exit 1
echo "sq done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $OUT/trace.log 2>&1 || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench16384_kernel_stats.csv
python3 tools/sq_summary.py $OUT/sq_a $OUT/sq_b $OUT/bench16384_kernel_stats.csv > $OUT/sq_counters.txt || exit 1
echo "trace done"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench16384.json 2> $OUT/bench16384.err || exit 1
echo "bench done"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --overlap > $OUT/bench16384_overlap.json 2>/dev/null || exit 1
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --tiled > $OUT/bench16384_tiled.json 2>/dev/null || exit 1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/sq_a $OUT/sq_b $OUT/trace
# calibration of the instruction-issue model used in DESIGN.md 6 (cycles per wave64 VALU instruction, latency of a
# dependent atomic) and the device's clocks
for m in valu_rate atomic_chain; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/micro/$m.hip -o /tmp/$m && timeout -k 5 60 /tmp/$m > $OUT/micro_$m.txt 2>&1
done
rocm-smi --showclocks --showpower > $OUT/rocm_smi.txt 2>&1 || true
ls -la $OUT
