#!/bin/bash
# Collects this round's measurement set on the GPU box into gpurun_out/<tag>/ (copied to profiles/<round>/ afterwards):
#   bench16384.json                           the un-profiled default bench line (headline = overlapped schedule, per-op
#                                             from its serial loop, verification, end_to_end, cpu_baseline)
#   bench16384_serial.json / _tiled.json      --no-overlap; --tiled (N = 1 through the multi-rank path)
#   bench16384_kernel_stats.csv               rocprofv3 --kernel-trace --stats of the SERIAL schedule (--no-overlap: a kernel
#                                             that runs beside others has no duration of its own)
#   pmc_{fetch,write}_counter_collection.csv  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, serial schedule
#   pmc_traffic.json                          per-kernel summary of the two (tools/pmc_traffic.py)
#   sq_counters.txt                           per-kernel SQ summary of two more --pmc runs (tools/sq_summary.py)
#   placement_*                               tools/placement_pmc.sh / placement_probe.py: the write-conflict classes
set -euo pipefail
: ${GRAFT_REPO_ROOT:?}
TAG=${1:-r4_final}
ROUND=${2:-r4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
. tools/pmc_pass.sh   # every counter pass goes through pmc_pass: one TCC byte counter per pass, program after --
OUT=gpurun_out/$TAG
mkdir -p $OUT profiles/$ROUND
# placement tuning off in the counter runs: its ~250 probe launches would only pad the traces (which block serves
# which raster does not change what a kernel moves)
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --no-overlap --no-placement"
pmc_pass $OUT/pmc_fetch FETCH_SIZE -- $B
pmc_pass $OUT/pmc_write WRITE_SIZE -- $B
cp $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $OUT/pmc_fetch_counter_collection.csv
cp $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) $OUT/pmc_write_counter_collection.csv
python3 tools/pmc_traffic.py $OUT/pmc_fetch_counter_collection.csv $OUT/pmc_write_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log
cp $OUT/pmc_traffic.json profiles/$ROUND/pmc_traffic.json
echo "pmc done"
pmc_pass $OUT/sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- $B
pmc_pass $OUT/sq_b SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -- $B
echo "sq done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-e2e --no-overlap --no-placement > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench16384_kernel_stats.csv
python3 tools/sq_summary.py $OUT/sq_a $OUT/sq_b $OUT/bench16384_kernel_stats.csv > $OUT/sq_counters.txt
echo "trace done"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench16384.json 2> $OUT/bench16384.err
echo "bench done"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-overlap > $OUT/bench16384_serial.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --tiled > $OUT/bench16384_tiled.json 2>/dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/sq_a $OUT/sq_b $OUT/trace
# round 4: the placement study is tools/micro/placement_map (physical blocks, delta scan) and tools/arena_probe.py,
# run on their own (profiles/r4/placement_map_*.txt, placement_arena_probe.txt); the conditioning and the real-terrain
# walkers likewise (tools/condition_bench.py, tools/ds_ranks_real.py)
python3 tools/condition_bench.py > $OUT/conditioning.txt 2>&1 || true
python3 tools/condition_bench.py 16384 >> $OUT/conditioning.txt 2>&1 || true
python3 tools/ds_ranks_real.py check finish > $OUT/downslope_ranks_real_terrain.txt 2>&1 || true
rocm-smi --showclocks --showpower > $OUT/rocm_smi.txt 2>&1 || true
ls -la $OUT
