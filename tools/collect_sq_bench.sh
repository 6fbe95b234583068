#!/bin/bash
# SQ counters of the TRACKER's kernels over bench.py's own launches (where the wave cycles of k_pvs / k_searchN / k_pose / the front end go):
#   bash tools/collect_sq_bench.sh r03 256
# Two counter passes (kernel trace only, as gpurun requires); a counter pass serialises the dispatches, so the kernels are seen alone.
TAG=${1:-r03}; S=${2:-256}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $(dirname $0)/..
ARGS="--streams $S --steps 12 --warmup 3 --cpu-seconds 1 --parity-check 0 --no-flat-out --ba-batch 10"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"
i=0
for C in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/sqb_$i
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/sqb_$i -o p -- python3 bench.py $ARGS > $OUT/${TAG}_sqb_${i}_run.json 2> $OUT/${TAG}_sqb_${i}.err
  echo "pass $i rc $?"
done
python3 tools/pmc_summary.py --min-grid $((S * 64)) $(find /tmp/sqb_1 /tmp/sqb_2 -name '*counter_collection.csv') > $OUT/${TAG}_sq_bench_summary.csv
python3 tools/sq_table.py $OUT/${TAG}_sq_bench_summary.csv | tee $OUT/${TAG}_sq_bench_table.txt
