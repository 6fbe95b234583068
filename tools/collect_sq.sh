#!/bin/bash
# SQ counters of k_ba_compute (where the wave cycles go) on one synchronous round of S identical problems (tools/pmc_ba_micro.py):
#   bash tools/collect_sq.sh r03 512
TAG=${1:-r03}; S=${2:-512}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $(dirname $0)/..
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM"
i=0
for C in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/sq_$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/sq_$i -o p -- python3 tools/pmc_ba_micro.py $S > $OUT/${TAG}_sq_${i}_run.json 2> $OUT/${TAG}_sq_${i}.err
  echo "pass $i rc $?"
done
python3 tools/pmc_summary.py --min-grid 512 $(find /tmp/sq_1 /tmp/sq_2 -name '*counter_collection.csv') | grep -E "^kernel|k_ba_compute" > $OUT/${TAG}_sq_ba_summary.csv
cat $OUT/${TAG}_sq_ba_summary.csv
