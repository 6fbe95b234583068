#!/bin/bash
# HBM traffic of k_ba_compute by PMC (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes, kernel trace only) on ONE
# round of S identical problems (tools/pmc_ba_micro.py): bench.py under --pmc does not finish within the box's limits (every one
# of its ~20,000 set-up dispatches is serialised through the counters).
#   bash tools/collect_pmc.sh r02 256
TAG=${1:-r02}; S=${2:-256}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/pmc_ba_micro.py $S > $OUT/${TAG}_pmc_ba_micro.json 2> $OUT/${TAG}_pmc_ba_micro.err; cat $OUT/${TAG}_pmc_ba_micro.json
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  echo "pass $C"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o p -- python3 tools/pmc_ba_micro.py $S > $OUT/${TAG}_pmc_${C}_run.json 2> $OUT/${TAG}_pmc_${C}.err
  echo "pass $C rc $?"
done
python3 tools/pmc_summary.py --min-grid 512 $(find /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv') > $OUT/${TAG}_pmc_hbm_summary.csv
head -12 $OUT/${TAG}_pmc_hbm_summary.csv
