#!/bin/bash
# HBM traffic of the bench's OWN launches by PMC: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes
# (kernel trace only, as gpurun requires) over `python3 bench.py --streams S ...`.  The set-up of bench.py is batched (front end over
# 256 keyframe images per call, eight keyframes per upload, one contiguous frame upload per stream), so that a counter pass -- which
# serialises every dispatch -- finishes inside the box's limit; under a profiler bench.py skips the all-cores CPU leg by itself.
#   bash tools/collect_pmc_bench.sh r03 256
TAG=${1:-r03}; S=${2:-256}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $(dirname $0)/..
ARGS="--streams $S --steps 42 --warmup 5 --cpu-seconds 1 --parity-check 0 --no-flat-out --ba-batch 10"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$C
  echo "pass $C"
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcb_$C -o p -- python3 bench.py $ARGS > $OUT/${TAG}_pmcb_${C}.json 2> $OUT/${TAG}_pmcb_${C}.err
  echo "pass $C rc $?"
done
# the batched launches only: grids of at least S workgroups x 64 lanes
python3 tools/pmc_summary.py --min-grid $((S * 64)) $(find /tmp/pmcb_FETCH_SIZE /tmp/pmcb_WRITE_SIZE -name '*counter_collection.csv') > $OUT/${TAG}_pmc_bench_summary.csv
head -40 $OUT/${TAG}_pmc_bench_summary.csv
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_bench_summary.csv $OUT/${TAG}_pmcb_FETCH_SIZE.json $OUT/${TAG}_pmcb_WRITE_SIZE.json $OUT/${TAG}_traffic
