#!/bin/bash
# Regenerates the per-round profile set of profiles/README.md on a GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r01
# Writes gpurun_out/<tag>_bench_default.json, <tag>_bench_under_rocprof.json, <tag>_bench_kernel_stats.csv,
# <tag>_pmc_hbm_summary.csv; copy the ones to be judged into profiles/.  Counters are collected in their own passes
# (--pmc never together with --stats / trace domains other than --kernel-trace).
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
echo "bench done"
rm -rf /tmp/prof1 /tmp/prof2 /tmp/prof3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o $TAG -- python3 bench.py > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rp1.err
cp "$(find /tmp/prof1 -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
head -2 "$(find /tmp/prof1 -name '*kernel_trace.csv' | head -1)"
python3 tools/trace_summary.py --min-grid 60000 "$(find /tmp/prof1 -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_bench_kernel_stats_batched.csv
echo "kernel trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof2 -o f -- python3 bench.py --cpu-seconds 1 > $OUT/${TAG}_pmc_fetch_bench.json 2> $OUT/${TAG}_rp2.err
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof3 -o w -- python3 bench.py --cpu-seconds 1 > $OUT/${TAG}_pmc_write_bench.json 2> $OUT/${TAG}_rp3.err
echo "write pass done"
python3 tools/pmc_summary.py --min-grid 60000 $(find /tmp/prof2 /tmp/prof3 -name '*counter_collection.csv') > $OUT/${TAG}_pmc_hbm_summary.csv
head -5 $OUT/${TAG}_pmc_hbm_summary.csv
