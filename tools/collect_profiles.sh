#!/bin/bash
# Regenerates the per-round profile set of profiles/README.md on a GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03 [steps]
# Writes gpurun_out/<tag>_bench_default.json, <tag>_bench_under_rocprof.json, <tag>_bench_kernel_stats.csv, <tag>_bench_kernel_stats_batched.csv,
# <tag>_timeline.txt; copy the ones to be judged into profiles/.  PMC counters: tools/collect_pmc.sh (own passes, never together with --stats).
set -e -o pipefail
TAG=${1:-r01}
STEPS=${2:-20}        # the driver's command: python bench.py --gpus 1 --steps 20 --warmup 5
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps $STEPS --warmup 5 > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
echo "bench done"
rm -rf /tmp/prof1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o $TAG -- python3 bench.py --steps $STEPS --warmup 5 --cpu-seconds 3 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rp1.err
cp "$(find /tmp/prof1 -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
python3 tools/trace_summary.py --min-grid 60000 "$(find /tmp/prof1 -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_bench_kernel_stats_batched.csv
python3 tools/timeline.py "$(find /tmp/prof1 -name '*kernel_trace.csv' | head -1)" --steps $STEPS > $OUT/${TAG}_timeline.txt
echo "kernel trace done"
# every launch of the dominant kernel by itself: start (ms after the first), duration, grid -- what the HIP-event figure of bench.py is checked against
python3 - "$(find /tmp/prof1 -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_ba_compute_launches.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]), r["Stream_Id"]) for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("k_ba_compute")]
rows.sort()
t0 = rows[0][0] if rows else 0
print("k_ba_compute launches of the traced run: start_ms duration_ms grid stream")
for a, b, g, s in rows:
    print("%10.3f %9.3f %8d %s" % ((a - t0) / 1e6, (b - a) / 1e6, g, s))
PY
cat $OUT/${TAG}_ba_compute_launches.txt
