#!/bin/bash
# The other bench lines of DESIGN.md section 6 / profiles/README.md, one file each (through gpurun from the repo root):
#   bash tools/collect_round_lines.sh r03
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
run() { name=$1; shift; python3 bench.py "$@" > $OUT/${TAG}_bench_$name.json 2> $OUT/${TAG}_bench_$name.err; python3 -c "
import json; d=json.load(open('$OUT/${TAG}_bench_$name.json')); print('$name', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d.get('parity_pose_maxdiff_stream0'))"; }
if [ -z "$SKIP_FIRST" ]; then
run k50 --steps 50 --warmup 5 --no-all-cores --cpu-seconds 3
run config_ba --config ba --steps 8 --warmup 2 --cpu-seconds 2
fi
run config3 --streams 2048 --width 1280 --height 720 --ba-window 10 --feeder-rects 1200 --steps 30 --warmup 5 --no-all-cores --cpu-seconds 4
run config3_np2000 --streams 2048 --width 1280 --height 720 --ba-window 10 --feeder-rects 1200 --steps 30 --warmup 5 --no-all-cores --cpu-seconds 4 --max-patches 2000 --no-flat-out
run host_frames --host-frames --streams 512 --steps 20 --warmup 5 --no-all-cores --cpu-seconds 1 --no-flat-out
run ordered --ba-sum-order 1 --steps 20 --warmup 5 --no-all-cores --cpu-seconds 1 --no-flat-out
run 1000frames_ordered --steps 1000 --streams 64 --max-keyframes 64 --ba-sum-order 1 --no-all-cores --cpu-seconds 3 --no-flat-out
VSLAM_PROFILE_SERIAL=1 python3 bench.py --steps 20 --warmup 5 --no-all-cores --cpu-seconds 1 --no-flat-out > $OUT/${TAG}_bench_serial_stages.json 2> $OUT/${TAG}_bench_serial_stages.err
python3 -c "
import json; d=json.load(open('$OUT/${TAG}_bench_serial_stages.json')); print('serial', d['value'], {k: round(v['ms_per_launch'],3) for k,v in d['stages'].items()})"
