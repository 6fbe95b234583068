#!/bin/bash
# The configurations of DESIGN.md section 6's table, one bench line each (bash tools/bench_table.sh, through gpurun).
set -e -o pipefail
OUT=gpurun_out
mkdir -p $OUT
run() { name=$1; shift; python3 bench.py --cpu-seconds 1 "$@" > $OUT/tab_$name.json 2> $OUT/tab_$name.err; python3 -c "
import json; d=json.load(open('$OUT/tab_$name.json')); print('$name', d['value'], d['ms_per_step'], d['config']['streams_tracking_good'], d['parity_pose_maxdiff_stream0'])"; }
run s512 --streams 512
run s256 --streams 256
run s256_sync --streams 256 --ba-delay 0
run s512_p11 --streams 512 --patch 11
run s512_sbi --streams 512 --use-sbi 1
run s1024_grow3 --grow-map 3
VSLAM_PROFILE_SERIAL=1 python3 bench.py --cpu-seconds 1 --streams 256 > $OUT/tab_s256_serial.json 2> $OUT/tab_s256_serial.err
python3 -c "
import json; d=json.load(open('$OUT/tab_s256_serial.json')); print('serial256', d['value'], {k: round(v['ms_per_launch'],3) for k,v in d['stages'].items()})"
