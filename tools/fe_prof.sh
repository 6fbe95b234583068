#!/bin/bash
# kernel times of the front end (tools/fe_bench.py under rocprofv3 --kernel-trace --stats) for the product library and any VSLAM_LIB variants
export TMPDIR=/tmp
for lib in "" "$@"; do
  rm -rf /tmp/pf
  VSLAM_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o fe -- python3 tools/fe_bench.py 1024 20 > /tmp/fe_prof.log 2>&1
  echo "== ${lib:-product} ${VSLAM_COMPACT_BAND}"; grep "ms/call" /tmp/fe_prof.log
  grep -v fillBuffer "$(find /tmp/pf -name '*kernel_stats.csv')" | head -4 | cut -d, -f1-4
done
