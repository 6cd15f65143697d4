#!/bin/bash
# kernel trace of a short un-graphed bench run, summarised per kernel: tools/trace_tail.sh TAG [ENV=VAL ...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=gpurun_out/trace_$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace --output-format rocpd -- python3 bench.py --steps 6 --warmup 2 --no-graph --no-secure --no-cpu-baseline --sustain-s 0 > $O/trace.log 2>&1
python3 tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) gpurun_out/stats_$tag.csv > /dev/null
head -45 gpurun_out/stats_$tag.csv
