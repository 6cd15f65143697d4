#!/bin/bash
# kernel trace + per-launch timeline of one DP-SGD step (bench.py --dp, launched eagerly)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/trace_dp
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace --output-format rocpd -- python3 bench.py --dp --steps 6 --warmup 2 --no-graph --no-secure --no-cpu-baseline --sustain-s 0 > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB gpurun_out/r03_dp_kernel_stats.csv > /dev/null
python3 tools/rocpd_timeline.py $DB gpurun_out/r03_dp_step_timeline.txt > /dev/null
tail -1 gpurun_out/r03_dp_step_timeline.txt
