#!/bin/bash
# (GPU box) kernel trace of the encrypted-inference benchmark: eager forwards with a live dealer, the graphed online phase, the
# one-launch dealer refill and the pipelined stream.   bash tools/profile_secure.sh r06 -> profiles/r06_secure_kernel_stats.csv
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r06}
O=gpurun_out/prof_secure_$R
rm -rf $O; mkdir -p $O profiles
rocprofv3 --kernel-trace --stats -d $O/trace --output-format rocpd -- python3 tools/bench_secure.py --images 1 > $O/trace.log 2>&1
python3 tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) profiles/${R}_secure_kernel_stats.csv > /dev/null
cp profiles/${R}_secure_kernel_stats.csv gpurun_out/
head -14 profiles/${R}_secure_kernel_stats.csv | cut -c1-160
echo "kernels of torch's (at::native) in the trace:"; grep -c "at::native" profiles/${R}_secure_kernel_stats.csv
grep "at::native" profiles/${R}_secure_kernel_stats.csv | cut -c1-160
