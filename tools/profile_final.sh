#!/bin/bash
# (GPU box) the round's evidence in one call: PMC / trace passes over the benchmark (tools/profile_round.sh), the timeline of the
# REPLAYED step, the encrypted-inference trace, the GPU test log and the benchmark line — everything lands in gpurun_out/ (merged
# back) under the names profiles/ keeps.   bash tools/profile_final.sh r06
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r06}
bash tools/profile_round.sh $R > gpurun_out/${R}_profile_round.log 2>&1
O=gpurun_out/prof_${R}_graph
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/trace --output-format rocpd -- python3 bench.py --steps 40 --warmup 5 --no-secure --no-cpu-baseline --sustain-s 0 > $O/trace.log 2>&1
python3 tools/rocpd_timeline.py $(find $O/trace -name "*.db" | head -1) gpurun_out/${R}_step_timeline_graph.txt > /dev/null
rm -rf $O/trace
bash tools/profile_secure.sh $R > gpurun_out/${R}_profile_secure.log 2>&1
bash tools/pmc_secure.sh > gpurun_out/${R}_pmc_secure.log 2>&1
rm -rf gpurun_out/prof_$R gpurun_out/prof_secure_$R gpurun_out/pmc_secure
python3 -m pytest tests -m gpu -q > gpurun_out/${R}_gpu_tests.log 2>&1
tail -3 gpurun_out/${R}_gpu_tests.log | cut -c1-200
python3 bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err
tail -c 400 gpurun_out/${R}_bench_final.json
