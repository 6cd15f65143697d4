#!/bin/bash
# (GPU box) is the training step POWER-bound?  Samples rocm-smi (socket power, clocks, cap) while the benchmark replays
# its step for ~15 s, and once idle.  tools/power_probe.sh
cd "$(dirname "$0")/.."
echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -iE "power|sclk|mclk|fclk" | head -12
python3 bench.py --steps 3000 --warmup 20 --no-secure --no-cpu-baseline --sustain-s 0 > gpurun_out/power_bench.log 2>&1 &
pid=$!
sleep 9
for i in 1 2 3 4 5 6; do
  echo "== under load, sample $i"; rocm-smi --showpower --showclocks 2>&1 | grep -iE "power|sclk|mclk" | head -6
  sleep 1
done
wait $pid
tail -1 gpurun_out/power_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
