#!/bin/bash
# step-time A/B of option settings on one box: tools/ab_step.sh ROUNDS "opt1=v1 opt2=v2" "opt1=w1" ...   (each argument = one
# configuration: space-separated NAME=VALUE pairs, "" = defaults); alternating runs of 200 replayed steps; ms per step
rounds=$1; shift
for i in $(seq $rounds); do
  for cfg in "$@"; do
    flags=""; for kv in $cfg; do flags="$flags --opt $kv"; done
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 $flags 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'])"
  done
done
