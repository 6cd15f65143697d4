#!/bin/bash
# step-time A/B of ENGINE schedule switches on one box: tools/ab_engine.sh ROUNDS "name=v ..." "" ...  (as tools/ab_step.sh, with
# bench.py --engine-opt instead of --opt)
rounds=$1; shift
for i in $(seq $rounds); do
  for cfg in "$@"; do
    flags=""; for kv in $cfg; do flags="$flags --engine-opt $kv"; done
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 $flags 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'])"
  done
done
