#!/bin/bash
# same-box A/B of one environment switch over the training bench: tools/ab_bench.sh VAR A B [rounds]
# (alternating runs; prints images/s, ms per step per run)
var=$1; a=$2; b=$3; rounds=${4:-2}; extra=${5:-}
for i in $(seq $rounds); do
  for v in $a $b; do
    env $var=$v python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 $extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_step'])"
  done
done
