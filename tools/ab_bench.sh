#!/bin/bash
# same-box A/B of one library option over the training bench: tools/ab_bench.sh OPTION A B [rounds] [extra bench args]
# (alternating runs; prints images/s, ms per step per run).  OPTION is a primia_set_option name (csrc/options.h), or
# engine:NAME for a ResNet18Engine schedule switch.  AB_KERNEL=substring also prints that kernel family's average launch.
var=$1; a=$2; b=$3; rounds=${4:-2}; extra=${5:-}
flag=--opt; name=$var
case $var in engine:*) flag=--engine-opt; name=${var#engine:};; esac
for i in $(seq $rounds); do
  for v in $a $b; do
    python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 $flag $name=$v $extra 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; f='${AB_KERNEL:-}'; print('$var=$v', d['value'], d['ms_per_step'], *[(n[:28], k[n]['avg_launch_us']) for n in k if f and f in n])"
  done
done
