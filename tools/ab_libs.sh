#!/bin/bash
# same-box A/B of library builds over the training bench: tools/ab_libs.sh ROUNDS lib1.so lib2.so ...  (the first one is
# usually a copy of the shipped library); alternating runs, 200 timed steps each, prints images/s, ms per step and the
# dominant family's TFLOP/s per run
rounds=$1; shift
cp primia_amd/libprimia_hip.so /tmp/_shipped.so
B="python bench.py --no-secure --no-cpu-baseline --sustain-s 0 --steps 200"
for i in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = base ]; then cp /tmp/_shipped.so primia_amd/libprimia_hip.so; else cp $lib primia_amd/libprimia_hip.so; fi
    $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
cp /tmp/_shipped.so primia_amd/libprimia_hip.so
