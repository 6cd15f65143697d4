#!/bin/bash
# same-box A/B of library builds over the training bench: tools/ab_libs.sh ROUNDS lib1.so lib2.so ...  (`base` = the shipped
# library); alternating runs, 200 timed steps each, prints images/s, ms per step and the layer1 / stem launches per run.
# The libraries are loaded with bench.py --lib: the shipped one is never overwritten.
rounds=$1; shift
B="python bench.py --no-secure --no-cpu-baseline --sustain-s 0 --steps 200"
for i in $(seq $rounds); do
  for lib in "$@"; do
    if [ "$lib" = base ]; then L=""; else L="--lib $lib"; fi
    $B $L 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); l=d['roofline']['layers_us']
c64=sum(v for k,v in l.items() if 'layer1' in k and not k.startswith('wgrad'))
print('$lib', d['value'], d['ms_per_step'], 'layer1 fwd+dgrad us', round(c64,1), 'stem wgrad', l.get('wgrad:conv1'), 'stem fwd', l.get('fwd:conv1'))"
  done
done
