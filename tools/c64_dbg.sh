#!/bin/bash
# layer1 data-gradient launch times under conv3x3_c64_kernel's timing-experiment bits (option c64_dbg: results are wrong when
# set).  The shipped library refuses the option: build the probe library first (here or on the GPU box):
#   python -m primia_amd.build --probe
for o in ${C64_DBG_BITS:-0 8 16 24 64 88}; do
  python bench.py --lib tools/micro/libprimia_probe.so --steps 40 --warmup 8 --no-cpu-baseline --no-secure --sustain-s 0 --opt c64_dbg=$o 2>/dev/null > gpurun_out/c64dbg_$o.log
done
python tools/show_layers.py dgrad:layer1 gpurun_out/c64dbg_*.log
