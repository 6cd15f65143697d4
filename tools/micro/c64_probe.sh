#!/bin/bash
# Builds tools/micro/libprimia_probe.so = the library with conv3x3_c64.hip compiled -DC64_PROBE=1 (s_memtime probes at the phase
# boundaries of the kernel's loop).  Run here (no GPU needed), then on the GPU box: python tools/micro/c64_probe.py
cd "$(dirname "$0")/../.."
OBJ=primia_amd/_obj
[ -d $OBJ ] || OBJ=$(python3 -c "from primia_amd import build; print(build.OBJ)")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DC64_PROBE=1 -DPRIMIA_PROBE=1 ${C64_DEFS} -c primia_amd/csrc/conv3x3_c64.hip -o /tmp/c64_probe.o || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPRIMIA_PROBE=1 -c primia_amd/csrc/options.hip -o /tmp/options_probe.o || exit 1
objs=$(ls $OBJ/*.o | grep -v "probe.o" | grep -v conv3x3_c64.o | grep -v options.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/micro/libprimia_probe.so $objs /tmp/c64_probe.o /tmp/options_probe.o && echo built tools/micro/libprimia_probe.so
