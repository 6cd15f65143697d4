#!/bin/bash
# (GPU box) builds conv_s2lh.hip with each set of -D switches given as arguments (quote a set; "" = the defaults), relinks
# the library and prints tools/transition_bench.py's last line per variant — same box, same data.
#   tools/s2lh_variants.sh "" "-DS2_DBG=0" "-DS2_DBG=0 -DS2_PROG_KARG=1"
cd "$(dirname "$0")/.."
C=primia_amd/csrc
cp primia_amd/libprimia_hip.so /tmp/_shipped.so
trap 'cp /tmp/_shipped.so primia_amd/libprimia_hip.so' EXIT INT TERM     # an interrupted run must not leave a variant installed
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $v -c $C/conv_s2lh.hip -o /tmp/s2v.o 2>/dev/null || { echo "build failed: $v"; continue; }
  objs=$(ls $C/_build/*.o | grep -v conv_s2lh.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o primia_amd/libprimia_hip.so $objs /tmp/s2v.o
  echo "== [$v]"
  python tools/transition_bench.py 2>/dev/null | tail -4 | cut -c1-120
done
cp /tmp/_shipped.so primia_amd/libprimia_hip.so
