# phase switches of the patch weight-gradient kernel (PRIMIA_WGP_NOEPI bits: 1 no epilogue, 2 no staging, 4 no compute)
cd "$(dirname "$0")/.."
for k in ${WGP_KERNELS:-1 0}; do
for v in ${WGP_PHASES:-0 1 2 3 5}; do
  echo "== PRIMIA_WGP32=$k PRIMIA_WGP_NOEPI=$v"
  for l in l1.3x3 l2.3x3 l3.3x3 l4.3x3; do
    PRIMIA_WGP32=$k PRIMIA_WGP_NOEPI=$v python tools/conv_layers.py 256 bf16 $l 2>/dev/null | grep "^l" | awk '{print "   ", $1, "wgrad_us", $(NF-1)}'
  done
done
done
