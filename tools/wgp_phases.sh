cd /root/repo
for v in 0 1 2 3 4 5 6; do
  echo "== PRIMIA_WGP_NOEPI=$v"
  for l in l1.3x3 l2.3x3 l3.3x3 l4.3x3; do
    PRIMIA_WGP_NOEPI=$v python tools/conv_layers.py 256 bf16 $l 2>/dev/null | grep "^l" | awk '{print $1, "wgrad_us", $(NF-1)}'
  done
done
