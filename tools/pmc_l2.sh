# L2 hit rate and memory-side traffic of the convolution kernels for one layer shape (separate PMC passes)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=${1:-l2.3x3}
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" ; do
  tag=$(echo $c | tr ' ' '_')
  rm -rf gpurun_out/pmc_l2_$tag
  rocprofv3 --pmc $c -d gpurun_out/pmc_l2_$tag --output-format rocpd -- python3 tools/conv_layers.py 256 bf16 $L > gpurun_out/pmc_l2_$tag.log 2>&1
  db=$(find gpurun_out/pmc_l2_$tag -name "*.db" | head -1)
  echo "== $c"
  python3 tools/rocpd_pmc.py $db ${2:-conv}
done
