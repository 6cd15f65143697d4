# PMC pass over the patch weight-gradient kernel alone (one layer shape), SQ counters only
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
L=${1:-l2.3x3}
for v in ${WGP_PHASES:-0 3}; do
  rm -rf gpurun_out/pmc_wgp_$v
  PRIMIA_WGP_NOEPI=$v rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
     -d gpurun_out/pmc_wgp_$v --output-format rocpd -- python3 tools/conv_layers.py 256 bf16 $L > /dev/null 2>&1
  db=$(find gpurun_out/pmc_wgp_$v -name "*.db" | head -1)
  echo "== NOEPI=$v"
  python3 tools/rocpd_pmc.py $db wgrad_patch
done
