#!/bin/bash
# Where do the matrix kernels wait?  Three PMC passes over the bench (no trace domains): wave-level wait / active cycles by
# instruction class, queue-full and LDS conflict cycles, instruction-fetch and texture-addresser stalls.
#   bash tools/pmc_stalls.sh r04   ->  profiles/r04_stall_counters.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r04}
B="python3 bench.py --steps 3 --warmup 1 --no-graph --no-secure --no-cpu-baseline --sustain-s 0"
O=gpurun_out/stalls_$R
rm -rf $O; mkdir -p $O profiles
out=profiles/${R}_stall_counters.txt
: > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $O/p$i --output-format rocpd -- $B > $O/p$i.log 2>&1
  echo "== pass $i: $set" >> $out
  python3 tools/rocpd_pmc.py $(find $O/p$i -name "*.db" | head -1) kernel | grep -i "lh2\|patch33\|c64\|igemm\|wgrad_tap\|stem_bwd\|stem_conv" | cut -c1-330 >> $out
done
rm -rf $O            # (the three databases exceed what gpurun merges back)
cp $out gpurun_out/
cat $out | cut -c1-300
