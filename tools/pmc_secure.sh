#!/bin/bash
# PMC pass over the FSS kernels of the encrypted-inference path (dif_keygen_kernel, dif_eval_kernel, dif_eval_local_kernel) on their own:
# vector-ALU busy / issue counters behind encrypted_inference.roofline (SQ counters only, no trace domains).
#   bash tools/pmc_secure.sh   ->  profiles/r06_secure_valu_pmc.json
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/pmc_secure
rm -rf $O; mkdir -p $O profiles
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d $O/a --output-format rocpd -- python3 tools/bench_secure.py --only-fss-roofline > $O/a.log 2>&1
python3 tools/secure_valu_util.py $(find $O/a -name "*.db" | head -1) > profiles/r06_secure_valu_pmc.json
cp profiles/r06_secure_valu_pmc.json gpurun_out/
cat profiles/r06_secure_valu_pmc.json
