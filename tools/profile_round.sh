#!/bin/bash
# Evidence passes of one round: kernel trace + stats, MFMA utilisation, HBM traffic (separate PMC passes, as
# MI355X_MICROARCH.md prescribes) over the benchmark's own workload.   bash tools/profile_round.sh r02
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r02}
B="python3 bench.py --steps 4 --warmup 1 --no-graph --no-secure --no-cpu-baseline --sustain-s 0"
O=gpurun_out/prof_$R
rm -rf $O; mkdir -p $O profiles
rocprofv3 --kernel-trace --stats -d $O/trace --output-format rocpd -- $B > $O/trace.log 2>&1
python3 tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) profiles/${R}_bench_kernel_stats.csv > /dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma --output-format rocpd -- $B > $O/mfma.log 2>&1
python3 tools/mfma_util.py $(find $O/mfma -name "*.db" | head -1) > profiles/${R}_mfma_util.json
rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format rocpd -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format rocpd -- $B > $O/write.log 2>&1
python3 tools/hbm_traffic.py $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) > profiles/${R}_hbm_traffic.json
# instruction mix per kernel (vector / matrix / scalar / LDS instructions issued, wave cycles): which kernels are bound by ISSUE
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/insts --output-format rocpd -- $B > $O/insts.log 2>&1
python3 tools/rocpd_pmc.py $(find $O/insts -name "*.db" | head -1) kernel | cut -c1-250 > profiles/${R}_inst_mix_latest.txt
head -12 profiles/${R}_bench_kernel_stats.csv
for f in bench_kernel_stats.csv mfma_util.json hbm_traffic.json inst_mix_latest.txt; do cp profiles/${R}_$f gpurun_out/ 2>/dev/null; done   # (only what this script made: the box's gpurun_out/ is merged back)
