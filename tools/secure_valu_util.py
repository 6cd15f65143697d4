"""Vector-ALU utilisation of the FSS kernels from a rocprofv3 PMC pass (tools/pmc_secure.sh):
    python tools/secure_valu_util.py results.db > profiles/r04_secure_valu_pmc.json
SQ_* counters are summed over the chip and count quad-cycles (MI355X_MICROARCH.md): per wave,
  valu_active = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES   (share of a wave's life with a VALU instruction executing)
and chip-wide  valu_util = SQ_INSTS_VALU * 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of the chip's
VALU issue capacity (one wave64 instruction per 4 cycles per SIMD) the kernel used — the measured counterpart of the
instruction-count model in tools/bench_secure.py (80 rounds x ~56 32-bit operations per SHA-512 compression).  (On this
rocprofv3 build SQ_ACTIVE_INST_VALU reports the same number as SQ_INSTS_VALU: an instruction count.)"""
import json
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), sum(duration) "
                   "from counters_collection group by kernel_name, counter_name").fetchall()
tab = {}
for k, c, v, n, d in rows:
    m = re.search(r"\b(dif_eval_local_kernel|dif_eval_kernel|dif_keygen_kernel|dpf_eval_kernel|dpf_keygen_kernel)\b", k)
    if m:
        tab.setdefault(m.group(1), {})[c] = (v / n, n, d / n)
out = {}
for k, t in tab.items():
    rec = {"launches": t["SQ_WAVE_CYCLES"][1], "avg_us": round(t["SQ_WAVE_CYCLES"][2] / 1e3, 2)}
    for c, (v, _, _) in t.items():
        rec[c] = int(v)
    if "SQ_ACTIVE_INST_VALU" in t and "SQ_WAVE_CYCLES" in t:
        rec["valu_active_per_wave_cycle"] = round(t["SQ_ACTIVE_INST_VALU"][0] / t["SQ_WAVE_CYCLES"][0], 4)
    if "SQ_INSTS_VALU" in t and "GRBM_GUI_ACTIVE" in t:
        simd_cycles = t["GRBM_GUI_ACTIVE"][0] / 8 * 1024
        rec["valu_insts_per_simd_cycle"] = round(t["SQ_INSTS_VALU"][0] / simd_cycles, 4)
        # a wave64 vector instruction of this integer code occupies its SIMD for 4 cycles (the kernels sit at 0.26-0.27
        # instructions per SIMD and cycle): share of the chip's issue capacity these instructions used
        rec["valu_issue_util"] = round(t["SQ_INSTS_VALU"][0] * 4 / simd_cycles, 4)
    out[k] = rec
out["_note"] = ("rocprofv3 --pmc over `python tools/bench_secure.py --only-fss-roofline` (2^20 comparisons per launch); SQ counters "
                "are chip sums in quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs")
print(json.dumps(out, indent=1))
