"""MFMA utilisation per kernel family from a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES,
GRBM_GUI_ACTIVE):  python tools/mfma_util.py results.db > profiles/r01_mfma_util.json
util = MFMA-busy cycles summed over the chip / (4 SIMDs x 256 CUs x kernel cycles); GRBM_GUI_ACTIVE comes
summed over the 8 XCDs."""
import json, re, sqlite3, sys

FAMS = ["conv3x3_lh2_kernel", "conv3x3_lh4_kernel", "conv_s2lh_kernel", "conv_wgrad_patch33lw_kernel", "conv_igemm_pair_kernel", "conv_wgrad_patch33_kernel", "conv_wgrad_tap_kernel", "stem_bwd_fused_kernel", "bn_relu_pool_fwd_key_kernel", "conv_wgrad_patch32_kernel", "wgrad_patch32_reduce_kernel", "conv_wgrad_patch_kernel", "wgrad_patch_reduce_kernel", "conv3x3_lh_kernel", "stem_conv_wgrad_kernel", "stem_conv_fwd_kernel", "conv3x3_c64_kernel",
        "conv_igemm_kernel", "conv_wgrad_kernel", "conv_wgrad_dma_kernel"]
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), sum(duration) "
                   "from counters_collection group by kernel_name, counter_name").fetchall()
tab = {}
for k, c, v, n, d in rows:
    for f in FAMS:
        if re.search(r"\b" + f + r"\b", k):
            t = tab.setdefault(f, {})
            t.setdefault(c, [0.0, 0, 0.0])
            t[c][0] += v
            t[c][1] += n
            t[c][2] += d
            break
out = {}
for f, t in tab.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in t:
        continue
    mf, n, dur = t["SQ_VALU_MFMA_BUSY_CYCLES"]
    rec = {"launches": n, "avg_us": round(dur / n / 1e3, 2), "mfma_busy_cycles_per_launch": int(mf / n)}
    if "GRBM_GUI_ACTIVE" in t:
        gui = t["GRBM_GUI_ACTIVE"][0] / t["GRBM_GUI_ACTIVE"][1]
        rec["gpu_cycles_per_launch"] = int(gui)
        rec["mfma_util"] = round(mf / n / (gui / 8 * 4 * 256), 4)  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    out[f] = rec
out["_note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py --steps 4 --warmup 1 --no-graph; "
                "mfma_util = MFMA-busy cycles (summed over all SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)")
print(json.dumps(out, indent=1))
