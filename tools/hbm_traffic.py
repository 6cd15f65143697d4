"""HBM bytes per launch per kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE):
    python tools/hbm_traffic.py fetch.db write.db > profiles/r01_hbm_traffic_v2.json
FETCH_SIZE is doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md); both counters
are in KiB... rocprofv3 reports them in KB units of 1024 B."""
import json, re, sqlite3, sys

FAMS = ["conv3x3_lh2_kernel", "conv3x3_lh4_kernel", "conv_s2lh_kernel", "conv_wgrad_patch33lw_kernel", "conv_igemm_pair_kernel", "conv_wgrad_patch33_kernel", "conv_wgrad_tap_kernel", "stem_bwd_fused_kernel", "bn_relu_pool_fwd_key_kernel", "conv_wgrad_patch32_kernel", "wgrad_patch32_reduce_kernel", "conv_wgrad_patch_kernel", "wgrad_patch_reduce_kernel", "conv3x3_lh_kernel", "stem_conv_wgrad_kernel", "stem_conv_fwd_kernel", "conv3x3_c64_kernel",
        "conv_igemm_kernel", "conv_wgrad_kernel", "conv_wgrad_dma_kernel", "colreduce2_kernel", "bn_bwd_apply_kernel",
        "bn_apply_kernel", "bn_relu_pool_fwd_kernel"]


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection "
                       "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    out = {}
    for k, v, n in rows:
        for f in FAMS:
            if re.search(r"\b" + f + r"\b", k):
                d = out.setdefault(f, [0.0, 0])
                d[0] += v
                d[1] += n
                break
    return out


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for f in FAMS:
    if f in fetch and f in write:
        fb = fetch[f][0] * 1024 * 2 / fetch[f][1]
        wb = write[f][0] * 1024 / write[f][1]
        res[f] = {"launches": fetch[f][1], "fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb),
                  "hbm_bytes_per_launch": int(fb + wb)}
res["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 4 --warmup 1 --no-graph "
                "--no-secure --no-cpu-baseline; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 reports half of "
                "wide coalesced reads); WRITE_SIZE uncalibrated; averages over all launches of a kernel family")
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from primia_amd.build import source_digest  # noqa: E402
res["_source_sha256"] = source_digest()      # the code these passes ran: bench.py quotes `traffic` only for this digest
print(json.dumps(res, indent=1))
