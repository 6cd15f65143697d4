"""Per-kernel PMC averages from a rocprofv3 rocpd DB: python tools/rocpd_pmc.py db [name-filter]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else "primia"
rows = cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration), grid_size "
                   "from counters_collection group by kernel_name, counter_name, grid_size").fetchall()
tab = {}
for k, c, v, n, d, g in rows:
    if flt not in k: continue
    k = re.sub(r"\bprimia::", "", k); k = re.sub(r"^void ", "", k)[:70]
    tab.setdefault((k, g), {"n": n, "dur_us": d / 1e3})[c] = v / n
for (k, g), d in sorted(tab.items()):
    print(f"{k:70s} grid={g:<8d} n={d['n']:<3d} dur={d['dur_us']:.1f}us " +
          " ".join(f"{c.replace('SQ_', '')}={v:.3g}" for c, v in d.items() if c not in ("n", "dur_us")))
