"""Summarise a rocprofv3 rocpd SQLite database: per-kernel calls / total / avg / share.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\bprimia::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"\"{short(n)}\",{c},{t/1e6:.3f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*t/total:.2f}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
