"""Per-launch timeline of ONE training step from a rocprofv3 rocpd database (kernel name, duration, gap to the
previous kernel), the step being delimited by two launches of the input-conversion kernel.
    python tools/rocpd_timeline.py x_results.db [out.txt] [marker-substring]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\bprimia::", "", name)
    name = re.sub(r"void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    marker = sys.argv[3] if len(sys.argv) > 3 else "nchw_to_nhwc"
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(idx) < 3:
        raise SystemExit("fewer than 3 steps in the trace")
    a, b = idx[-2], idx[-1]
    lines = [f"{'#':>3s} {'kernel':70s} {'us':>8s} {'gap us':>7s}"]
    prev_end = None
    tot = 0.0
    for i, (n, s, e) in enumerate(rows[a:b]):
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        lines.append(f"{i:3d} {short(n):70s} {(e - s) / 1e3:8.2f} {gap:7.2f}")
        prev_end = e
        tot += (e - s) / 1e3
    lines.append(f"step: {len(rows[a:b])} launches, kernel time {tot:.1f} us, wall {(rows[b][1] - rows[a][1]) / 1e3:.1f} us")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
