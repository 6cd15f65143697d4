#!/bin/bash
# (GPU box) VERDICT r05 item 6: the replayed step with and without 60-us idle gaps behind the 3x3 weight gradients —
# step-level table (no profiler), then per-kernel durations from two rocprofv3 kernel traces.  tools/power_idle_ab.sh [gap_us]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
gap=${1:-60}
python3 tools/power_idle_ab.py --gaps 0,30,$gap,120 --rounds 4 --steps 150
for v in 0 $gap; do
  O=gpurun_out/idle_$v
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/trace --output-format rocpd -- python3 tools/power_idle_ab.py --trace $v --steps 40 > $O/trace.log 2>&1
  python3 tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) gpurun_out/idle_$v.csv > /dev/null
  tail -1 $O/trace.log
done
python3 - <<PY
import csv
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['kernel'][:70]]=(int(r['calls']), float(r['total_ms']), float(r['avg_us']))
    return d
A=load('gpurun_out/idle_0.csv'); B=load('gpurun_out/idle_$gap.csv')
keys=sorted(set(A)|set(B), key=lambda k:-(A.get(k,(0,0,0))[1]+B.get(k,(0,0,0))[1]))
print(f"{'kernel (avg us per launch)':70s} {'calls':>6s} {'no gap':>9s} | {'calls':>6s} {'gap $gap us':>9s}   change")
for k in keys[:30]:
    a_=A.get(k,(0,0,0)); b_=B.get(k,(0,0,0))
    ch = f"{(b_[2]/a_[2]-1)*100:+.1f} %" if a_[2] and b_[2] else ""
    print(f"{k:70s} {a_[0]:6d} {a_[2]:9.2f} | {b_[0]:6d} {b_[2]:9.2f}   {ch}")
PY
