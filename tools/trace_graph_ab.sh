#!/bin/bash
# (GPU box) kernel trace of the REPLAYED training step under primia_set_option(OPT, A) and (OPT, B): per-kernel average
# durations side by side, and the sum per step — what a change does to the OTHER kernels of the step (clock, caches) shows
# here and nowhere else.   tools/trace_graph_ab.sh OPT A B [steps]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
opt=$1; a=$2; b=$3; steps=${4:-30}
for v in $a $b; do
  O=gpurun_out/tg_${opt}_$v
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats -d $O/trace --output-format rocpd -- python3 bench.py --steps $steps --warmup 5 --no-secure --no-cpu-baseline --sustain-s 0 --opt $opt=$v > $O/trace.log 2>&1
  python3 tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) gpurun_out/tg_${opt}_$v.csv > /dev/null
  tail -1 $O/trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$opt=$v', d['value'], d['ms_per_step'])"
done
python3 - <<PY
import csv
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        d[r['kernel'][:70]]=(int(r['calls']), float(r['total_ms']), float(r['avg_us']))
    return d
A=load('gpurun_out/tg_${opt}_$a.csv'); B=load('gpurun_out/tg_${opt}_$b.csv')
keys=sorted(set(A)|set(B), key=lambda k:-(A.get(k,(0,0,0))[1]+B.get(k,(0,0,0))[1]))
ta=sum(v[1] for v in A.values()); tb=sum(v[1] for v in B.values())
print(f"total kernel ms: {ta:.2f} -> {tb:.2f}")
for k in keys[:28]:
    a_=A.get(k,(0,0,0)); b_=B.get(k,(0,0,0))
    print(f"{k:70s} {a_[0]:6d} {a_[2]:9.2f} | {b_[0]:6d} {b_[2]:9.2f}  total {a_[1]:8.2f} -> {b_[1]:8.2f}")
PY
