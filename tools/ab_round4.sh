for i in 1 2 3 4 5; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round 5 defaults      ', d['value'], d['ms_per_step'])"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 --opt s2lh=0 --opt wgp_lw=0 --opt dgrad_cls_inner=0 --engine-opt dgrad_bnsums=false --engine-opt pair_bnsums=false --engine-opt acc_bnsums=false --engine-opt head_bnsums=false 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round 4 configuration', d['value'], d['ms_per_step'])"
done
