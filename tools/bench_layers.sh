python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 2>/dev/null | tail -1 > gpurun_out/bench_layers.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_layers.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
L=d['roofline']['layers_us']
for k,v in L.items(): print(f"{k:32s} {v:8.1f}")
for k,v in d['roofline']['kernels'].items(): print(k[:60], v)
PY
