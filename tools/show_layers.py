import json,sys
pat = sys.argv[1]
for f in sys.argv[2:]:
    d=json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
    L=d['roofline']['layers_us']
    print(f, d['ms_per_step'], {k:L[k] for k in L if any(p in k for p in pat.split(','))})
