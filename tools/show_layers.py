import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
    L=d['roofline']['layers_us']
    print(f, d['ms_per_step'], {k:L[k] for k in L if 'layer1' in k and 'dgrad' in k})
