# in-step A/B of conv_s2lh_kernel (option s2lh, bits: 1 data gradient of <= 128-channel dx, 2 forward, 4 data gradient everywhere):
# images/s, ms per step and the median launch of every transition-block convolution call (us)
for r in 1 2; do for v in ${S2LH_VALUES:-0 1 7}; do
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secure --sustain-s 0 --opt s2lh=$v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); L=d['roofline']['layers_us']; print('s2lh=$v', d['value'], d['ms_per_step'], {k: L[k] for k in L if ('.0.conv1' in k or 'downsample' in k) and 'layer1' not in k})"
done; done
