"""Stem forward tail at the benchmark's size (batch 256, 224 x 224, bf16): the chain on a stored conv1 output against the two
passes over the input that never store it (csrc/stem_fwd_fused.hip).   python tools/stem_fwd_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primia_amd import _lib
from primia_amd._lib import call, query

N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 224
dev, dt = torch.device("cuda:0"), _lib.dtype_code(torch.bfloat16)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 3, S, S, generator=g).to(dev)
xp = torch.zeros(N * (S + 6) * (S + 8), 4, dtype=torch.bfloat16, device=dev)
call("primia_nchw_to_nhwc_padded", x, xp, N, 3, S, S, 4, 3, 3, S + 6, S + 8, dt)
w = (torch.randn(64, 256, generator=g) * 0.05).to(torch.bfloat16).to(dev)
Ho, Hq = S // 2, S // 4
M = N * Ho * Ho
y = torch.empty(M, 64, dtype=torch.bfloat16, device=dev)
slots = query("primia_stem_conv_stat_slots", N, S, S)
sums = torch.zeros(slots, 2, 64, device=dev)
gamma, beta = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3
rm, rv, sm, si = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.empty(64, device=dev), torch.empty(64, device=dev)
p = torch.empty(N * Hq * Hq, 64, dtype=torch.bfloat16, device=dev)
am = torch.empty(N * Hq * Hq, 64, dtype=torch.uint8, device=dev)

def timed(name, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:58s} {us:8.1f} us")
    return us

a = timed("stem_conv_fwd_stats (writes y, 411 MB)", lambda: call("primia_stem_conv_fwd_stats", xp, w, y, sums, N, S, S, dt))
b = timed("bn_relu_maxpool_fwd_from_sums (reads y)", lambda: call("primia_bn_relu_maxpool_fwd_from_sums", y, p, am, gamma, beta, rm, rv, sm, si, sums, slots, N, Ho, Ho, 64, 1e-5, 0.1, dt))
c = timed("stem_conv_stats (nothing stored)", lambda: call("primia_stem_conv_stats", xp, w, sums, N, S, S, dt))
d = timed("bn_finalize_stats", lambda: call("primia_bn_finalize_stats", sums, slots, M, 64, 1e-5, 0.1, rm, rv, sm, si))
e = timed("stem_conv_pool (pooled + codes only)", lambda: call("primia_stem_conv_pool", xp, w, None, p, am, gamma, beta, sm, si, N, S, S, dt))
f = timed("stem_conv_pool (+ y stored)", lambda: call("primia_stem_conv_pool", xp, w, y, p, am, gamma, beta, sm, si, N, S, S, dt))
print(f"chain {a + b:.1f} us | two passes over the input {c + d + e:.1f} us | with y stored in pass 2 {c + d + f:.1f} us")
