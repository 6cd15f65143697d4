"""Where conv3x3_c64_kernel's waves spend their cycles (GPU box; build tools/micro/libprimia_probe.so with tools/micro/c64_probe.sh
first): layer1's forward convolution at batch 256 on the probe build, per-wave s_memtime sums of the loop's phases.
    python tools/micro/c64_probe.py [c64_dbg bits] [fwd | dgrad | dgrad_bn | acc | acc2 | acc3]"""
import ctypes, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from primia_amd import _lib

_lib.LIB_PATH = os.path.join(HERE, "libprimia_probe.so")      # before the first call loads it
from primia_amd._lib import ConvDesc, call, query, lib

dev = torch.device("cuda:0")
dt = _lib.dtype_code(torch.bfloat16)
if len(sys.argv) > 1:
    _lib.set_option("c64_dbg", int(sys.argv[1]))
N, H, C = 256, 56, 64
d = ConvDesc.make(N, H, H, C, C, 3, 3, 1, 1)
w = (torch.randn(C, C, 3, 3) * 0.05).to(torch.bfloat16)
wf = torch.empty(query("primia_conv_wfwd_elems", d), dtype=torch.bfloat16, device=dev)
wd = torch.empty(query("primia_conv_wdgrad_elems", d), dtype=torch.bfloat16, device=dev)
call("primia_conv_weight_prepare", d, C, w.to(dev), wf, wd, dt)
x = torch.relu(torch.randn(N * H * H, C, device=dev)).to(torch.bfloat16)
y = torch.empty_like(x)
slots = query("primia_conv_stat_slots_for", d, dt)
sums = torch.empty(slots * 2 * C, device=dev)
variant = sys.argv[2] if len(sys.argv) > 2 else "fwd"
M = N * H * H
ybn = (torch.randn(M, C, device=dev) * 1.2).to(torch.bfloat16)
mask = torch.randint(0, 256, (M * C // 8,), device=dev, dtype=torch.int32).to(torch.uint8)
mask2 = torch.randint(0, 256, (M * C // 8,), device=dev, dtype=torch.int32).to(torch.uint8)
c0, c1 = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
sl2 = max(query("primia_conv_dgrad_bnsums_slots", d, dt), query("primia_conv_dgrad_masked_acc_bnsums_slots", d, dt), 1)
sums2 = torch.empty(sl2 * 2 * C, device=dev)
run = {
    "fwd": lambda: call("primia_conv2d_fwd_stats", d, x, wf, y, sums, dt),
    "dgrad": lambda: call("primia_conv2d_dgrad", d, x, wd, y, 0, dt),
    "dgrad_bn": lambda: call("primia_conv2d_dgrad_bnsums", d, x, wd, y, ybn, c0, c1, c1, c0, sums2, dt),
    "acc": lambda: call("primia_conv2d_dgrad_masked_acc", d, x, wd, y, mask, dt),
    "acc2": lambda: call("primia_conv2d_dgrad_masked_acc_bnsums", d, x, wd, y, mask, 2, ybn, mask2, c0, c1, sums2, dt),
    "acc3": lambda: call("primia_conv2d_dgrad_masked_acc_bnsums", d, x, wd, y, mask, 3, ybn, None, c0, c1, sums2, dt),
}[variant]
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print(f"{variant}: launch {e0.elapsed_time(e1) * 100:.1f} us (probe build)")
n = slots * 4 * 16
buf = (ctypes.c_ulonglong * n)()
fn = lib().primia_c64_probe_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(ctypes.cast(buf, ctypes.c_void_p), n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(slots, 4, 16).astype(np.float64)
it = a[:, :, 10]
names = ["top wait", "barrier", "rows + requests", "stores + sums", "matrix loop", "tail (rest)", "tail: old-row wait", "tail: old rows + add", "tail: next old rows"]
per = a[:, :, :9] / it[:, :, None]
print(f"{slots} blocks x 4 waves, {it.mean():.1f} patches per block; cycles per patch and wave (mean over blocks | wave 0..3):")
for i, nm in enumerate(names):
    print(f"  {nm:14s} {per[:, :, i].mean():8.0f}   | " + " ".join(f"{per[:, wv, i].mean():7.0f}" for wv in range(4)))
print(f"  {'sum':14s} {per.sum(-1).mean():8.0f}")
