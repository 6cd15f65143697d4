"""Phase switches of conv_s2lh_kernel (GPU box): us per paired forward / data-gradient launch with parts of the kernel off
(option s2lh_dbg: 1 no halo DMA after the prologue, 2 no weight DMA, 4 no MFMA, 8 no stores, 16 no write-back at all, 32 no
BatchNorm partials, 64 forward halo addressed linearly instead of gathered).  python tools/s2lh_phases.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primia_amd import _lib

# the phase switches exist in probe builds only (python -m primia_amd.build --probe); the shipped library refuses s2lh_dbg
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libprimia_probe.so")
from primia_amd._lib import ConvDesc, call, query

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_lib.set_option("s2lh", 7)
dtype = torch.bfloat16; dt = _lib.dtype_code(dtype); dev = torch.device("cuda:0")

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for name, H, C, K in [("layer2.0", 56, 64, 128), ("layer3.0", 28, 128, 256), ("layer4.0", 14, 256, 512)]:
    d1 = ConvDesc.make(N, H, H, C, K, 3, 3, 2, 1); dd = ConvDesc.make(N, H, H, C, K, 1, 1, 2, 0)
    x = torch.randn(N * H * H, C, device=dev).clamp_min(0).to(dtype)
    dy1 = (torch.randn(N * d1.Ho * d1.Wo, K, device=dev) * 1e-2).to(dtype); dyd = dy1.clone()
    wf1 = (torch.randn(query("primia_conv_wfwd_elems", d1), device=dev) * 0.05).to(dtype)
    wg1 = wf1.clone()
    wfd = (torch.randn(query("primia_conv_wfwd_elems", dd), device=dev) * 0.05).to(dtype); wgd = wfd.clone()
    s1 = query("primia_conv_stat_slots_for", d1, dt)
    q1 = torch.zeros(s1, 2, K, device=dev); qd = torch.zeros(s1, 2, K, device=dev)
    y1 = torch.empty(N * d1.Ho * d1.Wo, K, dtype=dtype, device=dev); yd = torch.empty_like(y1)
    dx = torch.empty(N * H * H, C, dtype=dtype, device=dev)
    out = []
    for dbg in (0, 1, 2, 3, 4, 8, 32, 64, 15):
        _lib.set_option("s2lh_dbg", dbg)
        tf = timeit(lambda: call("primia_conv2d_fwd_stats_pair", d1, x, wf1, y1, q1, dd, wfd, yd, qd, dt))
        tg = timeit(lambda: call("primia_conv2d_dgrad_pair", d1, dy1, wg1, dd, dyd, wgd, dx, dt))
        out.append(f"dbg={dbg}: {tf:.0f}/{tg:.0f}")
    _lib.set_option("s2lh_dbg", 0)
    print(name, "fwd/dgrad us |", " | ".join(out))
