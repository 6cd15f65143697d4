"""Does a replayed hipGraph pay for ALTERNATING between the convolution kernels (64-158 KB of dynamic LDS) and the small
element-wise kernels?  rocprofv3's timeline of the replayed step shows ~5.7 us between a convolution and the BatchNorm kernel
next to it and 0.00 between two small kernels; this times three graphs WITHOUT the profiler:
    A: n x conv      B: n x small      C: n x (conv, small)         ->  T(C) - T(A) - T(B) = n x (cost of alternating)
(GPU box)  python tools/micro/graph_gap.py"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from primia_amd import _lib
from primia_amd._lib import ConvDesc, call, query

dev = torch.device("cuda:0")
dt = _lib.dtype_code(torch.bfloat16)


def make_conv(N, H, C, K):
    d = ConvDesc.make(N, H, H, C, K, 3, 3, 1, 1)
    w = (torch.randn(K, C, 3, 3) * 0.05).to(torch.bfloat16)
    wf = torch.empty(query("primia_conv_wfwd_elems", d), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(query("primia_conv_wdgrad_elems", d), dtype=torch.bfloat16, device=dev)
    call("primia_conv_weight_prepare", d, C, w.to(dev), wf, wd, dt)
    x = torch.randn(N * H * H, C, device=dev).to(torch.bfloat16)
    y = torch.empty(N * H * H, K, dtype=torch.bfloat16, device=dev)
    return lambda: call("primia_conv2d_fwd", d, x, wf, y, dt), query("primia_conv_kernel_id", d, 0, dt)


def timed(fn_list, n, reps=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fn_list:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                for f in fn_list:
                    f()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6


n = 40
a = torch.randn(1 << 16, device=dev)
b = torch.empty_like(a)
small = lambda: torch.mul(a, 2.0, out=b)
for name, (N, H, C, K) in {"c64 (68 KB LDS)": (8, 56, 64, 64), "lh2/lh4 (158 KB)": (16, 14, 256, 256), "lh4 layer4": (32, 7, 512, 512)}.items():
    conv, kid = make_conv(N, H, C, K)
    ta, tb, tc = timed([conv], n), timed([small], n), timed([conv, small], n)
    print(f"{name:18s} kernel id {kid}: per launch conv {ta / n:7.2f} us, small {tb / n:6.2f} us, alternating pair {tc / n:7.2f} us"
          f" -> alternating costs {(tc - ta - tb) / n:6.2f} us per pair")
