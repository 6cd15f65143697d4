"""Same-box A/B of the transition blocks' paired launches (GPU box): conv1 3x3/2 + downsample 1x1/2 forward with BatchNorm
partial sums (primia_conv2d_fwd_stats_pair), their data gradients (primia_conv2d_dgrad_pair) and weight gradients
(primia_conv2d_wgrad_pair_ws) under primia_set_option(OPT, 0 | 1) — default OPT = s2lh (conv_s2lh_kernel vs the implicit GEMM).
    python tools/transition_bench.py [N] [OPT]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primia_amd import _lib
from primia_amd._lib import ConvDesc, call, query

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
OPT = sys.argv[2] if len(sys.argv) > 2 else "s2lh"
dtype = torch.bfloat16
dt = _lib.dtype_code(dtype)
dev = torch.device("cuda:0")
layers = [("layer2.0", 56, 64, 128), ("layer3.0", 28, 128, 256), ("layer4.0", 14, 256, 512)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


tot = {}
for name, H, C, K in layers:
    d1 = ConvDesc.make(N, H, H, C, K, 3, 3, 2, 1)
    dd = ConvDesc.make(N, H, H, C, K, 1, 1, 2, 0)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N * H * H, C, device=dev, generator=g).clamp_min(0).to(dtype)        # post-ReLU operand
    dy1 = (torch.randn(N * d1.Ho * d1.Wo, K, device=dev, generator=g) * 1e-2).to(dtype)
    dyd = (torch.randn(N * d1.Ho * d1.Wo, K, device=dev, generator=g) * 1e-2).to(dtype)
    w1 = torch.randn(K, C, 3, 3, device=dev, generator=g) * 0.05
    wd_ = torch.randn(K, C, 1, 1, device=dev, generator=g) * 0.05
    wf1 = torch.empty(query("primia_conv_wfwd_elems", d1), dtype=dtype, device=dev)
    wg1 = torch.empty(query("primia_conv_wdgrad_elems", d1), dtype=dtype, device=dev)
    wfd = torch.empty(query("primia_conv_wfwd_elems", dd), dtype=dtype, device=dev)
    wgd = torch.empty(query("primia_conv_wdgrad_elems", dd), dtype=dtype, device=dev)
    call("primia_conv_weight_prepare", d1, C, w1, wf1, wg1, dt)
    call("primia_conv_weight_prepare", dd, C, wd_, wfd, wgd, dt)
    fl = 2.0 * N * d1.Ho * d1.Wo * K * C * 10
    res = {}
    for v in (0, 1):
        _lib.set_option(OPT, 7 * v if OPT == "s2lh" else v)
        s1 = query("primia_conv_stat_slots_for", d1, dt)
        sd = query("primia_conv_stat_slots_for", dd, dt)
        q1 = torch.zeros(s1, 2, K, device=dev)
        qd = torch.zeros(sd, 2, K, device=dev)
        y1 = torch.empty(N * d1.Ho * d1.Wo, K, dtype=dtype, device=dev)
        yd = torch.empty_like(y1)
        dx = torch.empty(N * H * H, C, dtype=dtype, device=dev)
        tf = timeit(lambda: call("primia_conv2d_fwd_stats_pair", d1, x, wf1, y1, q1, dd, wfd, yd, qd, dt))
        tg = timeit(lambda: call("primia_conv2d_dgrad_pair", d1, dy1, wg1, dd, dyd, wgd, dx, dt))
        need = query("primia_conv_wgrad_pair_ws_bytes", d1, dd, dt)
        tw = 0.0
        a1 = torch.zeros(query("primia_conv_wfwd_elems", d1), device=dev)
        ad = torch.zeros(query("primia_conv_wfwd_elems", dd), device=dev)
        if need > 0:
            ws = torch.empty(need // 4, device=dev)
            tw = timeit(lambda: call("primia_conv2d_wgrad_pair_ws", d1, x, dy1, a1, dd, dyd, ad, ws, need, dt))
        res[v] = (tf, tg, tw, y1.clone(), yd.clone(), dx.clone(), q1.double().sum(0), qd.double().sum(0), a1.clone(), ad.clone())
        for k, t in (("fwd", tf), ("dgrad", tg), ("wgrad", tw)):
            tot[(v, k)] = tot.get((v, k), 0.0) + t
    a, b = res[0], res[1]
    print(f"{name}: fwd {a[0]:6.1f} -> {b[0]:6.1f} us ({fl / b[0] / 1e6:5.0f} TF) | dgrad {a[1]:6.1f} -> {b[1]:6.1f} us "
          f"({fl / b[1] / 1e6:5.0f} TF) | wgrad {a[2]:6.1f} -> {b[2]:6.1f} us ({fl / max(b[2], 1e-9) / 1e6:5.0f} TF) | "
          f"relerr y1 {relerr(b[3], a[3]):.1e} yd {relerr(b[4], a[4]):.1e} dx {relerr(b[5], a[5]):.1e} "
          f"stats {relerr(b[6], a[6]):.1e} {relerr(b[7], a[7]):.1e} dw {relerr(b[8], a[8]):.1e} {relerr(b[9], a[9]):.1e}")
print(f"{OPT}: 0 -> 1 per step: " + " | ".join(f"{k} {tot[(0, k)]:.0f} -> {tot[(1, k)]:.0f} us" for k in ("fwd", "dgrad", "wgrad")))
