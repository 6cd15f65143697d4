"""Per-layer timing of the ResNet-18 convolution kernels at batch N (GPU box).
    python tools/conv_layers.py [N] [dtype]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primia_amd import _lib
from primia_amd._lib import ConvDesc, call, query

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
dt = _lib.dtype_code(dtype)
dev = torch.device("cuda:0")
layers = [("stem", 224, 4, 64, 7, 2, 3, 1), ("l1.3x3", 56, 64, 64, 3, 1, 1, 4), ("l2.0.c1", 56, 64, 128, 3, 2, 1, 1),
          ("l2.ds", 56, 64, 128, 1, 2, 0, 1), ("l2.3x3", 28, 128, 128, 3, 1, 1, 3), ("l3.0.c1", 28, 128, 256, 3, 2, 1, 1),
          ("l3.ds", 28, 128, 256, 1, 2, 0, 1), ("l3.3x3", 14, 256, 256, 3, 1, 1, 3), ("l4.0.c1", 14, 256, 512, 3, 2, 1, 1),
          ("l4.ds", 14, 256, 512, 1, 2, 0, 1), ("l4.3x3", 7, 512, 512, 3, 1, 1, 3)]

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for kv in os.environ.get("CONV_LAYERS_OPTS", "").split():      # e.g. CONV_LAYERS_OPTS="wgp_shape=1 wgp_lw=0"
    k, _, v = kv.partition("=")
    _lib.set_option(k, int(v))
only = sys.argv[3] if len(sys.argv) > 3 else None
if only:
    layers = [l for l in layers if l[0] == only]
tot = {"fwd": 0, "dgrad": 0, "wgrad": 0}
print(f"{'layer':9s} {'cnt':>3s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
for name, H, C, K, R, s, p, cnt in layers:
    d = ConvDesc.make(N, H, H, C, K, R, R, s, p)
    creal = 3 if name == "stem" else C
    x = torch.randn(N * H * H, C, device=dev).to(dtype)
    dy = torch.randn(N * d.Ho * d.Wo, K, device=dev).to(dtype)
    y = torch.empty_like(dy); dx = torch.empty_like(x)
    w = torch.randn(K, creal, R, R, device=dev) * 0.05
    wf = torch.empty(query("primia_conv_wfwd_elems", d), dtype=dtype, device=dev)
    wd = torch.empty(query("primia_conv_wdgrad_elems", d), dtype=dtype, device=dev) if name != "stem" else None
    call("primia_conv_weight_prepare", d, creal, w, wf, wd, dt)
    acc = torch.zeros(query("primia_conv_wfwd_elems", d), dtype=torch.float32, device=dev)
    fl = 2.0 * N * d.Ho * d.Wo * K * creal * R * R
    if name == "stem" and dtype == torch.bfloat16:   # the padded-input stem kernels the engine uses
        xp = torch.zeros(N * (H + 6) * (H + 8), 4, dtype=dtype, device=dev)
        tf = timeit(lambda: call("primia_stem_conv_fwd", xp, wf, y, N, H, H, dt))
        td = 0.0
        tw = timeit(lambda: call("primia_stem_conv_wgrad", xp, dy, acc, N, H, H, dt))
    else:
        tf = timeit(lambda: call("primia_conv2d_fwd", d, x, wf, y, dt))
        td = timeit(lambda: call("primia_conv2d_dgrad", d, dy, wd, dx, 0, dt)) if wd is not None else 0.0
        wsb = query("primia_conv_wgrad_ws_bytes", d, dt)
        if wsb > 0 and not os.environ.get("CONV_LAYERS_ATOMIC_WGRAD"):   # the engine's path: slabs + ordered reduce
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
            tw = timeit(lambda: call("primia_conv2d_wgrad_ws", d, x, dy, acc, ws, wsb, dt))
        else:
            tw = timeit(lambda: call("primia_conv2d_wgrad", d, x, dy, acc, dt))
        if os.environ.get("CONV_LAYERS_ACC") and wd is not None:   # accumulate-form data gradient instead
            td = timeit(lambda: call("primia_conv2d_dgrad", d, dy, wd, dx, 1, dt))
    f = lambda t: fl / (t * 1e-3) / 1e12 if t > 0 else 0
    print(f"{name:9s} {cnt:3d} {fl/1e9:7.1f} | {tf*1e3:8.1f} {f(tf):6.0f} | {td*1e3:8.1f} {f(td):6.0f} | {tw*1e3:8.1f} {f(tw):6.0f}")
    tot["fwd"] += tf * cnt; tot["dgrad"] += td * cnt; tot["wgrad"] += tw * cnt
print("per-step ms:", {k: round(v, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()), 3))
