"""Bandwidth of the data-path kernels at the training batch (GPU box):  python tools/datapipe_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primia_amd import datapipe as P

dev = torch.device("cuda:0")
x = torch.randn(512, 3, 224, 224, device=dev)            # 512 samples -> 256 mixed
y = P.To_one_hot(3)(torch.randint(0, 3, (512,)))
mix = P.MixUp(λ=0.4, p=0.0)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


t_mix = timeit(lambda: mix((x, y)))
t_ms = timeit(lambda: P.calc_mean_std(x))
nb = x.numel() * 4
print(json.dumps({"mixup_512x3x224x224": {"ms": round(t_mix * 1e3, 3), "GB_per_s": round(1.5 * nb / t_mix / 1e9, 1),
                                          "bytes": "read 2 halves + write 1 half"},
                  "mean_std_512x3x224x224": {"ms": round(t_ms * 1e3, 3), "GB_per_s": round(nb / t_ms / 1e9, 1)}}))
