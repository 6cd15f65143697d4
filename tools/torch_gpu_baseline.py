#!/usr/bin/env python3
"""One-off comparison point (not part of bench.py's contract): the reference's training step
executed by stock PyTorch-ROCm (MIOpen / hipBLASLt kernels) on the same MI355X, i.e. what a user
gets by running the reference's model with `--cuda` today.  Uses the oracle's functional ResNet-18
(tools/ may import oracle/ only for measurements like this; the product never does).

    python tools/torch_gpu_baseline.py [--batch 256] [--steps 20]

Prints one JSON line per variant: fp32 NCHW (the reference as written) and bf16 autocast
channels-last (the best stock configuration).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.nn.functional as F

from oracle import train_oracle as O
from primia_amd.resnet_spec import init_state_dict, resnet18_spec


def run(batch, steps, variant):
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    sd = {k: v.to(dev) for k, v in init_state_dict(resnet18_spec(3, 3, 224, "max")).items()}
    keys = O.param_keys(sd)
    x = torch.randn(batch, 3, 224, 224, device=dev)
    t = torch.randint(0, 3, (batch,), device=dev)
    if variant == "bf16_channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
        for k in keys:
            if sd[k].dim() == 4:
                sd[k] = sd[k].contiguous(memory_format=torch.channels_last)

    def step():
        for k in keys:
            sd[k].requires_grad_(True)
            sd[k].grad = None
        if variant == "bf16_channels_last":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                logits = O.forward(sd, x, True, "max", 224)
            loss = F.cross_entropy(logits.float(), t)
        else:
            loss = F.cross_entropy(O.forward(sd, x, True, "max", 224), t)
        loss.backward()
        with torch.no_grad():
            for k in keys:
                sd[k].add_(sd[k].grad, alpha=-1e-4)
        for k in keys:
            sd[k].requires_grad_(False)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"baseline": "stock PyTorch-ROCm " + torch.__version__, "variant": variant, "batch": batch,
                      "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(batch / dt, 1)}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    for v in ("fp32_nchw", "bf16_channels_last"):
        run(a.batch, a.steps, v)
