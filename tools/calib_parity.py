"""Calibration for the end-to-end parity bars (prints measurements; tests/test_gpu_train_step.py holds the bars)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import train_oracle as O  # noqa: E402
from primia_amd import resnet_spec as rs  # noqa: E402
from primia_amd.engine import ResNet18Engine  # noqa: E402

cuda = torch.device("cuda:0")


def nchw(t, N, hw, C):
    return t.view(N, hw, hw, C).permute(0, 3, 1, 2).contiguous()


def engine_masks(eng, N):
    m = {}
    for name, v in eng.t.items():
        if name == "stem.z" or name.endswith(".a1") or name.endswith(".out"):
            if name == "pool.out":
                continue
            C = v.shape[1]
            hw = int(round((v.shape[0] // N) ** 0.5))
            m[name] = nchw(v.float().cpu(), N, hw, C) > 0
    return m


def fp32_case(batch, size, seed=42):
    torch.manual_seed(seed)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"))
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, 3, (batch,), generator=g)
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda)
    eng.fuse_stem = False
    eng.load_state_dict(sd)
    logits = eng.forward(x.to(cuda))
    eng.loss_backward(y.to(cuda))
    masks = engine_masks(eng, batch)
    for use in (False, True):
        osd = {k: v.clone() for k, v in sd.items()}
        for k in O.param_keys(osd):
            osd[k].requires_grad_(True)
        taps = {}
        ol = O.forward(osd, x, True, "max", size, taps=taps, relu_masks=masks if use else None)
        torch.nn.functional.cross_entropy(ol, y).backward()
        if not use:
            flips = {n: int((masks[n] != (taps[n].detach() > 0)).sum()) for n in masks}
            print("flips", {k: v for k, v in flips.items() if v}, "of", sum(m.numel() for m in masks.values()))
        worst = 0.0
        for k, _ in eng.p_entries:
            a, b = eng.gviews[k].double().cpu().flatten(), osd[k].grad.double().flatten()
            e = ((a - b).norm() / b.norm()).item()
            worst = max(worst, e)
        print(f"fp32 b{batch} s{size} masks={use}: logits {((logits.cpu() - ol.detach()).norm() / ol.detach().norm()).item():.2e} worst grad rel {worst:.2e}")


def bf16_full():
    B, S = 256, 224
    torch.manual_seed(42)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, S, "max"))
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, 3, S, S, generator=g)
    y = torch.randint(0, 3, (B,), generator=g)
    eng = ResNet18Engine(B, 3, 3, S, "max", dtype=torch.bfloat16, device=cuda)
    eng.load_state_dict(sd)
    logits = eng.forward(x.to(cuda)).float().cpu()
    loss = eng.loss_backward(y.to(cuda)).item()
    t = time.time()
    ologits, oloss, ograds = O.train_step(sd, x, y, 0.0, 0.0)
    print(f"oracle step at batch 256: {time.time() - t:.1f} s")
    print(f"bf16 full: logits rel {((logits - ologits).norm() / ologits.norm()).item():.3e} loss {loss:.6f} vs {oloss.item():.6f}")
    errs = {}
    for k, _ in eng.p_entries:
        a, b = eng.gviews[k].double().cpu().flatten(), ograds[k].double().flatten()
        errs[k] = ((a - b).norm() / b.norm()).item()
    for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:12]:
        print(f"   {k}: {v:.3e}")
    print("   median", sorted(errs.values())[len(errs) // 2])


if __name__ == "__main__":
    fp32_case(4, 224)
    fp32_case(8, 64)
    fp32_case(8, 64, seed=7)
    bf16_full()
