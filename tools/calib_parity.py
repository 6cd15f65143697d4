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
        if use:
            # the same masked network in float64: how far is each fp32 implementation from the exact gradient?
            dsd = {k: v.clone().double() for k, v in sd.items()}
            for k in O.param_keys(dsd):
                dsd[k].requires_grad_(True)
            dl = O.forward(dsd, x.double(), True, "max", size, relu_masks=masks)
            torch.nn.functional.cross_entropy(dl, y).backward()
            we = wo = 0.0
            for k, _ in eng.p_entries:
                t = dsd[k].grad.flatten()
                we = max(we, ((eng.gviews[k].double().cpu().flatten() - t).norm() / t.norm()).item())
                wo = max(wo, ((osd[k].grad.double().flatten() - t).norm() / t.norm()).item())
            print(f"   vs float64 gradient: engine {we:.2e}, torch-CPU fp32 oracle {wo:.2e}")
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
    # the oracle with the engine's storage format (bf16 buffers) on the same batch
    osd = {k: v.clone() for k, v in sd.items()}
    for k in O.param_keys(osd):
        osd[k].requires_grad_(True)
    ol = O.forward(osd, x, True, "max", S, bf16_storage=True)
    torch.nn.functional.cross_entropy(ol, y).backward()
    print(f"vs bf16-storage oracle: logits rel {((logits - ol.detach()).norm() / ol.detach().norm()).item():.3e}")
    e2 = sorted(((eng.gviews[k].double().cpu().flatten() - osd[k].grad.double().flatten()).norm() / osd[k].grad.double().norm()).item()
                for k, _ in eng.p_entries)
    print("   grad rel: median", e2[len(e2) // 2], "max", e2[-1])
    e3 = sorted(((ograds[k].double().flatten() - osd[k].grad.double().flatten()).norm() / ograds[k].double().norm()).item()
                for k, _ in eng.p_entries)
    print("   fp32 oracle vs bf16-storage oracle: median", e3[len(e3) // 2], "max", e3[-1])
    # per-layer, full size: every conv's forward output and weight gradient from the ENGINE'S OWN operands
    import torch.nn.functional as F
    N = B
    def nchw_(t, hw):
        return t.float().cpu().view(N, hw, hw, -1).permute(0, 3, 1, 2).contiguous()
    for blk in eng.spec.blocks[:3] + eng.spec.blocks[-1:]:
        p = blk.prefix
        for c, xin, yout, dyn in ((blk.conv1, None, "y1", "dy1"), (blk.conv2, "a1", "y2", "dy2")):
            d = eng.convs[c.name].desc
            if xin is None:
                i = eng.spec.blocks.index(blk)
                xin_t = eng.t["pool.out"] if i == 0 else eng.t[eng.spec.blocks[i - 1].prefix + ".out"]
            else:
                xin_t = eng.t[f"{p}.{xin}"]
            xc, yc, dyc = nchw_(xin_t, d.H), nchw_(eng.t[f"{p}.{yout}"], d.Ho), nchw_(eng.t[f"{p}.{dyn}"], d.Ho)
            w = sd[c.name + ".weight"].bfloat16().float()
            yref = F.conv2d(xc, w, None, d.stride, d.pad)
            gref = torch.nn.grad.conv2d_weight(xc, w.shape, dyc, d.stride, d.pad)
            g = eng.gviews[c.name + ".weight"].float().cpu()
            print(f"   {c.name}: fwd rel {((yc - yref).norm() / yref.norm()).item():.2e}  wgrad rel {((g - gref).norm() / gref.norm()).item():.2e}")


if __name__ == "__main__":
    fp32_case(4, 224)
    fp32_case(8, 64)
    fp32_case(8, 64, seed=7)
    bf16_full()
