"""Encrypted-inference timing (BASELINE.json configs[4], all three roles on one GPU):
ms/image for the online phase (primitives pre-provisioned) and for the dealer (triples + FSS keys).
    python tools/bench_secure.py [--size 224] [--pf 16] [--images 2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from primia_amd import resnet_spec as rs
from primia_amd.secure import Dealer, PreloadedDealer, SecureContext, SecureResNet18

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--pf", type=int, default=16)
ap.add_argument("--images", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(42)
sd = rs.init_state_dict(rs.resnet18_spec(3, 3, a.size, "max"))
g = torch.Generator().manual_seed(1)
img = torch.randn(1, 3, a.size, a.size, generator=g).to(dev)

def run(dealer, share_model=True):
    ctx = SecureContext(dealer, 10, a.pf)
    model = SecureResNet18(ctx, sd, input_size=a.size)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model(img)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out, ctx

# warm-up (JIT-free, but first launches / allocator)
run(Dealer(dev, seed=0))
res = []
for i in range(a.images):
    d = Dealer(dev, seed=100 + i); d.tape = []
    t_total, out_a, ctx = run(d)
    t_online, out_b, _ = run(PreloadedDealer(d.tape, dev))
    assert torch.equal(out_a, out_b), "replayed run must be bit-identical"
    res.append((t_total, t_online))
    del d
tt = sum(r[0] for r in res) / len(res); to = sum(r[1] for r in res) / len(res)
print(json.dumps({"metric": "encrypted_inference_ms_per_image", "online_ms": round(to * 1e3, 1),
                  "with_dealer_ms": round(tt * 1e3, 1), "dealer_ms": round((tt - to) * 1e3, 1),
                  "precision_fractional": a.pf, "size": a.size, "dif_evals": ctx.stats["dif_evals"],
                  "beaver_matmul": ctx.stats["beaver_matmul"], "beaver_mul": ctx.stats["beaver_mul"],
                  "topology": "party0 + party1 + dealer on one MI355X (LocalOpener)"}))
