"""Debug aid (GPU box): compare every intermediate of the fp32 engine with the oracle's taps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import train_oracle as O
from primia_amd import resnet_spec as rs
from primia_amd.engine import ResNet18Engine

size, batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 8
dtype = torch.float32
torch.manual_seed(42)
sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"))
eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=dtype, device="cuda:0")
eng.load_state_dict(sd)
g = torch.Generator().manual_seed(43)
x = torch.randn(batch, 3, size, size, generator=g)
y = torch.randint(0, 3, (batch,), generator=g)
taps = {}
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
keys = O.param_keys(sd64)
for k in keys:
    sd64[k].requires_grad_(True)
logits = O.forward(sd64, x.double(), True, "max", size, taps)
loss = torch.nn.functional.cross_entropy(logits, y)
loss.backward()

def rel(a, b):
    return ((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30)).item()

def nchw(t, hw):
    C = t.shape[1]
    return t.view(batch, hw, hw, C).permute(0, 3, 1, 2)

res = []
for rep in range(2):
    eng.forward(x.cuda())
    eng.loss_backward(y.cuda())
    torch.cuda.synchronize()
    res.append({k: v.clone() for k, v in eng.gviews.items()})
print("deterministic grads:", all(torch.equal(res[0][k], res[1][k]) for k in res[0]))
for k in res[0]:
    if not torch.equal(res[0][k], res[1][k]):
        print("  nondet", k, rel(res[0][k], res[1][k].double().cpu()))
        break
print("logits", rel(eng.logits, logits.detach()))
for name, t in taps.items():
    if name not in eng.t:
        continue
    hw = t.shape[-1]
    print(f"fwd {name:16s} {rel(nchw(eng.t[name], hw), t.detach()):.2e}")
# gradient taps
gmap = {"stem.y": "stem.dy", "pool.out": "pool.dout"}
for blk in eng.spec.blocks:
    p = blk.prefix
    gmap[p + ".y1"] = p + ".dy1"; gmap[p + ".y2"] = p + ".dy2"; gmap[p + ".a1"] = p + ".da1"
for name, t in taps.items():
    if name in gmap and t.grad is not None:
        hw = t.shape[-1]
        print(f"bwd d{name:16s} {rel(nchw(eng.t[gmap[name]], hw), t.grad):.2e}")
for k in keys:
    print(f"grad {k:34s} {rel(eng.gviews[k], sd64[k].grad):.2e}")

# ---- finer: snapshot the gradient entering each block (dz of bn2's backward) -------------------
snaps = {}
orig = eng._bn_bwd
def hooked(conv_name, y, z, dz, dy, g_out, relu):
    snaps[conv_name + ".dz_in"] = dz.clone()
    orig(conv_name, y, z, dz, dy, g_out, relu)
    snaps[conv_name + ".dy_out"] = dy.clone()
    if g_out is not None:
        snaps[conv_name + ".g_out"] = g_out.clone()
eng._bn_bwd = hooked
eng.forward(x.cuda()); eng.loss_backward(y.cuda()); torch.cuda.synchronize()
for blk in eng.spec.blocks:
    p = blk.prefix
    t = taps[p + ".out"]
    hw = t.shape[-1]
    print(f"dout-in {p:10s} {rel(nchw(snaps[blk.conv2.name + '.dz_in'], hw), t.grad):.2e}")

from primia_amd.resnet_spec import bn_name
for blk in eng.spec.blocks:
    cn = blk.conv2.name
    p = blk.prefix
    b = bn_name(cn)
    y2 = eng.t[p + ".y2"].double().cpu(); out = eng.t[p + ".out"].double().cpu()
    dz = snaps[cn + ".dz_in"].double().cpu()
    mean, invstd = [s.double().cpu() for s in eng.save[b]]
    gamma = eng.views[b + ".weight"].double().cpu()
    g = dz * (out > 0)
    xh = (y2 - mean) * invstd
    dy = gamma * invstd * (g - g.mean(0) - xh * (g * xh).mean(0))
    got = snaps[cn + ".dy_out"].double().cpu()
    # statistics actually consistent with y2?
    m2 = y2.mean(0); v2 = y2.var(0, unbiased=False)
    print(f"bn2bwd {p:10s} kernel-vs-recomputed {((got-dy).norm()/dy.norm()).item():.2e}  "
          f"mean err {((mean-m2).norm()/m2.norm()).item():.2e} invstd err {((invstd-1/(v2+1e-5).sqrt()).norm()/invstd.norm()).item():.2e} "
          f"g_out err {((snaps[cn+'.g_out'].double().cpu()-g).norm()/g.norm()).item():.2e}")
