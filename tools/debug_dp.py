import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from oracle import train_oracle as O
from primia_amd import resnet_spec as rs
from primia_amd.engine import ResNet18Engine
import torch.nn.functional as F
batch, size = 4, 64
torch.manual_seed(9)
sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"), "group")
eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device="cuda:0", norm="group")
eng.load_state_dict(sd)
g = torch.Generator().manual_seed(10)
x = torch.randn(batch, 3, size, size, generator=g); y = torch.randint(0, 3, (batch,), generator=g)
eng.dp_trace = OrderedDict()
eng.forward(x.cuda()); eng.dp_loss_backward(y.cuda(), 1.0, 0.0, noise=torch.zeros(eng.P, device="cuda"))
keys = O.param_keys(sd)
per = []
for n in range(batch):
    s2 = {k: v.clone() for k, v in sd.items()}
    for k in keys: s2[k].requires_grad_(True)
    F.cross_entropy(O.forward(s2, x[n:n+1], True, "max", size), y[n:n+1]).backward()
    per.append({k: s2[k].grad.double().pow(2).sum().item() for k in keys})
prev = torch.zeros(batch, dtype=torch.float64)
from primia_amd.resnet_spec import bn_name
for name, cum in eng.dp_trace.items():
    d = (cum.cpu() - prev); prev = cum.cpu()
    if name == "fc": want = [p["fc.weight"] + p["fc.bias"] for p in per]
    elif name + ".weight" in per[0] and name + ".bias" in per[0]: want = [p[name + ".weight"] + p[name + ".bias"] for p in per]
    else: want = [p[name + ".weight"] for p in per]
    r = [d[i].item() / max(want[i], 1e-30) for i in range(batch)]
    print(f"{name:32s} ratio {r[0]:.4f} {r[1]:.4f} {r[2]:.4f} {r[3]:.4f}")
