import torch, time, sys
sys.path.insert(0, "/root/repo")
from primia_amd import _lib
from primia_amd._lib import call, query
cuda = torch.device("cuda:0")
N, H, C = 256, 112, 64
M = N * H * H
y = torch.randn(M, C, device=cuda).to(torch.bfloat16)
Ho = 56
pooled = torch.empty(N * Ho * Ho, C, dtype=torch.bfloat16, device=cuda)
arg = torch.empty(N * Ho * Ho, C, dtype=torch.uint8, device=cuda)
g, b = torch.ones(C, device=cuda), torch.zeros(C, device=cuda)
rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
sm, si = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
wsb = query("primia_bn_workspace_bytes", M, C)
ws = torch.zeros(wsb, dtype=torch.uint8, device=cuda)
dt = _lib.dtype_code(torch.bfloat16)
import inspect
def run():
    call("primia_bn_relu_maxpool_fwd", y, pooled, arg, g, b, rm, rv, sm, si, N, H, H, C, 1e-5, 0.1, ws, wsb, dt)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize()
print("bn+relu+pool fwd (stats + finalize + pool):", (time.perf_counter() - t0) / 20 * 1e6, "us")
