import torch, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from primia_amd import _lib
from primia_amd._lib import call, query
cuda = torch.device("cuda:0")
dt = _lib.dtype_code(torch.bfloat16)
for (N, H, C) in [(256, 56, 64), (256, 28, 128)]:
    M = N * H * H
    y = torch.randn(M, C, device=cuda).to(torch.bfloat16)
    dz = torch.randn(M, C, device=cuda).to(torch.bfloat16)
    dy = torch.empty_like(y)
    g, b = torch.ones(C, device=cuda), torch.zeros(C, device=cuda)
    sm, si = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    dg, db = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    wsb = query("primia_bn_workspace_bytes", M, C)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=cuda)
    def run():
        call("primia_bn_relu_bwd", y, dz, dy, g, b, sm, si, dg, db, M, C, ws, wsb, dt)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"bn_relu_bwd N={N} H={H} C={C}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
