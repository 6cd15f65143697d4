"""Times the stem's backward tail at BASELINE configs[1] (batch 256, 224 x 224, bf16): the two-call chain
(primia_bn_relu_maxpool_bwd -> primia_stem_conv_wgrad_ws) against primia_bn_relu_maxpool_bwd(dy = NULL) ->
primia_stem_bwd_fused, and checks that the weight gradients have the same bits.  `python tools/stem_bwd_bench.py [N]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from primia_amd import _lib  # noqa: E402
from primia_amd._lib import ConvDesc, call, query  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    S, dev, dtype = 224, "cuda:0", torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 3, S, S, generator=g).to(dev)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.05).to(dev)
    desc = ConvDesc.make(N, S, S, 4, 64, 7, 7, 2, 3)
    wf = torch.empty(query("primia_conv_wfwd_elems", desc), dtype=dtype, device=dev)
    call("primia_conv_weight_prepare", desc, 3, w, wf, None, dt)
    xp = torch.zeros(N * (S + 6) * (S + 8), 4, dtype=dtype, device=dev)
    call("primia_nchw_to_nhwc_padded", x, xp, N, 3, S, S, 4, 3, 3, S + 6, S + 8, dt)
    Ho, Hq = S // 2, S // 4
    M = N * Ho * Ho
    y = torch.empty(M, 64, dtype=dtype, device=dev)
    call("primia_stem_conv_fwd", xp, wf, y, N, S, S, dt)
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    ws_bytes = query("primia_bn_workspace_bytes", M, 64)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    sm, si = torch.empty(64, device=dev), torch.empty(64, device=dev)
    p = torch.empty(N * Hq * Hq, 64, dtype=dtype, device=dev)
    am = torch.empty(N * Hq * Hq, 64, dtype=torch.uint8, device=dev)
    call("primia_bn_relu_maxpool_fwd", y, p, am, gamma, beta, rm, rv, sm, si, N, Ho, Ho, 64, 1e-5, 0.1, ws, ws_bytes, dt)
    dp = torch.randn(N * Hq * Hq, 64, generator=g).to(dtype).to(dev)
    wws_bytes = query("primia_stem_conv_wgrad_ws_bytes", N, S, S)
    wws = torch.empty(wws_bytes // 4, device=dev)
    n = query("primia_conv_wfwd_elems", desc)
    dy = torch.empty_like(y)
    dg, db = torch.empty(64, device=dev), torch.empty(64, device=dev)
    a0, a1 = torch.zeros(n, device=dev), torch.zeros(n, device=dev)

    def chain_a():
        call("primia_bn_relu_maxpool_bwd", y, p, dp, am, dy, gamma, beta, sm, si, dg, db, N, Ho, Ho, 64, ws, ws_bytes, dt)

    def chain_b():
        call("primia_stem_conv_wgrad_ws", xp, dy, a0, wws, wws_bytes, N, S, S, dt)

    def fused_a():
        call("primia_bn_relu_maxpool_bwd", y, p, dp, am, None, gamma, beta, sm, si, dg, db, N, Ho, Ho, 64, ws, ws_bytes, dt)

    def fused_b():
        call("primia_stem_bwd_fused", xp, y, dp, am, gamma, beta, sm, si, dg, db, a1, wws, wws_bytes, N, S, S, dt)

    def timeit(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    ta, tb = timeit(chain_a), timeit(chain_b)
    fa, fb = timeit(fused_a), timeit(fused_b)
    print(f"N={N}: chain  sums+apply {ta:7.1f} us + wgrad {tb:7.1f} us = {ta + tb:7.1f} us")
    print(f"N={N}: fused  sums       {fa:7.1f} us + wgrad {fb:7.1f} us = {fa + fb:7.1f} us   "
          f"(bit-identical: {bool(torch.equal(a0, a1))})")


if __name__ == "__main__":
    main()
