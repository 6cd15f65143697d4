"""GPU parity tests, one per training kernel, called through the C ABI.

Oracle for each op = the native torch-CPU fp32 op the reference's workers execute
(SURVEY.md §8a T2-T8).  Tolerances:
  fp32 kernels : 1e-5 relative (north_star), measured as ||a-b|| / ||b|| over the tensor;
  bf16 kernels : inputs are rounded to bf16 first and the fp32 oracle runs on the ROUNDED values,
                 so the only differences are accumulation order and the final bf16 store
                 (<= 2^-8 relative per element) — bound 1e-2 normwise, stated per test.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from primia_amd import _lib  # noqa: E402
from primia_amd._lib import ConvDesc, call, query  # noqa: E402

DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 1e-5 if dtype == torch.float32 else 1e-2


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def rnd(x, dtype):
    """Round to the compute dtype and come back to fp32 (what the kernel will see)."""
    return x.to(dtype).to(torch.float32)


def to_nhwc(x, dtype, dev, c_pad=None):
    """CPU NCHW fp32 -> device [N*H*W, C] in dtype (test-side conversion)."""
    N, C, H, W = x.shape
    y = x.permute(0, 2, 3, 1).contiguous()
    if c_pad and c_pad > C:
        y = F.pad(y, (0, c_pad - C))
    return y.reshape(N * H * W, -1).to(dtype).to(dev)


def from_nhwc(y, N, H, W):
    C = y.shape[1]
    return y.to(torch.float32).cpu().view(N, H, W, C).permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # N, H, C, K, R, stride, pad
    (2, 16, 64, 64, 3, 1, 1),
    (2, 16, 64, 128, 3, 2, 1),
    (2, 16, 64, 128, 1, 2, 0),
    (1, 10, 128, 128, 3, 1, 1),   # M = 100: ragged pixel tile
    (3, 7, 256, 512, 3, 2, 1),    # odd size, stride 2
    (2, 8, 512, 512, 3, 1, 1),
    (4, 14, 256, 512, 3, 2, 1),   # layer4.0.conv1 at batch 4 (ragged: 784 / 196 pixels)
    (4, 14, 256, 512, 1, 2, 0),   # layer4.0.downsample
    (4, 28, 128, 256, 3, 2, 1),   # layer3.0.conv1
    (4, 7, 512, 512, 3, 1, 1),    # layer4.1
    (3, 14, 256, 256, 3, 1, 1),   # layer3.1 (16-wide sub-patches in the halo-patch wgrad)
    (2, 28, 128, 128, 3, 1, 1),   # layer2.1 (ragged right edge: 28 = 3.5 x 8)
    (1, 56, 64, 64, 3, 1, 1),     # layer1 (exact 4x8 tiling)
    (3, 12, 64, 64, 3, 1, 1),     # layer1 shape, ragged 8x8 patches (12 = 1.5 x 8), odd patch count
    (5, 8, 64, 64, 3, 1, 1),      # one patch per image, odd total
    # linear-halo kernel (conv3x3_lh.hip) corner cases
    (1, 30, 64, 128, 3, 1, 1),    # widest image it serves (W = 30), ONE 64-channel chunk (9 steps in all)
    (1, 31, 64, 128, 3, 1, 1),    # W = 31: falls back to the implicit GEMM
    (3, 5, 128, 128, 3, 1, 1),    # M = 75: a single ragged 256-pixel tile, images narrower than the halo shift table
    (7, 9, 192, 256, 3, 1, 1),    # 3 chunks, 2 channel tiles, ragged last pixel tile (M = 567)
    (2, 2, 128, 128, 3, 1, 1),    # 2x2 images: every tap but the centre leaves the image somewhere
]


def prep_weights(desc, w, dtype, dev, c_real, need_dgrad=True):
    wf = torch.empty(query("primia_conv_wfwd_elems", desc), dtype=dtype, device=dev)
    wd = torch.empty(query("primia_conv_wdgrad_elems", desc), dtype=dtype, device=dev) if need_dgrad else None
    call("primia_conv_weight_prepare", desc, c_real, w.to(dev), wf, wd, _lib.dtype_code(dtype))
    return wf, wd


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(cuda, dtype, case):
    N, H, C, K, R, s, p = case
    g = torch.Generator().manual_seed(1234 + H + C)
    x = rnd(torch.randn(N, C, H, H, generator=g), dtype)
    w = rnd(torch.randn(K, C, R, R, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, H, H, C, K, R, R, s, p)
    dt = _lib.dtype_code(dtype)
    wf, wd = prep_weights(desc, w, dtype, cuda, C)

    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, s, p)
    dy = rnd(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(dy)

    xd = to_nhwc(x, dtype, cuda)
    y = torch.empty(N * desc.Ho * desc.Wo, K, dtype=dtype, device=cuda)
    call("primia_conv2d_fwd", desc, xd, wf, y, dt)
    assert relerr(from_nhwc(y, N, desc.Ho, desc.Wo), y_ref.detach()) < tol(dtype)

    dyd = to_nhwc(dy, dtype, cuda)
    dx = torch.empty(N * H * H, C, dtype=dtype, device=cuda)
    call("primia_conv2d_dgrad", desc, dyd, wd, dx, 0, dt)
    assert relerr(from_nhwc(dx, N, H, H), xr.grad) < tol(dtype)
    # accumulate form: dx2 = base + dgrad
    base = rnd(torch.randn(N, C, H, H, generator=g), dtype)
    dx2 = to_nhwc(base, dtype, cuda)
    call("primia_conv2d_dgrad", desc, dyd, wd, dx2, 1, dt)
    assert relerr(from_nhwc(dx2, N, H, H), xr.grad + base) < tol(dtype)

    acc = torch.zeros(query("primia_conv_wfwd_elems", desc), dtype=torch.float32, device=cuda)
    call("primia_conv2d_wgrad", desc, xd, dyd, acc, dt)
    dw = torch.empty(K, C, R, R, dtype=torch.float32, device=cuda)
    call("primia_conv_wgrad_finalize", desc, C, acc, dw)
    # wgrad accumulates in fp32 in both modes (inputs already rounded) -> fp32-class tolerance
    assert relerr(dw, wr.grad) < (1e-5 if dtype == torch.float32 else 1e-4)

    # atomic-free path (primia_conv2d_wgrad_ws): same gradient, OVERWRITES a dirty dw_acc, bit-identical from run
    # to run; a missing / short workspace falls back to the accumulate path
    need = query("primia_conv_wgrad_ws_bytes", desc, dt)
    assert need > 0
    served = True
    ws = torch.empty(max(need, 16) // 4, dtype=torch.float32, device=cuda)
    runs = []
    for _ in range(2):
        acc2 = torch.full_like(acc, 7.0)
        call("primia_conv2d_wgrad_ws", desc, xd, dyd, acc2, ws, need, dt)
        dw2 = torch.empty_like(dw)
        call("primia_conv_wgrad_finalize", desc, C, acc2, dw2)
        runs.append(dw2)
    assert relerr(runs[0], wr.grad) < (1e-5 if dtype == torch.float32 else 1e-4)
    if served:
        assert torch.equal(runs[0], runs[1])
    acc3 = torch.zeros_like(acc)
    call("primia_conv2d_wgrad_ws", desc, xd, dyd, acc3, None, 0, dt)
    dw3 = torch.empty_like(dw)
    call("primia_conv_wgrad_finalize", desc, C, acc3, dw3)
    assert relerr(dw3, wr.grad) < (1e-5 if dtype == torch.float32 else 1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H", [(2, 32), (1, 18)])
def test_stem_conv(cuda, dtype, N, H):
    g = torch.Generator().manual_seed(99)
    x = rnd(torch.randn(N, 3, H, H, generator=g), dtype)
    w = rnd(torch.randn(64, 3, 7, 7, generator=g) * 0.1, dtype)
    desc = ConvDesc.make(N, H, H, 4, 64, 7, 7, 2, 3)
    dt = _lib.dtype_code(dtype)
    wf, _ = prep_weights(desc, w, dtype, cuda, 3, need_dgrad=False)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(x, wr, None, 2, 3)
    dy = rnd(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(dy)

    # boundary conversion kernel: NCHW fp32 -> NHWC (3 -> 4 channels)
    xd = torch.empty(N * H * H, 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc", x.to(cuda), xd, N, 3, H, H, 4, dt)
    assert torch.equal(xd.cpu(), to_nhwc(x, dtype, "cpu", c_pad=4))
    back = torch.empty(N, 3, H, H, dtype=torch.float32, device=cuda)
    call("primia_nhwc_to_nchw", xd, back, N, 3, H, H, 4, dt)
    assert torch.equal(back.cpu(), x)

    y = torch.empty(N * desc.Ho * desc.Wo, 64, dtype=dtype, device=cuda)
    call("primia_conv2d_fwd", desc, xd, wf, y, dt)
    assert relerr(from_nhwc(y, N, desc.Ho, desc.Wo), y_ref.detach()) < tol(dtype)

    acc = torch.zeros(query("primia_conv_wfwd_elems", desc), dtype=torch.float32, device=cuda)
    call("primia_conv2d_wgrad", desc, xd, to_nhwc(dy, dtype, cuda), acc, dt)
    dw = torch.empty(64, 3, 7, 7, dtype=torch.float32, device=cuda)
    call("primia_conv_wgrad_finalize", desc, 3, acc, dw)
    assert relerr(dw, wr.grad) < (1e-5 if dtype == torch.float32 else 1e-4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,relu,res,N,H", [(64, 1, 0, 3, 9), (128, 1, 1, 3, 9), (512, 0, 0, 3, 9),
                                               (256, 1, 1, 3, 9), (256, 1, 1, 4, 14), (512, 1, 1, 4, 7),
                                               (64, 1, 0, 2, 56)])
def test_batchnorm_train(cuda, dtype, C, relu, res, N, H):
    M = N * H * H
    g = torch.Generator().manual_seed(5 + C)
    y = rnd(torch.randn(N, C, H, H, generator=g) * 2 + 0.5, dtype)
    r = rnd(torch.randn(N, C, H, H, generator=g), dtype) if res else None
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g)
    rm0, rv0 = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    dt = _lib.dtype_code(dtype)

    yr = y.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    rm, rv = rm0.clone(), rv0.clone()
    z_ref = F.batch_norm(yr, rm, rv, gr, br, True, 0.1, 1e-5)
    if res:
        z_ref = z_ref + rr
    if relu:
        z_ref = F.relu(z_ref)
    dz = rnd(torch.randn(z_ref.shape, generator=g), dtype)
    z_ref.backward(dz)

    ws_bytes = query("primia_bn_workspace_bytes", M, C)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    yd = to_nhwc(y, dtype, cuda)
    rd = to_nhwc(r, dtype, cuda) if res else None
    z = torch.empty_like(yd)
    d_rm, d_rv = rm0.to(cuda), rv0.to(cuda)
    sm = torch.empty(C, device=cuda)
    si = torch.empty(C, device=cuda)
    call("primia_bn_fwd_train", yd, rd, z, gamma.to(cuda), beta.to(cuda), d_rm, d_rv, sm, si, M, C, 1e-5, 0.1,
         relu, ws, ws_bytes, dt)
    assert relerr(from_nhwc(z, N, H, H), z_ref.detach()) < tol(dtype)
    assert relerr(d_rm, rm) < 1e-5 and relerr(d_rv, rv) < 1e-5

    # backward uses the kernel's own z for the ReLU mask
    dzd = to_nhwc(dz, dtype, cuda)
    dy = torch.empty_like(yd)
    dgam = torch.empty(C, device=cuda)
    dbet = torch.empty(C, device=cuda)
    g_out = torch.empty_like(yd) if res else None
    call("primia_bn_bwd", yd, z if relu else None, dzd, dy, g_out, gamma.to(cuda), sm, si, dgam, dbet, M, C, relu,
         ws, ws_bytes, dt)
    t = tol(dtype)
    assert relerr(from_nhwc(dy, N, H, H), yr.grad) < (2e-5 if dtype == torch.float32 else 2e-2)
    assert relerr(dgam, gr.grad) < (1e-5 if dtype == torch.float32 else 5e-3)
    assert relerr(dbet, br.grad) < (1e-5 if dtype == torch.float32 else 5e-3)
    if res:
        assert relerr(from_nhwc(g_out, N, H, H), rr.grad) < t
    if relu and not res:
        # recomputed-mask form: identical results without reading z
        dy2, dgam2, dbet2 = torch.empty_like(yd), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
        call("primia_bn_relu_bwd", yd, dzd, dy2, gamma.to(cuda), beta.to(cuda), sm, si, dgam2, dbet2, M, C, ws,
             ws_bytes, dt)
        assert torch.equal(dy2, dy) and torch.equal(dgam2, dgam) and torch.equal(dbet2, dbet)

    # eval mode
    z2 = torch.empty_like(yd)
    call("primia_bn_fwd_eval", yd, rd, z2, gamma.to(cuda), beta.to(cuda), d_rm, d_rv, M, C, 1e-5, relu, dt)
    ze = F.batch_norm(y, d_rm.cpu(), d_rv.cpu(), gamma, beta, False, 0.1, 1e-5)
    if res:
        ze = ze + r
    if relu:
        ze = F.relu(ze)
    assert relerr(from_nhwc(z2, N, H, H), ze) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,kind", [(2, 16, "max"), (1, 9, "max"), (2, 12, "avg")])
def test_pool3x3s2(cuda, dtype, N, H, kind):
    C = 64
    g = torch.Generator().manual_seed(3)
    # ReLU-like input with many exact ties at 0 (the stem feeds post-ReLU activations)
    x = rnd(F.relu(torch.randn(N, C, H, H, generator=g)), dtype)
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1) if kind == "max" else F.avg_pool2d(xr, 3, 2, 1)
    dy = rnd(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(dy)
    Ho = y_ref.shape[2]
    dt = _lib.dtype_code(dtype)
    xd = to_nhwc(x, dtype, cuda)
    y = torch.empty(N * Ho * Ho, C, dtype=dtype, device=cuda)
    dx = torch.empty_like(xd)
    if kind == "max":
        am = torch.empty(N * Ho * Ho * C, dtype=torch.uint8, device=cuda)
        call("primia_maxpool3x3s2_fwd", xd, y, am, N, H, H, C, dt)
        call("primia_maxpool3x3s2_bwd", to_nhwc(dy, dtype, cuda), am, dx, N, H, H, C, dt)
        assert torch.equal(from_nhwc(y, N, Ho, Ho), y_ref.detach())  # a max is exact
    else:
        call("primia_avgpool3x3s2_fwd", xd, y, N, H, H, C, dt)
        call("primia_avgpool3x3s2_bwd", to_nhwc(dy, dtype, cuda), dx, N, H, H, C, dt)
        assert relerr(from_nhwc(y, N, Ho, Ho), y_ref.detach()) < tol(dtype)
    assert relerr(from_nhwc(dx, N, H, H), xr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_head(cuda, dtype):
    N, HW, C, NC = 5, 4, 512, 3
    g = torch.Generator().manual_seed(11)
    x = rnd(torch.randn(N, C, 2, 2, generator=g), dtype)
    w = torch.randn(NC, C, generator=g) * 0.05
    b = torch.randn(NC, generator=g)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    feat_ref = torch.flatten(F.avg_pool2d(xr, 2), 1)
    logits_ref = F.linear(feat_ref, wr, br)
    dl = torch.randn(N, NC, generator=g)
    logits_ref.backward(dl)
    dt = _lib.dtype_code(dtype)
    xd = to_nhwc(x, dtype, cuda)
    feat = torch.empty(N, C, device=cuda)
    call("primia_global_avgpool_fwd", xd, feat, N, HW, C, dt)
    logits = torch.empty(N, NC, device=cuda)
    call("primia_linear_fwd", feat, w.to(cuda), b.to(cuda), logits, N, C, NC)
    assert relerr(logits, logits_ref.detach()) < 1e-5
    dfeat = torch.empty(N, C, device=cuda)
    dw = torch.empty(NC, C, device=cuda)
    db = torch.empty(NC, device=cuda)
    call("primia_linear_bwd", feat, w.to(cuda), dl.to(cuda), dfeat, dw, db, N, C, NC)
    assert relerr(dw, wr.grad) < 1e-5 and relerr(db, br.grad) < 1e-5
    dx = torch.empty_like(xd)
    call("primia_global_avgpool_bwd", dfeat, dx, N, HW, C, dt)
    assert relerr(from_nhwc(dx, N, 2, 2), xr.grad) < tol(dtype)
    # the head in two launches (what the engine calls): same feat / logits / dx
    feat2, logits2, dx2 = torch.empty_like(feat), torch.empty_like(logits), torch.empty_like(xd)
    call("primia_head_fwd", xd, w.to(cuda), b.to(cuda), feat2, logits2, N, HW, C, NC, dt)
    assert relerr(feat2, feat_ref.detach()) < 1e-5 and relerr(logits2, logits_ref.detach()) < 1e-5
    call("primia_head_bwd", w.to(cuda), dl.to(cuda), dx2, N, HW, C, NC, dt)
    assert relerr(from_nhwc(dx2, N, 2, 2), xr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_head_fused_resnet_shape(cuda, dtype):
    """AvgPool2d(7) -> Linear(512, 3) at the ResNet-18 shape (49 pixels: ragged over the 4 / 2 row groups)."""
    N, C, NC = 3, 512, 3
    g = torch.Generator().manual_seed(12)
    x = rnd(torch.randn(N, C, 7, 7, generator=g), dtype)
    w = torch.randn(NC, C, generator=g) * 0.05
    b = torch.randn(NC, generator=g)
    xr = x.clone().requires_grad_(True)
    feat_ref = torch.flatten(F.avg_pool2d(xr, 7), 1)
    logits_ref = F.linear(feat_ref, w, b)
    dl = torch.randn(N, NC, generator=g)
    logits_ref.backward(dl)
    dt = _lib.dtype_code(dtype)
    xd = to_nhwc(x, dtype, cuda)
    feat, logits, dx = torch.empty(N, C, device=cuda), torch.empty(N, NC, device=cuda), torch.empty_like(xd)
    call("primia_head_fwd", xd, w.to(cuda), b.to(cuda), feat, logits, N, 49, C, NC, dt)
    assert relerr(feat, feat_ref.detach()) < 1e-5 and relerr(logits, logits_ref.detach()) < 1e-5
    call("primia_head_bwd", w.to(cuda), dl.to(cuda), dx, N, 49, C, NC, dt)
    assert relerr(from_nhwc(dx, N, 7, 7), xr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_head_backward_with_bn_backward_sums(cuda, dtype):
    """primia_head_bwd_bnsums: the head's backward pass also forms the backward sums of the last block's residual BatchNorm, +
    primia_bn_bwd_mask_from_sums, against primia_head_bwd followed by primia_bn_bwd_mask with its own reduction pass: dx
    bit-identical, dgamma / dbeta to summation order, dy to one rounding."""
    N, C, NC, HW = 5, 512, 3, 49
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(31)
    w = (torch.randn(NC, C, generator=g) * 0.05).to(cuda)
    dl = torch.randn(N, NC, generator=g).to(cuda)
    M = N * HW
    ch = 4 if dtype == torch.float32 else 8
    y = (torch.randn(M, C, generator=g) * 1.2 - 0.1).to(dtype).to(cuda)
    mask = torch.randint(0, 1 << ch, (M * C // ch,), generator=g, dtype=torch.int32).to(torch.uint8).to(cuda)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda)
    mean = y.float().mean(0)
    invstd = 1.0 / torch.sqrt(y.float().var(0, unbiased=False) + 1e-5)
    dx_a = torch.empty(M, C, dtype=dtype, device=cuda)
    call("primia_head_bwd", w, dl, dx_a, N, HW, C, NC, dt)
    ws = torch.zeros(query("primia_bn_workspace_bytes", M, C), dtype=torch.uint8, device=cuda)
    dy_a, dg_a, db_a = torch.empty_like(dx_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_bwd_mask", y, mask, dx_a, dy_a, None, gamma, mean, invstd, dg_a, db_a, M, C, ws, ws.numel(), dt)
    dx_b = torch.empty_like(dx_a)
    sums = torch.full((N, 2, C), 5.0, device=cuda)
    call("primia_head_bwd_bnsums", w, dl, dx_b, y, mask, mean, invstd, sums, N, HW, C, NC, dt)
    dy_b, dg_b, db_b = torch.empty_like(dx_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_bwd_mask_from_sums", y, mask, dx_b, dy_b, None, gamma, mean, invstd, dg_b, db_b, sums, N, M, C, dt)
    assert torch.equal(dx_a, dx_b)
    assert relerr(dg_b, dg_a) < 2e-5 and relerr(db_b, db_a) < 2e-5
    assert relerr(dy_b.float(), dy_a.float()) < (1e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("weighted", [False, True])
def test_xent(cuda, weighted):
    import sys, os
    from oracle import train_oracle as O

    N, C = 37, 3
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(N, C, generator=g) * 3
    cw = torch.tensor([0.5, 1.0, 2.0]) if weighted else None
    tgt = torch.randint(0, C, (N,), generator=g)
    lr = logits.clone().requires_grad_(True)
    loss_ref = F.cross_entropy(lr, tgt, weight=cw)
    loss_ref.backward()
    loss = torch.zeros(1, device=cuda)
    dl = torch.empty(N, C, device=cuda)
    call("primia_xent_hard", logits.to(cuda), tgt.to(cuda), cw.to(cuda) if weighted else None, loss, dl, N, C)
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    assert relerr(dl, lr.grad) < 1e-5
    soft = torch.rand(N, C, generator=g)
    soft = soft / soft.sum(1, keepdim=True)
    lr2 = logits.clone().requires_grad_(True)
    l2 = O.cross_entropy_one_hot(lr2, soft, cw)
    l2.backward()
    call("primia_xent_soft", logits.to(cuda), soft.to(cuda), cw.to(cuda) if weighted else None, loss, dl, N, C)
    assert abs(loss.item() - l2.item()) < 1e-5 * abs(l2.item())
    assert relerr(dl, lr2.grad) < 1e-5
    # the reference's own Cross_entropy_one_hot class, executed when the fixture was minted
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_schedule.npz"))
    o, t = torch.from_numpy(gold["ce.out"]), torch.from_numpy(gold["ce.target"])
    key = "w" if weighted else "nw"
    dl6 = torch.empty(6, C, device=cuda)
    call("primia_xent_soft", o.to(cuda), t.to(cuda), cw.to(cuda) if weighted else None, loss, dl6, 6, C)
    assert abs(loss.item() - gold[f"ce.{key}.mean.loss"].item()) < 1e-5 * abs(gold[f"ce.{key}.mean.loss"].item())
    assert relerr(dl6, torch.from_numpy(gold[f"ce.{key}.mean.grad"])) < 1e-5


def test_optimizers_and_fx(cuda):
    from oracle import train_oracle as O

    n = 100003  # not a multiple of 4: exercises the scalar tail
    g = torch.Generator().manual_seed(8)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p = p0.clone()
    O.sgd_step([p], [gr], 1e-2, 5e-4)
    pd = p0.to(cuda)
    call("primia_sgd_step", pd, gr.to(cuda), n, 1e-2, 5e-4)
    assert relerr(pd, p) < 1e-6
    # Adam, 3 steps
    p = p0.clone()
    st = {}
    pd, m, v = p0.to(cuda), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    for step in range(1, 4):
        gs = torch.randn(n, generator=g)
        O.adam_step([p], [gs], st, 1e-3, (0.5, 0.99), 1e-8, 5e-4)
        call("primia_adam_step", pd, gs.to(cuda), m, v, n, 1e-3, 0.5, 0.99, 1e-8, 5e-4, step)
    assert relerr(pd, p) < 1e-5
    # fixed-precision encode/decode (precision.py:117-144) at the literal 10^16 and at 10^3
    x = torch.cat([torch.randn(1000, generator=g) * 1e-3, torch.tensor([0.0, -0.5, 0.5, 1e-17, -123.456])])
    for pf in (16, 3):
        if pf == 16:
            xs = x.clamp(-900, 900)
        else:
            xs = x
        q_ref = O.fix_encode(xs, 10, pf)
        q = torch.empty(xs.numel(), dtype=torch.int64, device=cuda)
        call("primia_fx_encode", xs.to(cuda), q, xs.numel(), float(10 ** pf))
        assert torch.equal(q.cpu(), q_ref)
        back = torch.empty(xs.numel(), device=cuda)
        call("primia_fx_decode", q, back, xs.numel(), float(10 ** pf))
        assert torch.equal(back.cpu(), O.fix_decode(q_ref, 10, pf))
    y = torch.randn(n, generator=g)
    yd = y.to(cuda)
    call("primia_scale", yd, n, 0.125)
    assert torch.equal(yd.cpu(), y * 0.125)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 64, 64, 3, 1, 1), (1, 10, 128, 128, 3, 1, 1), (4, 14, 256, 512, 3, 2, 1)])
def test_conv_fwd_fused_bn_stats(cuda, dtype, case):
    """conv forward that also accumulates the BatchNorm batch sums, then BN from those sums ==
    conv followed by batch_norm(training) on the CPU."""
    N, H, C, K, R, s, p = case
    g = torch.Generator().manual_seed(77 + H)
    x = rnd(torch.randn(N, C, H, H, generator=g), dtype)
    w = rnd(torch.randn(K, C, R, R, generator=g) * 0.05, dtype)
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    desc = ConvDesc.make(N, H, H, C, K, R, R, s, p)
    dt = _lib.dtype_code(dtype)
    wf, _ = prep_weights(desc, w, dtype, cuda, C, need_dgrad=False)
    M = N * desc.Ho * desc.Wo
    y = torch.empty(M, K, dtype=dtype, device=cuda)
    slots = query("primia_conv_stat_slots_for", desc, dt)   # 64 atomic slots (fp32) or one partial per kernel tile
    part = torch.zeros(slots, 2, K, device=cuda)
    call("primia_conv2d_fwd_stats", desc, to_nhwc(x, dtype, cuda), wf, y, part, dt)
    sums = part.double().sum(0).reshape(-1)
    y_ref = F.conv2d(x, w, None, s, p)
    y_st = from_nhwc(y, N, desc.Ho, desc.Wo)          # statistics are taken over the STORED values
    assert relerr(y_st, y_ref) < tol(dtype)
    s1 = y_st.double().sum(dim=(0, 2, 3))
    s2 = (y_st.double() ** 2).sum(dim=(0, 2, 3))
    assert relerr(sums[:K], s1) < 2e-5 and relerr(sums[K:], s2) < 2e-5
    z = torch.empty_like(y)
    rm, rv = torch.zeros(K, device=cuda), torch.ones(K, device=cuda)
    sm, si = torch.empty(K, device=cuda), torch.empty(K, device=cuda)
    call("primia_bn_fwd_train_from_sums", y, None, z, gamma.to(cuda), beta.to(cuda), rm, rv, sm, si, part, slots, M, K,
         1e-5, 0.1, 1, dt)
    rm_ref, rv_ref = torch.zeros(K), torch.ones(K)
    z_ref = F.relu(F.batch_norm(y_st, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    assert relerr(from_nhwc(z, N, desc.Ho, desc.Wo), z_ref) < (5e-5 if dtype == torch.float32 else 1e-2)
    assert relerr(rm, rm_ref) < 1e-4 and relerr(rv, rv_ref) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,C", [(2, 16, 64), (1, 9, 64), (3, 14, 128)])
def test_bn_relu_maxpool_fused_matches_unfused(cuda, dtype, N, H, C):
    """The fused stem tail: forward bit-identical to bn_fwd_train -> maxpool; backward equal to
    maxpool_bwd -> bn_bwd up to rounding (see primia_bn_relu_maxpool_bwd), odd sizes included."""
    g = torch.Generator().manual_seed(77 + H)
    dt = _lib.dtype_code(dtype)
    M = N * H * H
    Ho = (H - 1) // 2 + 1
    yd = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g) * 2 + 0.3, dtype), dtype, cuda)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda)
    beta = torch.randn(C, generator=g).to(cuda)
    gamma[5], beta[5] = 0.0, 0.7     # a dead scale: z is constant, xhat must come from y (fallback path)
    ws_bytes = query("primia_bn_workspace_bytes", M, C)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    rm0, rv0 = torch.randn(C, generator=g).to(cuda), (torch.rand(C, generator=g) + 0.5).to(cuda)

    # unfused
    rm, rv = rm0.clone(), rv0.clone()
    sm, si = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    z = torch.empty_like(yd)
    call("primia_bn_fwd_train", yd, None, z, gamma, beta, rm, rv, sm, si, M, C, 1e-5, 0.1, 1, ws, ws_bytes, dt)
    p = torch.empty(N * Ho * Ho, C, dtype=dtype, device=cuda)
    am = torch.empty(N * Ho * Ho, C, dtype=torch.uint8, device=cuda)
    call("primia_maxpool3x3s2_fwd", z, p, am, N, H, H, C, dt)
    dp = to_nhwc(rnd(torch.randn(N, C, Ho, Ho, generator=g), dtype), dtype, cuda)
    dz = torch.empty_like(yd)
    call("primia_maxpool3x3s2_bwd", dp, am, dz, N, H, H, C, dt)
    dy, dg, db = torch.empty_like(yd), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_bwd", yd, z, dz, dy, None, gamma, sm, si, dg, db, M, C, 1, ws, ws_bytes, dt)

    # fused
    rm2, rv2 = rm0.clone(), rv0.clone()
    sm2, si2 = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    p2, am2 = torch.empty_like(p), torch.empty_like(am)
    call("primia_bn_relu_maxpool_fwd", yd, p2, am2, gamma, beta, rm2, rv2, sm2, si2, N, H, H, C, 1e-5, 0.1, ws,
         ws_bytes, dt)
    assert torch.equal(p2, p) and torch.equal(am2, am)
    assert torch.equal(sm2, sm) and torch.equal(si2, si) and torch.equal(rm2, rm) and torch.equal(rv2, rv)
    dy2, dg2, db2 = torch.empty_like(yd), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_relu_maxpool_bwd", yd, p, dp, am, dy2, gamma, beta, sm, si, dg2, db2, N, H, H, C, ws, ws_bytes, dt)
    # the fused backward sums dgamma / dbeta at pooled resolution (no intermediate rounding of the pool gradient,
    # xhat from the stored activation): equal to the unfused chain up to rounding
    f32 = dtype == torch.float32
    assert relerr(db2, db) < (1e-6 if f32 else 2e-3) and relerr(dg2, dg) < (1e-5 if f32 else 5e-3)
    assert relerr(dy2.float(), dy.float()) < (1e-5 if f32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batched_weight_prepare_and_finalize_match_single_calls(cuda, dtype):
    """primia_conv_weight_prepare_many / primia_conv_wgrad_finalize_many (tiled kernels for the regular
    convs, element-wise for the stem) against the per-conv entry points, bit for bit."""
    import ctypes

    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(3)
    shapes = [(4, 64, 7, 2, 3, 3), (64, 64, 3, 1, 1, 64), (64, 128, 3, 2, 1, 64), (64, 128, 1, 2, 0, 64),
              (128, 256, 3, 1, 1, 128), (256, 512, 1, 2, 0, 256)]   # C(stored), K, R, stride, pad, c_real
    descs, creal, w, wf, wd, wf1, wd1, acc, dw, dw1 = [], [], [], [], [], [], [], [], [], []
    for C, K, R, s, p, cr in shapes:
        d = ConvDesc.make(2, 16, 16, C, K, R, R, s, p)
        descs.append(d)
        creal.append(cr)
        w.append((torch.randn(K, cr, R, R, generator=g) * 0.1).to(cuda))
        nf, nd = query("primia_conv_wfwd_elems", d), query("primia_conv_wdgrad_elems", d)
        wf.append(torch.zeros(nf, dtype=dtype, device=cuda)); wf1.append(torch.zeros(nf, dtype=dtype, device=cuda))
        has_d = R != 7
        wd.append(torch.zeros(nd, dtype=dtype, device=cuda) if has_d else None)
        wd1.append(torch.zeros(nd, dtype=dtype, device=cuda) if has_d else None)
        acc.append(torch.randn(nf, generator=g).to(cuda))
        dw.append(torch.zeros(K, cr, R, R, device=cuda)); dw1.append(torch.zeros(K, cr, R, R, device=cuda))
    n = len(shapes)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    arr = lambda ts: (ctypes.c_void_p * n)(*[vp(t) for t in ts])
    call("primia_conv_weight_prepare_many", (ConvDesc * n)(*descs), (ctypes.c_int * n)(*creal), arr(w), arr(wf),
         arr(wd), n, dt)
    call("primia_conv_wgrad_finalize_many", (ConvDesc * n)(*descs), (ctypes.c_int * n)(*creal), arr(acc), arr(dw), n)
    for i in range(n):
        call("primia_conv_weight_prepare", descs[i], creal[i], w[i], wf1[i], wd1[i], dt)
        call("primia_conv_wgrad_finalize", descs[i], creal[i], acc[i], dw1[i])
        assert torch.equal(wf[i], wf1[i]), f"w_fwd {shapes[i]}"
        if wd[i] is not None:
            assert torch.equal(wd[i], wd1[i]), f"w_dgrad {shapes[i]}"
        assert torch.equal(dw[i], dw1[i]), f"dw {shapes[i]}"


@pytest.mark.parametrize("N,S", [(2, 64), (3, 32), (1, 96)])
def test_stem_conv_fwd_padded_matches_conv2d_fwd(cuda, N, S):
    """primia_stem_conv_fwd on the padded input against torch (bf16 tolerance) and against the generic
    implicit GEMM on the unpadded input (same accumulation order: bit-identical)."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(S)
    x = rnd(torch.randn(N, 3, S, S, generator=g), dtype)
    w = rnd(torch.randn(64, 3, 7, 7, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, S, S, 4, 64, 7, 7, 2, 3)
    wf, _ = prep_weights(desc, w, dtype, cuda, 3, need_dgrad=False)
    x4 = torch.empty(N * S * S, 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc", x.to(cuda), x4, N, 3, S, S, 4, dt)
    y_ref = torch.empty(N * (S // 2) ** 2, 64, dtype=dtype, device=cuda)
    call("primia_conv2d_fwd", desc, x4, wf, y_ref, dt)
    Hp, Wp = S + 6, S + 8
    xp = torch.zeros(N * Hp * Wp, 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc_padded", x.to(cuda), xp, N, 3, S, S, 4, 3, 3, Hp, Wp, dt)
    y = torch.full_like(y_ref, float("nan"))
    call("primia_stem_conv_fwd", xp, wf, y, N, S, S, dt)
    t_ref = F.conv2d(x, w, None, 2, 3)
    assert relerr(from_nhwc(y, N, S // 2, S // 2), t_ref) < tol(dtype)
    assert torch.equal(y, y_ref)


@pytest.mark.parametrize("N,S", [(2, 64), (75, 64), (3, 96)])
def test_stem_conv_wgrad_padded_matches_unpadded(cuda, N, S):
    # (75, 64): 600 patches over 256 blocks -> 3 patches per block (multi-stage ring, ragged last block)
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(9)
    x = rnd(torch.randn(N, 3, S, S, generator=g), dtype)
    dy = to_nhwc(rnd(torch.randn(N, 64, S // 2, S // 2, generator=g), dtype), dtype, cuda)
    desc = ConvDesc.make(N, S, S, 4, 64, 7, 7, 2, 3)
    x4 = torch.empty(N * S * S, 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc", x.to(cuda), x4, N, 3, S, S, 4, dt)
    xp = torch.zeros(N * (S + 6) * (S + 8), 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc_padded", x.to(cuda), xp, N, 3, S, S, 4, 3, 3, S + 6, S + 8, dt)
    n = query("primia_conv_wfwd_elems", desc)
    a0, a1 = torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    call("primia_conv2d_wgrad", desc, x4, dy, a0, dt)
    call("primia_stem_conv_wgrad", xp, dy, a1, N, S, S, dt)
    d0, d1 = torch.empty(64, 3, 7, 7, device=cuda), torch.empty(64, 3, 7, 7, device=cuda)
    call("primia_conv_wgrad_finalize", desc, 3, a0, d0)
    call("primia_conv_wgrad_finalize", desc, 3, a1, d1)
    assert relerr(d1, d0) < 2e-5   # fp32 atomics: order differs run to run
    assert float(a1.view(64, 256)[:, 7 * 32:].abs().max()) == 0.0 and float(a1.view(64, 8, 8, 4)[:, :, 7].abs().max()) == 0.0


@pytest.mark.parametrize("N,H", [(3, 16), (2, 56), (5, 12)])
def test_layer1_conv_emits_batchnorm_partials(cuda, N, H):
    """The 64->64 kernel's per-block BatchNorm partials: summed over the slots they equal the column sums of
    the output AS STORED, and bn_fwd_train_from_sums on them matches bn_fwd_train on the same output."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(H)
    x = to_nhwc(rnd(torch.randn(N, 64, H, H, generator=g), dtype), dtype, cuda)
    w = rnd(torch.randn(64, 64, 3, 3, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, H, H, 64, 64, 3, 3, 1, 1)
    wf, _ = prep_weights(desc, w, dtype, cuda, 64)
    slots = query("primia_conv_stat_slots_for", desc, dt)
    assert 1 <= slots <= 512 and slots != query("primia_conv_stat_slots")
    M = N * H * H
    y = torch.empty(M, 64, dtype=dtype, device=cuda)
    sums = torch.full((slots, 2, 64), float("nan"), device=cuda)   # written, not accumulated
    call("primia_conv2d_fwd_stats", desc, x, wf, y, sums, dt)
    y2 = torch.empty_like(y)
    call("primia_conv2d_fwd", desc, x, wf, y2, dt)
    assert torch.equal(y, y2)
    yf = y.float()
    assert relerr(sums[:, 0].sum(0), yf.sum(0)) < 1e-5 and relerr(sums[:, 1].sum(0), (yf * yf).sum(0)) < 1e-5
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(cuda), torch.randn(64, generator=g).to(cuda)
    ws_bytes = query("primia_bn_workspace_bytes", M, 64)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    outs = []
    for fused in (False, True):
        rm, rv = torch.zeros(64, device=cuda), torch.ones(64, device=cuda)
        sm, si, z = torch.empty(64, device=cuda), torch.empty(64, device=cuda), torch.empty_like(y)
        if fused:
            call("primia_bn_fwd_train_from_sums", y, None, z, gamma, beta, rm, rv, sm, si, sums, slots, M, 64, 1e-5, 0.1, 1, dt)
        else:
            call("primia_bn_fwd_train", y, None, z, gamma, beta, rm, rv, sm, si, M, 64, 1e-5, 0.1, 1, ws, ws_bytes, dt)
        outs.append((sm, si, rm, rv, z.float()))
    for a, b in zip(*outs):
        assert relerr(a, b) < 1e-5


@pytest.mark.parametrize("N,H,C,K", [(2, 28, 128, 128), (4, 14, 256, 256), (5, 7, 512, 512), (3, 10, 128, 256)])
def test_wide_conv_emits_batchnorm_partials(cuda, N, H, C, K):
    """The linear-halo kernel (wide 3x3 / stride-1 layers) emits one deterministic BatchNorm partial per pixel tile:
    summed over the slots they equal the column sums of the output AS STORED (ragged last tile included)."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(H + C)
    x = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g), dtype), dtype, cuda)
    w = rnd(torch.randn(K, C, 3, 3, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, H, H, C, K, 3, 3, 1, 1)
    wf, _ = prep_weights(desc, w, dtype, cuda, C)
    slots = query("primia_conv_stat_slots_for", desc, dt)
    M = N * H * H
    assert slots == (M + 195) // 196      # 196-pixel tiles (conv3x3_lh2.hip; 392 when every CU gets a tile)
    y = torch.empty(M, K, dtype=dtype, device=cuda)
    sums = torch.full((slots, 2, K), float("nan"), device=cuda)   # written, not accumulated
    call("primia_conv2d_fwd_stats", desc, x, wf, y, sums, dt)
    y2 = torch.empty_like(y)
    call("primia_conv2d_fwd", desc, x, wf, y2, dt)
    assert torch.equal(y, y2)
    yf = y.float()
    assert relerr(sums[:, 0].sum(0), yf.sum(0)) < 1e-5 and relerr(sums[:, 1].sum(0), (yf * yf).sum(0)) < 1e-5
    sums2 = torch.full_like(sums, float("nan"))
    call("primia_conv2d_fwd_stats", desc, x, wf, y2, sums2, dt)
    assert torch.equal(sums, sums2)


@pytest.mark.parametrize("N,H,C,K,bm", [(160, 28, 128, 128, 392), (100, 28, 128, 128, 196), (300, 14, 128, 256, 392),
                                        (37, 7, 256, 384, 196)])
def test_wide_conv_persistent_tiles(cuda, N, H, C, K, bm):
    """conv3x3_lh2.hip with MORE tiles than CUs (a block walks several tiles, rings running through the tile
    boundaries, uneven tile counts per block): forward + per-tile BatchNorm partials, data gradient, masked
    accumulating data gradient — against torch-CPU autograd on bf16-rounded operands."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(N + H)
    x = rnd(torch.randn(N, C, H, H, generator=g), dtype)
    w = rnd(torch.randn(K, C, 3, 3, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, H, H, C, K, 3, 3, 1, 1)
    wf, wd = prep_weights(desc, w, dtype, cuda, C)
    M = N * H * H
    slots = query("primia_conv_stat_slots_for", desc, dt)
    assert slots == (M + bm - 1) // bm
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, w, None, 1, 1)
    dy = rnd(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(dy)
    xd = to_nhwc(x, dtype, cuda)
    y = torch.empty(M, K, dtype=dtype, device=cuda)
    sums = torch.full((slots, 2, K), float("nan"), device=cuda)
    call("primia_conv2d_fwd_stats", desc, xd, wf, y, sums, dt)
    assert relerr(from_nhwc(y, N, H, H), y_ref.detach()) < tol(dtype)
    yf = y.float()
    assert relerr(sums[:, 0].sum(0), yf.sum(0)) < 1e-5 and relerr(sums[:, 1].sum(0), (yf * yf).sum(0)) < 1e-5
    # every tile's partial on its own
    pad = slots * bm - M
    yt = torch.cat([yf, yf.new_zeros(pad, K)]).view(slots, bm, K)
    assert relerr(sums[:, 0], yt.sum(1)) < 1e-5 and relerr(sums[:, 1], (yt * yt).sum(1)) < 1e-5
    y2 = torch.empty_like(y)
    call("primia_conv2d_fwd", desc, xd, wf, y2, dt)
    assert torch.equal(y, y2)
    dyd = to_nhwc(dy, dtype, cuda)
    dx = torch.empty(M, C, dtype=dtype, device=cuda)
    call("primia_conv2d_dgrad", desc, dyd, wd, dx, 0, dt)
    assert relerr(from_nhwc(dx, N, H, H), xr.grad) < tol(dtype)
    base = rnd(torch.randn(N, C, H, H, generator=g), dtype)
    keep = torch.rand(N, C, H, H, generator=g) > 0.4
    based = to_nhwc(base, dtype, cuda)
    bits = to_nhwc(keep.float(), torch.float32, cuda).to(torch.uint8).view(M * C // 8, 8)
    mask = (bits << torch.arange(8, device=cuda, dtype=torch.uint8)).sum(1).to(torch.uint8)
    dx2 = based.clone()
    call("primia_conv2d_dgrad_masked_acc", desc, dyd, wd, dx2, mask, dt)
    assert relerr(from_nhwc(dx2, N, H, H), xr.grad + base * keep) < tol(dtype)
    dx3 = based.clone()
    call("primia_conv2d_dgrad", desc, dyd, wd, dx3, 1, dt)
    assert relerr(from_nhwc(dx3, N, H, H), xr.grad + base) < tol(dtype)


@pytest.mark.parametrize("N,H,C,K", [(100, 28, 128, 128), (37, 7, 256, 384), (256, 7, 512, 512), (3, 14, 256, 256)])
def test_loader_wave_kernel_is_bit_identical_to_the_linear_halo_kernel(cuda, N, H, C, K):
    """conv3x3_lh4_kernel (8 matrix + 4 loader waves, one barrier per step; serves the shapes that take 196-pixel tiles) against
    conv3x3_lh2_kernel on the same tiles (option lh4 = 0): forward output, per-tile BatchNorm partials, data gradient, accumulating
    and masked-accumulating data gradient — the SAME BITS (same tiles, same order of every sum)."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(7 * N + H)
    x = rnd(torch.randn(N, C, H, H, generator=g).clamp(min=0), dtype)
    w = rnd(torch.randn(K, C, 3, 3, generator=g) * 0.05, dtype)
    dy = rnd(torch.randn(N, K, H, H, generator=g), dtype)
    desc = ConvDesc.make(N, H, H, C, K, 3, 3, 1, 1)
    wf, wd = prep_weights(desc, w, dtype, cuda, C)
    M = N * H * H
    xd, dyd = to_nhwc(x, dtype, cuda), to_nhwc(dy, dtype, cuda)
    base = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g), dtype), dtype, cuda)
    mask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).to(cuda)
    outs = {}
    try:
        for lh4 in (1, 0):
            _lib.set_option("lh4", lh4)
            slots = query("primia_conv_stat_slots_for", desc, dt)
            assert slots == (M + 195) // 196          # 196-pixel tiles either way
            y = torch.full((M, K), float("nan"), dtype=dtype, device=cuda)
            sums = torch.full((slots, 2, K), float("nan"), device=cuda)
            call("primia_conv2d_fwd_stats", desc, xd, wf, y, sums, dt)
            dx = torch.full((M, C), float("nan"), dtype=dtype, device=cuda)
            call("primia_conv2d_dgrad", desc, dyd, wd, dx, 0, dt)
            dxa = base.clone()
            call("primia_conv2d_dgrad", desc, dyd, wd, dxa, 1, dt)
            dxm = base.clone()
            call("primia_conv2d_dgrad_masked_acc", desc, dyd, wd, dxm, mask, dt)
            outs[lh4] = (y, sums, dx, dxa, dxm)
    finally:
        _lib.set_option("lh4", 1)
    for a, b, name in zip(outs[1], outs[0], ("y", "partials", "dx", "dx +=", "dx += masked")):
        assert torch.equal(a, b), name
    assert torch.isfinite(outs[1][0].float()).all() and float(outs[1][0].float().abs().max()) > 0


def test_stem_conv_emits_batchnorm_partials_and_fused_tail_from_sums(cuda):
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    N, S = 3, 64
    g = torch.Generator().manual_seed(31)
    x = rnd(torch.randn(N, 3, S, S, generator=g), dtype)
    w = rnd(torch.randn(64, 3, 7, 7, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, S, S, 4, 64, 7, 7, 2, 3)
    wf, _ = prep_weights(desc, w, dtype, cuda, 3, need_dgrad=False)
    xp = torch.zeros(N * (S + 6) * (S + 8), 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc_padded", x.to(cuda), xp, N, 3, S, S, 4, 3, 3, S + 6, S + 8, dt)
    Ho = S // 2
    M = N * Ho * Ho
    y0 = torch.empty(M, 64, dtype=dtype, device=cuda)
    call("primia_stem_conv_fwd", xp, wf, y0, N, S, S, dt)
    slots = query("primia_stem_conv_stat_slots", N, S, S)
    sums = torch.full((slots, 2, 64), float("nan"), device=cuda)
    y = torch.empty_like(y0)
    call("primia_stem_conv_fwd_stats", xp, wf, y, sums, N, S, S, dt)
    assert torch.equal(y, y0)
    yf = y.float()
    assert relerr(sums[:, 0].sum(0), yf.sum(0)) < 1e-5 and relerr(sums[:, 1].sum(0), (yf * yf).sum(0)) < 1e-5
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(cuda), torch.randn(64, generator=g).to(cuda)
    ws_bytes = query("primia_bn_workspace_bytes", M, 64)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    Hp = (Ho - 1) // 2 + 1
    res = []
    for fused in (False, True):
        rm, rv = torch.zeros(64, device=cuda), torch.ones(64, device=cuda)
        sm, si = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
        p = torch.empty(N * Hp * Hp, 64, dtype=dtype, device=cuda)
        am = torch.empty(N * Hp * Hp, 64, dtype=torch.uint8, device=cuda)
        if fused:
            call("primia_bn_relu_maxpool_fwd_from_sums", y, p, am, gamma, beta, rm, rv, sm, si, sums, slots, N, Ho, Ho, 64,
                 1e-5, 0.1, dt)
        else:
            call("primia_bn_relu_maxpool_fwd", y, p, am, gamma, beta, rm, rv, sm, si, N, Ho, Ho, 64, 1e-5, 0.1, ws,
                 ws_bytes, dt)
        res.append((sm, si, rm, rv, p.float()))
    for a, b in zip(*res):
        assert relerr(a, b) < 1e-5


@pytest.mark.parametrize("N,H,C,K", [(4, 28, 128, 256), (2, 56, 64, 128), (3, 14, 256, 512), (2, 6, 64, 128)])
def test_transition_block_weight_gradients_in_one_launch(cuda, N, H, C, K):
    """primia_conv2d_wgrad_pair_ws (conv1 3x3/2 + downsample 1x1/2 of a transition block, the downsample as a tenth tap of
    the per-tap kernel) against the two single calls and against the fp32 oracle on the rounded operands."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(5 + H)
    x = rnd(torch.randn(N, C, H, H, generator=g), dtype)
    Ho = H // 2
    dy1 = rnd(torch.randn(N, K, Ho, Ho, generator=g), dtype)
    dyd = rnd(torch.randn(N, K, Ho, Ho, generator=g), dtype)
    d1, dd = ConvDesc.make(N, H, H, C, K, 3, 3, 2, 1), ConvDesc.make(N, H, H, C, K, 1, 1, 2, 0)
    need = query("primia_conv_wgrad_pair_ws_bytes", d1, dd, dt)
    assert need > 0
    xd, dy1d, dydd = to_nhwc(x, dtype, cuda), to_nhwc(dy1, dtype, cuda), to_nhwc(dyd, dtype, cuda)
    ws = torch.full((need // 4,), float("nan"), device=cuda)
    n1, nd = query("primia_conv_wfwd_elems", d1), query("primia_conv_wfwd_elems", dd)
    a1, ad = torch.full((n1,), float("nan"), device=cuda), torch.full((nd,), float("nan"), device=cuda)
    call("primia_conv2d_wgrad_pair_ws", d1, xd, dy1d, a1, dd, dydd, ad, ws, need, dt)
    # single calls
    s1 = query("primia_conv_wgrad_ws_bytes", d1, dt)
    sd = query("primia_conv_wgrad_ws_bytes", dd, dt)
    w1, wd = torch.empty(max(s1, 16) // 4, device=cuda), torch.empty(max(sd, 16) // 4, device=cuda)
    b1, bd = torch.zeros(n1, device=cuda), torch.zeros(nd, device=cuda)
    call("primia_conv2d_wgrad_ws", d1, xd, dy1d, b1, w1, s1, dt)
    call("primia_conv2d_wgrad_ws", dd, xd, dydd, bd, wd, sd, dt)
    assert relerr(a1, b1) < 2e-6 and relerr(ad, bd) < 2e-6      # same products, another grouping of the ordered sums
    # oracle
    g1, gd = torch.empty(K, C, 3, 3, device=cuda), torch.empty(K, C, 1, 1, device=cuda)
    call("primia_conv_wgrad_finalize", d1, C, a1, g1)
    call("primia_conv_wgrad_finalize", dd, C, ad, gd)
    xr = x.clone().requires_grad_(False)
    w1r = torch.zeros(K, C, 3, 3, requires_grad=True)
    wdr = torch.zeros(K, C, 1, 1, requires_grad=True)
    (F.conv2d(xr, w1r, None, 2, 1) * dy1).sum().backward()
    (F.conv2d(xr, wdr, None, 2, 0) * dyd).sum().backward()
    assert relerr(g1, w1r.grad) < 1e-4 and relerr(gd, wdr.grad) < 1e-4


@pytest.mark.parametrize("N,H,C,count", [(32, 56, 64, 4), (24, 28, 128, 3), (16, 14, 256, 3), (40, 7, 512, 3), (3, 16, 64, 2)])
def test_same_shape_layers_share_one_weight_gradient_launch(cuda, N, H, C, count):
    """primia_conv2d_wgrad_group_ws: n layers of one 3x3 / stride-1 shape in one launch of the patch kernel (blocks split n
    ways, each layer's slabs reduced in split order) against n single calls — the same products, another grouping of
    the ordered sums — and every layer bit-identical run after run."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    d = ConvDesc.make(N, H, H, C, C, 3, 3, 1, 1)
    n = query("primia_conv_wgrad_group_size", d, count, dt)
    assert 0 <= n <= min(count, 4)
    if n < 2:
        pytest.skip(f"group size {n} for this shape and batch")
    need = query("primia_conv_wgrad_group_ws_bytes", d, n, dt)
    assert need > 0
    g = torch.Generator().manual_seed(H + C)
    xs = [to_nhwc(rnd(torch.randn(N, C, H, H, generator=g).relu(), dtype), dtype, cuda) for _ in range(n)]
    dys = [to_nhwc(rnd(torch.randn(N, C, H, H, generator=g) * 1e-2, dtype), dtype, cuda) for _ in range(n)]
    ne = query("primia_conv_wfwd_elems", d)

    def grouped():
        ws = torch.full((need // 4,), float("nan"), device=cuda)
        accs = [torch.full((ne,), float("nan"), device=cuda) for _ in range(n)]
        args = []
        for i in range(4):
            args += [xs[i], dys[i], accs[i]] if i < n else [None, None, None]
        call("primia_conv2d_wgrad_group_ws", d, n, *args, ws, need, dt)
        return accs

    a, b = grouped(), grouped()
    s1 = query("primia_conv_wgrad_ws_bytes", d, dt)
    w1 = torch.empty(max(s1, 16) // 4, device=cuda)
    for i in range(n):
        assert torch.equal(a[i], b[i])
        single = torch.zeros(ne, device=cuda)
        call("primia_conv2d_wgrad_ws", d, xs[i], dys[i], single, w1, s1, dt)
        assert relerr(a[i], single) < 2e-6, i


@pytest.mark.parametrize("N,S", [(2, 64), (7, 224), (3, 96), (5, 32), (2, 256)])
def test_stem_forward_without_its_activation(cuda, N, S):
    """primia_stem_conv_stats -> primia_bn_finalize_stats -> primia_stem_conv_pool (conv1's output is never stored) against
    primia_stem_conv_fwd_stats -> primia_bn_relu_maxpool_fwd_from_sums on a stored y: the same partial sums, statistics,
    pooled activations and argmax codes BIT for bit; with y requested, the same conv output.  (5, 32): one patch per band
    and per image row (every halo comes from the image border or a one-patch carry); (2, 256): eight patches per row."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(2000 + N + S)
    x = rnd(torch.randn(N, 3, S, S, generator=g), dtype)
    w = rnd(torch.randn(64, 3, 7, 7, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, S, S, 4, 64, 7, 7, 2, 3)
    wf, _ = prep_weights(desc, w, dtype, cuda, 3, need_dgrad=False)
    xp = torch.zeros(N * (S + 6) * (S + 8), 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc_padded", x.to(cuda), xp, N, 3, S, S, 4, 3, 3, S + 6, S + 8, dt)
    Ho, Hq = S // 2, S // 4
    M = N * Ho * Ho
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(cuda), (torch.randn(64, generator=g) * 0.3).to(cuda)
    gamma[5] = -gamma[5]          # a negative scale: the maximum of z is then a minimum of y
    slots = query("primia_stem_conv_stat_slots", N, S, S)
    # the chain on a stored y
    y0 = torch.empty(M, 64, dtype=dtype, device=cuda)
    sums0 = torch.full((slots, 2, 64), float("nan"), device=cuda)
    call("primia_stem_conv_fwd_stats", xp, wf, y0, sums0, N, S, S, dt)
    rm0, rv0 = torch.zeros(64, device=cuda), torch.ones(64, device=cuda)
    sm0, si0 = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
    p0 = torch.empty(N * Hq * Hq, 64, dtype=dtype, device=cuda)
    a0 = torch.empty(N * Hq * Hq, 64, dtype=torch.uint8, device=cuda)
    call("primia_bn_relu_maxpool_fwd_from_sums", y0, p0, a0, gamma, beta, rm0, rv0, sm0, si0, sums0, slots, N, Ho, Ho, 64, 1e-5,
         0.1, dt)
    # two passes over the input
    assert query("primia_stem_conv_pool_ok", N, S, S, dt) == 1
    sums1 = torch.full((slots, 2, 64), float("nan"), device=cuda)
    call("primia_stem_conv_stats", xp, wf, sums1, N, S, S, dt)
    assert torch.equal(sums1, sums0)
    rm1, rv1 = torch.zeros(64, device=cuda), torch.ones(64, device=cuda)
    sm1, si1 = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
    call("primia_bn_finalize_stats", sums1, slots, M, 64, 1e-5, 0.1, rm1, rv1, sm1, si1)
    for a, b in ((sm1, sm0), (si1, si0), (rm1, rm0), (rv1, rv0)):
        assert torch.equal(a, b)
    for with_y in (False, True):
        p1 = torch.full((N * Hq * Hq, 64), float("nan"), dtype=dtype, device=cuda)
        a1 = torch.full((N * Hq * Hq, 64), 99, dtype=torch.uint8, device=cuda)
        y1 = torch.full((M, 64), float("nan"), dtype=dtype, device=cuda) if with_y else None
        call("primia_stem_conv_pool", xp, wf, y1, p1, a1, gamma, beta, sm1, si1, N, S, S, dt)
        assert torch.equal(p1, p0), f"{int((p1 != p0).sum())} of {p0.numel()} pooled values differ"
        assert torch.equal(a1, a0), f"{int((a1 != a0).sum())} argmax codes differ"
        if with_y:
            assert torch.equal(y1, y0)
    assert int(a0.max()) <= 8 and float(p0.float().max()) > 0


@pytest.mark.parametrize("N,S", [(2, 64), (75, 64), (3, 96), (5, 32)])
def test_stem_backward_fused_is_bit_identical_to_the_chain(cuda, N, S):
    """primia_stem_bwd_fused (bn1 <- relu <- maxpool backward apply inside conv1's weight-gradient kernel, dy never
    stored) against primia_bn_relu_maxpool_bwd -> primia_stem_conv_wgrad_ws: same dgamma / dbeta (the same reduction
    kernels run), and the SAME BITS in the weight gradient.  (75, 64): several patches per block, ragged last block;
    (5, 32): one patch per image, every window column / row at the pooled image's border."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(1000 + N + S)
    x = rnd(torch.randn(N, 3, S, S, generator=g), dtype)
    w = rnd(torch.randn(64, 3, 7, 7, generator=g) * 0.05, dtype)
    desc = ConvDesc.make(N, S, S, 4, 64, 7, 7, 2, 3)
    wf, _ = prep_weights(desc, w, dtype, cuda, 3, need_dgrad=False)
    xp = torch.zeros(N * (S + 6) * (S + 8), 4, dtype=dtype, device=cuda)
    call("primia_nchw_to_nhwc_padded", x.to(cuda), xp, N, 3, S, S, 4, 3, 3, S + 6, S + 8, dt)
    Ho = S // 2
    Hq = (Ho - 1) // 2 + 1
    M = N * Ho * Ho
    y = torch.empty(M, 64, dtype=dtype, device=cuda)
    call("primia_stem_conv_fwd", xp, wf, y, N, S, S, dt)
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(cuda), (torch.randn(64, generator=g) * 0.3).to(cuda)
    ws_bytes = query("primia_bn_workspace_bytes", M, 64)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    rm, rv = torch.zeros(64, device=cuda), torch.ones(64, device=cuda)
    sm, si = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
    p = torch.empty(N * Hq * Hq, 64, dtype=dtype, device=cuda)
    am = torch.empty(N * Hq * Hq, 64, dtype=torch.uint8, device=cuda)
    call("primia_bn_relu_maxpool_fwd", y, p, am, gamma, beta, rm, rv, sm, si, N, Ho, Ho, 64, 1e-5, 0.1, ws, ws_bytes, dt)
    dp = to_nhwc(rnd(torch.randn(N, 64, Hq, Hq, generator=g), dtype), dtype, cuda)
    wws_bytes = query("primia_stem_conv_wgrad_ws_bytes", N, S, S)
    assert wws_bytes > 0
    wws = torch.empty(wws_bytes // 4, device=cuda)
    n = query("primia_conv_wfwd_elems", desc)

    # the chain
    dy = torch.empty_like(y)
    dg, db = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
    call("primia_bn_relu_maxpool_bwd", y, p, dp, am, dy, gamma, beta, sm, si, dg, db, N, Ho, Ho, 64, ws, ws_bytes, dt)
    a0 = torch.full((n,), float("nan"), device=cuda)
    call("primia_stem_conv_wgrad_ws", xp, dy, a0, wws, wws_bytes, N, S, S, dt)
    # fused: sums only, then the weight gradient straight from y / dpooled / argmax
    dg2, db2 = torch.empty(64, device=cuda), torch.empty(64, device=cuda)
    call("primia_bn_relu_maxpool_bwd", y, p, dp, am, None, gamma, beta, sm, si, dg2, db2, N, Ho, Ho, 64, ws, ws_bytes, dt)
    assert torch.equal(dg2, dg) and torch.equal(db2, db)
    wws.fill_(float("nan"))
    a1 = torch.full((n,), float("nan"), device=cuda)
    call("primia_stem_bwd_fused", xp, y, dp, am, gamma, beta, sm, si, dg2, db2, a1, wws, wws_bytes, N, S, S, dt)
    assert torch.isfinite(a1).all()
    assert float(a0.abs().max()) > 0
    assert torch.equal(a1, a0), f"max diff {float((a1 - a0).abs().max())} of {float(a0.abs().max())}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm_relu_mask_variant_is_bit_identical(cuda, dtype):
    """primia_bn_fwd_train_mask / primia_bn_bwd_mask (1-bit ReLU mask for residual layers) against
    primia_bn_fwd_train / primia_bn_bwd: identical z, statistics, dy, g_out, dgamma, dbeta."""
    dt = _lib.dtype_code(dtype)
    N, H, C = 3, 9, 128
    M = N * H * H
    g = torch.Generator().manual_seed(99)
    yd = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g) * 2 + 0.5, dtype), dtype, cuda)
    rd = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g), dtype), dtype, cuda)
    dzd = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g), dtype), dtype, cuda)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    ws_bytes = query("primia_bn_workspace_bytes", M, C)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    outs = []
    for masked in (False, True):
        rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
        sm, si, z = torch.empty(C, device=cuda), torch.empty(C, device=cuda), torch.empty_like(yd)
        dy, go = torch.empty_like(yd), torch.empty_like(yd)
        dg, db = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
        if masked:
            mask = torch.empty(yd.numel() * yd.element_size() // 16, dtype=torch.uint8, device=cuda)
            call("primia_bn_fwd_train_mask", yd, rd, z, mask, gamma, beta, rm, rv, sm, si, None, 0, M, C, 1e-5, 0.1, ws,
                 ws_bytes, dt)
            call("primia_bn_bwd_mask", yd, mask, dzd, dy, go, gamma, sm, si, dg, db, M, C, ws, ws_bytes, dt)
        else:
            call("primia_bn_fwd_train", yd, rd, z, gamma, beta, rm, rv, sm, si, M, C, 1e-5, 0.1, 1, ws, ws_bytes, dt)
            call("primia_bn_bwd", yd, z, dzd, dy, go, gamma, sm, si, dg, db, M, C, 1, ws, ws_bytes, dt)
        outs.append((z, sm, si, rm, rv, dy, go, dg, db))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,C,K", [(2, 16, 64, 128), (3, 28, 128, 256), (4, 14, 256, 512), (1, 6, 64, 128)])
def test_conv_dgrad_pair_matches_two_passes(cuda, dtype, N, H, C, K):
    """Transition block: dgrad(conv1 3x3/2) + dgrad(downsample 1x1/2) in one pass vs torch autograd of the sum,
    and vs the two-call form it replaces (same values up to ONE bf16 rounding instead of two)."""
    g = torch.Generator().manual_seed(77 + H + C)
    x = rnd(torch.randn(N, C, H, H, generator=g), dtype)
    w1 = rnd(torch.randn(K, C, 3, 3, generator=g) * 0.05, dtype)
    wd_ = rnd(torch.randn(K, C, 1, 1, generator=g) * 0.1, dtype)
    d1 = ConvDesc.make(N, H, H, C, K, 3, 3, 2, 1)
    dd = ConvDesc.make(N, H, H, C, K, 1, 1, 2, 0)
    dt = _lib.dtype_code(dtype)
    _, w1d = prep_weights(d1, w1, dtype, cuda, C)
    _, wdd = prep_weights(dd, wd_, dtype, cuda, C)
    xr = x.clone().requires_grad_(True)
    y1 = F.conv2d(xr, w1, None, 2, 1)
    yd = F.conv2d(xr, wd_, None, 2, 0)
    dy1 = rnd(torch.randn(y1.shape, generator=g), dtype)
    dyd = rnd(torch.randn(yd.shape, generator=g), dtype)
    (y1 * dy1).sum().add((yd * dyd).sum()).backward()
    dy1d, dydd = to_nhwc(dy1, dtype, cuda), to_nhwc(dyd, dtype, cuda)
    dx = torch.full((N * H * H, C), 3.0, dtype=dtype, device=cuda)   # dirty: the call overwrites
    call("primia_conv2d_dgrad_pair", d1, dy1d, w1d, dd, dydd, wdd, dx, dt)
    assert relerr(from_nhwc(dx, N, H, H), xr.grad) < tol(dtype)
    dx2 = torch.empty_like(dx)
    call("primia_conv2d_dgrad", d1, dy1d, w1d, dx2, 0, dt)
    call("primia_conv2d_dgrad", dd, dydd, wdd, dx2, 1, dt)
    assert relerr(dx, dx2) < tol(dtype)
    # argument checks: the second descriptor must be the 1x1/2 sibling of the first
    with pytest.raises(_lib.PrimiaError):
        call("primia_conv2d_dgrad_pair", d1, dy1d, w1d, d1, dydd, wdd, dx, dt)


@pytest.mark.parametrize("N,H,C", [(2, 16, 64), (1, 56, 64), (2, 28, 128), (4, 14, 256), (5, 7, 512), (3, 56, -64), (2, 20, -64)])
def test_dgrad_masked_accumulate(cuda, c64_blocks, N, H, C):
    """primia_conv2d_dgrad_masked_acc: dx = relu_mask(dx) + dgrad(dy), bit-identical to masking dx first and then
    calling primia_conv2d_dgrad(accumulate = 1) (identity blocks: the BatchNorm backward no longer writes the
    masked residual gradient)."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    if C < 0:      # 64 channels on a handful of persistent blocks: many patches per block (the counted waits of the ring)
        C = -C
        c64_blocks(3)
    desc = ConvDesc.make(N, H, H, C, C, 3, 3, 1, 1)
    assert query("primia_conv_dgrad_masked_acc_ok", desc, dt) == 1
    g = torch.Generator().manual_seed(3 * H + C)
    w = rnd(torch.randn(C, C, 3, 3, generator=g) * 0.05, dtype)
    _, wd = prep_weights(desc, w, dtype, cuda, C)
    dy = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g), dtype), dtype, cuda)
    base = to_nhwc(rnd(torch.randn(N, C, H, H, generator=g), dtype), dtype, cuda)
    M = N * H * H
    bits = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.int32).to(torch.uint8).to(cuda)
    keep = ((bits.to(torch.int32).unsqueeze(1) >> torch.arange(8, device=cuda)) & 1).reshape(M, C).to(torch.bool)
    ref = torch.where(keep, base, torch.zeros_like(base))
    call("primia_conv2d_dgrad", desc, dy, wd, ref, 1, dt)
    out = base.clone()
    call("primia_conv2d_dgrad_masked_acc", desc, dy, wd, out, bits, dt)
    assert torch.equal(out, ref)
    # a shape no masking write-back serves
    d2 = ConvDesc.make(2, 16, 16, 64, 128, 3, 3, 2, 1)
    assert query("primia_conv_dgrad_masked_acc_ok", d2, dt) == 0
    assert query("primia_conv_dgrad_masked_acc_ok", desc, _lib.dtype_code(torch.float32)) == 0


@pytest.fixture
def c64_blocks():
    """set the number of persistent blocks of conv3x3_c64_kernel (few blocks = many patches per block: the counted waits of
    its staging ring only run then); restored afterwards"""
    def set_blocks(n):
        _lib.set_option("c64_blocks", n)
    yield set_blocks
    _lib.set_option("c64_blocks", 0)


@pytest.mark.parametrize("N,H,blocks", [(2, 16, 0), (1, 56, 0), (3, 12, 0), (9, 8, 0), (2, 24, 0), (2, 24, 3), (4, 56, 5), (3, 12, 2)])
def test_masked_accumulate_dgrad_with_residual_bn_backward_sums(cuda, c64_blocks, N, H, blocks):
    """primia_conv2d_dgrad_masked_acc_bnsums, mode 2 (conv3x3_c64_kernel<true, 3, 2>): the accumulating data gradient of an
    identity block's conv1 also forms the backward sums of the residual BatchNorm of the block in FRONT of it, +
    primia_bn_bwd_mask_from_sums, against primia_conv2d_dgrad_masked_acc followed by primia_bn_bwd_mask with its own reduction
    pass: dx bit-identical, dgamma / dbeta to summation order, dy to one bf16 rounding; partials bit-repeatable."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    C = 64
    g = torch.Generator().manual_seed(11 * N + H)
    c64_blocks(blocks)
    desc = ConvDesc.make(N, H, H, C, C, 3, 3, 1, 1)
    slots = query("primia_conv_dgrad_masked_acc_bnsums_slots", desc, dt)
    assert slots > 0 and (blocks == 0 or slots <= blocks)
    assert query("primia_conv_dgrad_masked_acc_bnsums_slots", ConvDesc.make(N, H, H, 128, 128, 3, 3, 1, 1), dt) == 0
    _, wd = prep_weights(desc, rnd(torch.randn(C, C, 3, 3, generator=g) * 0.05, dtype), dtype, cuda, C)
    M = N * H * H
    dy1 = (torch.randn(M, C, generator=g) * 0.3).to(dtype).to(cuda)
    base = (torch.randn(M, C, generator=g)).to(dtype).to(cuda)                 # gradient of the block's output
    acc_mask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.int32).to(torch.uint8).to(cuda)
    y = (torch.randn(M, C, generator=g) * 1.2 - 0.1).to(dtype).to(cuda)        # the residual BatchNorm's input
    mask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.int32).to(torch.uint8).to(cuda)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda)
    mean = y.float().mean(0)
    invstd = 1.0 / torch.sqrt(y.float().var(0, unbiased=False) + 1e-5)
    # the chain
    dx_a = base.clone()
    call("primia_conv2d_dgrad_masked_acc", desc, dy1, wd, dx_a, acc_mask, dt)
    ws = torch.zeros(query("primia_bn_workspace_bytes", M, C), dtype=torch.uint8, device=cuda)
    dy_a, dg_a, db_a = torch.empty_like(dx_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_bwd_mask", y, mask, dx_a, dy_a, None, gamma, mean, invstd, dg_a, db_a, M, C, ws, ws.numel(), dt)
    # fused
    dx_b = base.clone()
    sums = torch.full((slots, 2, C), 5.0, device=cuda)
    call("primia_conv2d_dgrad_masked_acc_bnsums", desc, dy1, wd, dx_b, acc_mask, 2, y, mask, mean, invstd, sums, dt)
    dy_b, dg_b, db_b = torch.empty_like(dx_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_bwd_mask_from_sums", y, mask, dx_b, dy_b, None, gamma, mean, invstd, dg_b, db_b, sums, slots, M, C, dt)
    assert torch.equal(dx_a, dx_b)
    assert relerr(dg_b, dg_a) < 2e-5 and relerr(db_b, db_a) < 2e-5
    assert relerr(dy_b.float(), dy_a.float()) < 2e-3
    dx_c, sums2 = base.clone(), torch.empty_like(sums)
    call("primia_conv2d_dgrad_masked_acc_bnsums", desc, dy1, wd, dx_c, acc_mask, 2, y, mask, mean, invstd, sums2, dt)
    assert torch.equal(sums, sums2) and torch.equal(dx_c, dx_b)
    with pytest.raises(_lib.PrimiaError):
        call("primia_conv2d_dgrad_masked_acc_bnsums", desc, dy1, wd, dx_c, acc_mask, 2, y, None, mean, invstd, sums2, dt)
    with pytest.raises(_lib.PrimiaError):
        call("primia_conv2d_dgrad_masked_acc_bnsums", desc, dy1, wd, dx_c, acc_mask, 1, y, mask, mean, invstd, sums2, dt)


@pytest.mark.parametrize("N,Hs,blocks", [(2, 32, 0), (1, 112, 0), (3, 24, 0), (2, 18, 0), (2, 112, 4), (3, 48, 2), (2, 18, 2)])
def test_masked_accumulate_dgrad_with_stem_bn_backward_sums(cuda, c64_blocks, N, Hs, blocks):
    """primia_conv2d_dgrad_masked_acc_bnsums, mode 3 (conv3x3_c64_kernel<true, 3, 3>): layer1.0.conv1's accumulating data
    gradient writes the max-pool's output gradient and forms the stem BatchNorm's backward sums at pooled resolution, +
    primia_bn_relu_maxpool_bwd_from_sums, against primia_conv2d_dgrad_masked_acc followed by primia_bn_relu_maxpool_bwd(dy = NULL)
    — its PoolScatterFn reduction pass: dx bit-identical, dgamma / dbeta to summation order — one channel with gamma == 0
    and one with gamma = 1e-6 against beta = 0.3 (a pretrained bn1 has such channels): xhat = (p - beta) / gamma
    from the STORED p would be bf16 rounding noise / gamma there, both are served from y at the argmax by the finalize
    kernel.  dgamma / dbeta are also held to the float64 sums over (dpooled, y at the argmax): the two special channels
    exactly (1e-4), the others within the rounding of the stored p (5e-2)."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    C = 64
    g = torch.Generator().manual_seed(5 * N + Hs)
    Ho = (Hs - 1) // 2 + 1
    Ms, M = N * Hs * Hs, N * Ho * Ho
    ystem = to_nhwc(rnd(torch.randn(N, C, Hs, Hs, generator=g) * 2 + 0.3, dtype), dtype, cuda)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda)
    beta = (torch.randn(C, generator=g) * 0.5).to(cuda)
    gamma[5], beta[5] = 0.0, 0.7
    gamma[9] = -0.8
    gamma[17], beta[17] = 1e-6, 0.3
    gamma[23], beta[23] = -2e-4, 0.25
    ws = torch.zeros(query("primia_bn_workspace_bytes", Ms, C), dtype=torch.uint8, device=cuda)
    rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    sm, si = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    pool = torch.empty(M, C, dtype=dtype, device=cuda)
    am = torch.empty(M, C, dtype=torch.uint8, device=cuda)
    call("primia_bn_relu_maxpool_fwd", ystem, pool, am, gamma, beta, rm, rv, sm, si, N, Hs, Hs, C, 1e-5, 0.1, ws, ws.numel(), dt)
    c64_blocks(blocks)
    desc = ConvDesc.make(N, Ho, Ho, C, C, 3, 3, 1, 1)
    slots = query("primia_conv_dgrad_masked_acc_bnsums_slots", desc, dt)
    assert slots > 0
    _, wd = prep_weights(desc, rnd(torch.randn(C, C, 3, 3, generator=g) * 0.05, dtype), dtype, cuda, C)
    dy1 = (torch.randn(M, C, generator=g) * 0.3).to(dtype).to(cuda)
    base = (torch.randn(M, C, generator=g)).to(dtype).to(cuda)
    acc_mask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.int32).to(torch.uint8).to(cuda)
    # the chain
    dp_a = base.clone()
    call("primia_conv2d_dgrad_masked_acc", desc, dy1, wd, dp_a, acc_mask, dt)
    dg_a, db_a = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_relu_maxpool_bwd", ystem, pool, dp_a, am, None, gamma, beta, sm, si, dg_a, db_a, N, Hs, Hs, C, ws,
         ws.numel(), dt)
    # fused
    dp_b = base.clone()
    sums = torch.full((slots, 2, C), 5.0, device=cuda)
    call("primia_conv2d_dgrad_masked_acc_bnsums", desc, dy1, wd, dp_b, acc_mask, 3, pool, None, beta, gamma, sums, dt)
    dg_b, db_b = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_relu_maxpool_bwd_from_sums", ystem, pool, dp_b, am, gamma, beta, sm, si, dg_b, db_b, sums, slots, N, Hs, Hs,
         C, dt)
    assert torch.equal(dp_a, dp_b)
    assert relerr(db_b, db_a) < 2e-5 and relerr(dg_b, dg_a) < 2e-5
    assert abs(float(dg_b[5]) - float(dg_a[5])) <= 2e-5 * max(1.0, abs(float(dg_a[5])))
    # float64 reference from the tensors the kernels read: g = dpooled * [p > 0] lands on y's argmax element
    pv, dpv, code = pool.double().cpu().view(N, Ho, Ho, C), dp_b.double().cpu().view(N, Ho, Ho, C), am.cpu().view(N, Ho, Ho, C).long()
    yv = ystem.double().cpu().view(N, Hs, Hs, C)
    hh = (2 * torch.arange(Ho).view(1, Ho, 1, 1) - 1 + code // 3).clamp(0, Hs - 1)
    ww = (2 * torch.arange(Ho).view(1, 1, Ho, 1) - 1 + code % 3).clamp(0, Hs - 1)
    nn_, cc = torch.arange(N).view(N, 1, 1, 1).expand_as(code), torch.arange(C).view(1, 1, 1, C).expand_as(code)
    xh = (yv[nn_, hh, ww, cc] - sm.double().cpu()) * si.double().cpu()
    gg = dpv * (pv > 0)
    db_ref, dg_ref = gg.sum((0, 1, 2)), (gg * xh).sum((0, 1, 2))
    assert relerr(db_b.double().cpu(), db_ref) < 1e-5
    for c in (5, 17, 23):       # served from y: exact up to summation order
        assert abs(float(dg_b[c]) - float(dg_ref[c])) <= 1e-4 * max(1.0, abs(float(dg_ref[c]))), (c, float(dg_b[c]), float(dg_ref[c]))
    assert relerr(dg_b.double().cpu(), dg_ref) < 5e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,C", [(3, 9, 128), (2, 14, 256), (5, 7, 512), (4, 28, 128)])
def test_batchnorm_backward_pair_is_bit_identical(cuda, dtype, N, H, C):
    """primia_bn_bwd_pair == primia_bn_bwd_mask(y2, g_out = g) followed by primia_bn_bwd(yd, dz = g, relu = 0)."""
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(H + C)
    M = N * H * H
    mk = lambda s=1.0, b=0.0: to_nhwc(rnd(torch.randn(N, C, H, H, generator=g) * s + b, dtype), dtype, cuda)
    y2, yd, res, dz = mk(1.3, 0.2), mk(0.7, -0.1), mk(), mk()
    ws_bytes = 1024 * 3 * C * 4
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    gam2, bet2 = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    gamd, betd = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    sm2, si2, smd, sid = (torch.empty(C, device=cuda) for _ in range(4))
    z, idn = torch.empty_like(y2), torch.empty_like(y2)
    ch = 4 if dtype == torch.float32 else 8
    mask = torch.empty(M * C // ch, dtype=torch.uint8, device=cuda)
    rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    call("primia_bn_fwd_train", yd, None, idn, gamd, betd, rm.clone(), rv.clone(), smd, sid, M, C, 1e-5, 0.1, 0, ws, ws_bytes, dt)
    call("primia_bn_fwd_train_mask", y2, idn, z, mask, gam2, bet2, rm, rv, sm2, si2, None, 0, M, C, 1e-5, 0.1, ws, ws_bytes, dt)
    # separate
    dy2a, ga, dyda = torch.empty_like(y2), torch.empty_like(y2), torch.empty_like(y2)
    d2a, b2a, dda, bda = (torch.empty(C, device=cuda) for _ in range(4))
    call("primia_bn_bwd_mask", y2, mask, dz, dy2a, ga, gam2, sm2, si2, d2a, b2a, M, C, ws, ws_bytes, dt)
    call("primia_bn_bwd", yd, None, ga, dyda, None, gamd, smd, sid, dda, bda, M, C, 0, ws, ws_bytes, dt)
    # fused
    dy2b, dydb = torch.empty_like(y2), torch.empty_like(y2)
    d2b, b2b, ddb, bdb = (torch.empty(C, device=cuda) for _ in range(4))
    call("primia_bn_bwd_pair", y2, yd, dz, mask, dy2b, dydb, gam2, sm2, si2, gamd, smd, sid, d2b, b2b, ddb, bdb, M, C, ws,
         ws_bytes, dt)
    for a, b in ((dy2a, dy2b), (dyda, dydb), (d2a, d2b), (b2a, b2b), (dda, ddb), (bda, bdb)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,H,C", [(3, 9, 128), (2, 14, 256), (4, 28, 128)])
def test_batchnorm_forward_pair_is_bit_identical(cuda, dtype, N, H, C):
    """primia_bn_fwd_train_pair == primia_bn_fwd_train(yd -> idn, relu = 0) + primia_bn_fwd_train_mask(y2, idn)."""
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(2 * H + C)
    M = N * H * H
    mk = lambda s=1.0, b=0.0: to_nhwc(rnd(torch.randn(N, C, H, H, generator=g) * s + b, dtype), dtype, cuda)
    y2, yd = mk(1.3, 0.2), mk(0.7, -0.1)
    ws_bytes = query("primia_bn_workspace_bytes", M, C)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=cuda)
    gam2, bet2 = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    gamd, betd = (torch.rand(C, generator=g) + 0.5).to(cuda), torch.randn(C, generator=g).to(cuda)
    ch = 4 if dtype == torch.float32 else 8
    outs = []
    for fused in (False, True):
        sm2, si2, smd, sid = (torch.empty(C, device=cuda) for _ in range(4))
        rm2, rv2, rmd, rvd = torch.zeros(C, device=cuda), torch.ones(C, device=cuda), torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
        z = torch.empty_like(y2)
        mask = torch.empty(M * C // ch, dtype=torch.uint8, device=cuda)
        if fused:
            call("primia_bn_fwd_train_pair", y2, yd, z, mask, gam2, bet2, rm2, rv2, sm2, si2, None, 0, gamd, betd, rmd, rvd,
                 smd, sid, None, 0, M, C, 1e-5, 0.1, ws, ws_bytes, dt)
        else:
            idn = torch.empty_like(y2)
            call("primia_bn_fwd_train", yd, None, idn, gamd, betd, rmd, rvd, smd, sid, M, C, 1e-5, 0.1, 0, ws, ws_bytes, dt)
            call("primia_bn_fwd_train_mask", y2, idn, z, mask, gam2, bet2, rm2, rv2, sm2, si2, None, 0, M, C, 1e-5, 0.1, ws,
                 ws_bytes, dt)
        outs.append((z, mask, sm2, si2, smd, sid, rm2, rv2, rmd, rvd))
    for a, b in zip(*outs):
        assert torch.equal(a, b)




S2LH_CASES = [
    (2, 16, 64, 128),     # layer2.0 shape: dx has 64 channels -> two parity classes per tile (half-zero weight tiles)
    (3, 28, 128, 256),    # layer3.0: one class per tile, ragged last pixel tile (588 = 3 x 192 + 12)
    (4, 14, 256, 512),    # layer4.0: 4 channel tiles, 36-step programs
    (1, 6, 64, 128),      # 3 x 3 parity grid: every shifted tap leaves the image somewhere
    (5, 4, 128, 128),     # 2 x 2 parity grid, 20 parity pixels in all
    (5, 2, 128, 128),     # 1 x 1 parity grid: not served (falls back)
    (2, 56, 64, 128),     # widest grid the kernel takes (Wo = 28), 1,568 parity pixels = 8.17 tiles
    (1, 58, 64, 128),     # Wo = 29: not served, falls back to the implicit GEMM
]


@pytest.mark.parametrize("N,H,C,K", S2LH_CASES)
def test_parity_plane_kernel_every_mode(cuda, N, H, C, K):
    """conv_s2lh_kernel (csrc/conv_s2lh.hip) — the transition blocks' stride-2 3x3 + 1x1 layers as stride-1 problems on
    parity planes — in EVERY form it has, forced with option s2lh = 7 (by default only the data gradient of the narrow
    layers takes it): conv1 alone, downsample alone, the paired forward with BatchNorm partial sums, the data gradient
    alone and paired, each against torch autograd (torchlib/models.py:219-235, 433-436) at one bf16 rounding, and the
    partial sums against the sums of the stored values."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(5 + H + C)
    x = rnd(torch.randn(N, C, H, H, generator=g).clamp_min(0), dtype)
    w1 = rnd(torch.randn(K, C, 3, 3, generator=g) * 0.05, dtype)
    wd_ = rnd(torch.randn(K, C, 1, 1, generator=g) * 0.1, dtype)
    d1 = ConvDesc.make(N, H, H, C, K, 3, 3, 2, 1)
    dd = ConvDesc.make(N, H, H, C, K, 1, 1, 2, 0)
    xr = x.clone().requires_grad_(True)
    y1_ref = F.conv2d(xr, w1, None, 2, 1)
    yd_ref = F.conv2d(xr, wd_, None, 2, 0)
    dy1 = rnd(torch.randn(y1_ref.shape, generator=g), dtype)
    dyd = rnd(torch.randn(yd_ref.shape, generator=g), dtype)
    g1 = torch.autograd.grad(y1_ref, xr, dy1, retain_graph=True)[0]
    gd = torch.autograd.grad(yd_ref, xr, dyd)[0]
    try:
        _lib.set_option("s2lh", 7)
        served = 2 <= H // 2 <= 28
        assert query("primia_conv_kernel_id", d1, 0, dt) == (5 if served else 1)
        assert query("primia_conv_kernel_id", d1, 1, dt) == (5 if served else 1)
        w1f, w1d = prep_weights(d1, w1, dtype, cuda, C)
        wdf, wdd = prep_weights(dd, wd_, dtype, cuda, C)
        xd = to_nhwc(x, dtype, cuda)
        M2 = N * d1.Ho * d1.Wo
        # single launches
        y1 = torch.full((M2, K), 7.0, dtype=dtype, device=cuda)
        yd = torch.full((M2, K), 7.0, dtype=dtype, device=cuda)
        call("primia_conv2d_fwd", d1, xd, w1f, y1, dt)
        call("primia_conv2d_fwd", dd, xd, wdf, yd, dt)
        assert relerr(from_nhwc(y1, N, d1.Ho, d1.Wo), y1_ref.detach()) < tol(dtype)
        assert relerr(from_nhwc(yd, N, d1.Ho, d1.Wo), yd_ref.detach()) < tol(dtype)
        # the pair, with partial sums
        s1, sd = query("primia_conv_stat_slots_for", d1, dt), query("primia_conv_stat_slots_for", dd, dt)
        assert s1 == sd and (not served or s1 == (M2 + 191) // 192)
        q1 = torch.full((s1, 2, K), 9.0, device=cuda)
        qd = torch.full((sd, 2, K), 9.0, device=cuda)
        p1, pd = torch.empty_like(y1), torch.empty_like(yd)
        call("primia_conv2d_fwd_stats_pair", d1, xd, w1f, p1, q1, dd, wdf, pd, qd, dt)
        assert torch.equal(p1, y1) and torch.equal(pd, yd)
        if query("primia_conv_stats_per_tile", d1, dt) == 1:
            for q, y in ((q1, p1), (qd, pd)):
                v = y.double()
                assert relerr(q[:, 0].double().sum(0), v.sum(0)) < 1e-5
                assert relerr(q[:, 1].double().sum(0), (v * v).sum(0)) < 1e-5
        # data gradients: conv1 alone, then the pair (every element of dx written)
        dy1d, dydd = to_nhwc(dy1, dtype, cuda), to_nhwc(dyd, dtype, cuda)
        dx = torch.full((N * H * H, C), 3.0, dtype=dtype, device=cuda)
        call("primia_conv2d_dgrad", d1, dy1d, w1d, dx, 0, dt)
        assert relerr(from_nhwc(dx, N, H, H), g1) < tol(dtype)
        dx.fill_(3.0)
        call("primia_conv2d_dgrad_pair", d1, dy1d, w1d, dd, dydd, wdd, dx, dt)
        assert relerr(from_nhwc(dx, N, H, H), g1 + gd) < tol(dtype)
        # ... and bit-repeatable
        dx2 = torch.empty_like(dx)
        call("primia_conv2d_dgrad_pair", d1, dy1d, w1d, dd, dydd, wdd, dx2, dt)
        assert torch.equal(dx, dx2)
        # the implicit GEMM on the same operands: one bf16 rounding apart at most
        _lib.set_option("s2lh", 0)
        assert query("primia_conv_kernel_id", d1, 1, dt) == 1
        dx3 = torch.empty_like(dx)
        call("primia_conv2d_dgrad_pair", d1, dy1d, w1d, dd, dydd, wdd, dx3, dt)
        assert relerr(dx, dx3) < 4e-3
    finally:
        _lib.set_option("s2lh", 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C,mask", [(802816 // 16, 64, False), (12544, 512, True), (3000, 128, True), (64, 256, False)])
def test_bn_apply_with_inline_finalize_is_bit_identical(cuda, dtype, M, C, mask):
    """primia_bn_fwd_train_apply_inline — the finalize launch folded into the apply kernel (its first C / 4 blocks combine the
    partial sums and raise one flag each, every block waits for its channels' flags) — against the two-launch forms
    primia_bn_fwd_train_from_sums / primia_bn_fwd_train_mask on the same partial sums: z, the ReLU mask, mean, invstd and the
    running statistics bit for bit; (64, 256): fewer apply blocks than finalize slices, the entry point falls back."""
    g = torch.Generator().manual_seed(M + C)
    dt = _lib.dtype_code(dtype)
    y = (torch.randn(M, C, generator=g) * 1.5 + 0.3).to(dtype).to(cuda)
    res = torch.randn(M, C, generator=g).to(dtype).to(cuda)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda)
    beta = torch.randn(C, generator=g).to(cuda)
    slots = 37
    # partial sums of a ragged 37-way row split, as a conv epilogue would leave them
    sums = torch.zeros(slots, 2, C, device=cuda)
    yf = y.float()
    bounds = torch.linspace(0, M, slots + 1).long().tolist()
    for i in range(slots):
        blk = yf[bounds[i]:bounds[i + 1]]
        sums[i, 0], sums[i, 1] = blk.sum(0), (blk * blk).sum(0)

    def run(inline):
        z = torch.empty_like(y)
        mk = torch.zeros(M * C * y.element_size() // 16, dtype=torch.uint8, device=cuda) if mask else None
        rm, rv = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
        sm, si = torch.empty(C, device=cuda), torch.empty(C, device=cuda)
        if inline:
            flags = torch.zeros(128, dtype=torch.int32, device=cuda)
            call("primia_bn_fwd_train_apply_inline", y, res, z, mk, gamma, beta, rm, rv, sm, si, sums, slots, M, C, 1e-5, 0.1, 1,
                 flags, dt)
        elif mask:
            ws = torch.zeros(query("primia_bn_workspace_bytes", M, C), dtype=torch.uint8, device=cuda)
            call("primia_bn_fwd_train_mask", y, res, z, mk, gamma, beta, rm, rv, sm, si, sums, slots, M, C, 1e-5, 0.1, ws,
                 ws.numel(), dt)
        else:
            call("primia_bn_fwd_train_from_sums", y, res, z, gamma, beta, rm, rv, sm, si, sums, slots, M, C, 1e-5, 0.1, 1, dt)
        torch.cuda.synchronize()
        return z, mk, rm, rv, sm, si

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        if u is not None:
            assert torch.equal(u, v)
    # and it is BatchNorm: against torch on the stored values
    ref = torch.nn.functional.batch_norm(yf.cpu(), None, None, gamma.cpu(), beta.cpu(), True, 0.1, 1e-5)
    ref = torch.relu(ref + res.float().cpu())
    assert relerr(a[0].float().cpu(), ref) < (2e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("N,H,C,K", [(4, 28, 128, 128), (7, 9, 256, 192), (6, 7, 512, 512), (3, 14, 256, 256),
                                     (2, 56, 64, 64), (3, 12, 64, 64), (5, 8, 64, 64)])     # (64 -> 64: conv3x3_c64_kernel)
def test_dgrad_with_bn_backward_sums_matches_the_separate_reduction(cuda, N, H, C, K):
    """primia_conv2d_dgrad_bnsums — the data gradient of a 3x3 / stride-1 layer whose write-back also forms the two sums the
    BatchNorm backward of the layer in front of it needs — + primia_bn_relu_bwd_from_sums, against the chain they replace
    (primia_conv2d_dgrad, then primia_bn_relu_bwd with its own reduction pass): dx bit-identical, dgamma / dbeta / dy to fp32
    summation order (the partials are per conv tile instead of per row slab), and against torch autograd of
    relu(batch_norm(y)) -> conv (torchlib/models.py:268-284)."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(N + H + C)
    desc = ConvDesc.make(N, H, H, C, K, 3, 3, 1, 1)
    slots = query("primia_conv_dgrad_bnsums_slots", desc, dt)
    assert slots > 0
    M = N * H * H
    y = (torch.randn(M, C, generator=g) * 1.3 + 0.2).to(dtype).to(cuda)           # the BatchNorm's input (conv1's output)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda)
    beta = (torch.randn(C, generator=g) * 0.3).to(cuda)
    mean = y.float().mean(0)
    invstd = 1.0 / torch.sqrt(y.float().var(0, unbiased=False) + 1e-5)
    w = rnd(torch.randn(K, C, 3, 3, generator=g) * 0.05, dtype)
    _, wd = prep_weights(desc, w, dtype, cuda, C)
    dy2 = (torch.randn(M, K, generator=g) * 0.1).to(dtype).to(cuda)               # gradient w.r.t. conv2's output
    # the chain
    dz_a = torch.empty(M, C, dtype=dtype, device=cuda)
    call("primia_conv2d_dgrad", desc, dy2, wd, dz_a, 0, dt)
    ws = torch.zeros(query("primia_bn_workspace_bytes", M, C), dtype=torch.uint8, device=cuda)
    dy_a, dg_a, db_a = torch.empty_like(dz_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_relu_bwd", y, dz_a, dy_a, gamma, beta, mean, invstd, dg_a, db_a, M, C, ws, ws.numel(), dt)
    # fused
    dz_b = torch.empty_like(dz_a)
    sums = torch.full((slots, 2, C), 7.0, device=cuda)
    call("primia_conv2d_dgrad_bnsums", desc, dy2, wd, dz_b, y, mean, invstd, gamma, beta, sums, dt)
    dy_b, dg_b, db_b = torch.empty_like(dz_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_relu_bwd_from_sums", y, dz_b, dy_b, gamma, beta, mean, invstd, dg_b, db_b, sums, slots, M, C, dt)
    assert torch.equal(dz_a, dz_b)
    assert relerr(dg_b, dg_a) < 2e-5 and relerr(db_b, db_a) < 2e-5
    assert relerr(dy_b.float(), dy_a.float()) < 2e-3        # (dy is rounded to bf16 after constants that differ in the last bits)
    # run to run
    sums2 = torch.empty_like(sums)
    call("primia_conv2d_dgrad_bnsums", desc, dy2, wd, dz_b, y, mean, invstd, gamma, beta, sums2, dt)
    assert torch.equal(sums, sums2)
    # and it is the gradient: dgamma / dbeta from torch on the stored tensors
    yy = y.float().cpu().requires_grad_(True)
    gg, bb = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True)
    z = torch.relu(torch.nn.functional.batch_norm(yy, None, None, gg, bb, True, 0.1, 1e-5))
    z.backward(dz_a.float().cpu())
    assert relerr(dg_b.cpu(), gg.grad) < 1e-3 and relerr(db_b.cpu(), bb.grad) < 1e-3


@pytest.mark.parametrize("N,H,C", [(3, 16, 64), (2, 56, 64), (5, 12, 64), (3, 28, 128), (5, 14, 256), (2, 10, 128)])
def test_transition_dgrad_pair_with_bn_backward_sums(cuda, N, H, C):
    """primia_conv2d_dgrad_pair_bnsums (conv_s2lh_kernel for 64-channel dx, conv_igemm_kernel's parity-class walk for the wider
    ones): the paired data gradient of a transition block whose
    write-back also forms the backward sums of the residual BatchNorm in front of the block, + primia_bn_bwd_mask_from_sums,
    against primia_conv2d_dgrad_pair followed by primia_bn_bwd_mask with its own reduction pass: dx bit-identical, dgamma / dbeta
    to summation order, dy to one bf16 rounding; partials bit-repeatable."""
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    K = 2 * C
    g = torch.Generator().manual_seed(N * H)
    d1 = ConvDesc.make(N, H, H, C, K, 3, 3, 2, 1)
    dd = ConvDesc.make(N, H, H, C, K, 1, 1, 2, 0)
    slots = query("primia_conv_dgrad_pair_bnsums_slots", d1, dt)
    assert slots == (2 * ((N * (H // 2) ** 2 + 191) // 192) if C == 64 else 4 * ((N * (H // 2) ** 2 + 127) // 128))
    _, w1d = prep_weights(d1, rnd(torch.randn(K, C, 3, 3, generator=g) * 0.05, dtype), dtype, cuda, C)
    _, wdd = prep_weights(dd, rnd(torch.randn(K, C, 1, 1, generator=g) * 0.1, dtype), dtype, cuda, C)
    M2, M = N * (H // 2) ** 2, N * H * H
    dy1 = (torch.randn(M2, K, generator=g) * 0.1).to(dtype).to(cuda)
    dyd = (torch.randn(M2, K, generator=g) * 0.1).to(dtype).to(cuda)
    y = (torch.randn(M, C, generator=g) * 1.2 - 0.1).to(dtype).to(cuda)          # the residual BatchNorm's input
    mask = torch.randint(0, 256, (M * C // 8,), generator=g, dtype=torch.uint8).to(cuda)
    gamma = (torch.rand(C, generator=g) + 0.5).to(cuda)
    mean = y.float().mean(0)
    invstd = 1.0 / torch.sqrt(y.float().var(0, unbiased=False) + 1e-5)
    # the chain
    dx_a = torch.empty(M, C, dtype=dtype, device=cuda)
    call("primia_conv2d_dgrad_pair", d1, dy1, w1d, dd, dyd, wdd, dx_a, dt)
    ws = torch.zeros(query("primia_bn_workspace_bytes", M, C), dtype=torch.uint8, device=cuda)
    dy_a, dg_a, db_a = torch.empty_like(dx_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_bwd_mask", y, mask, dx_a, dy_a, None, gamma, mean, invstd, dg_a, db_a, M, C, ws, ws.numel(), dt)
    # fused
    dx_b = torch.empty_like(dx_a)
    sums = torch.full((slots, 2, C), 5.0, device=cuda)
    call("primia_conv2d_dgrad_pair_bnsums", d1, dy1, w1d, dd, dyd, wdd, dx_b, y, mask, mean, invstd, sums, dt)
    dy_b, dg_b, db_b = torch.empty_like(dx_a), torch.empty(C, device=cuda), torch.empty(C, device=cuda)
    call("primia_bn_bwd_mask_from_sums", y, mask, dx_b, dy_b, None, gamma, mean, invstd, dg_b, db_b, sums, slots, M, C, dt)
    assert torch.equal(dx_a, dx_b)
    assert relerr(dg_b, dg_a) < 2e-5 and relerr(db_b, db_a) < 2e-5
    assert relerr(dy_b.float(), dy_a.float()) < 2e-3
    sums2 = torch.empty_like(sums)
    call("primia_conv2d_dgrad_pair_bnsums", d1, dy1, w1d, dd, dyd, wdd, dx_b, y, mask, mean, invstd, sums2, dt)
    assert torch.equal(sums, sums2)
