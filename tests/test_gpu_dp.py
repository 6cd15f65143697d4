"""GPU: GroupNorm network and the DP-SGD gradient (BASELINE configs[3]) against
the oracle.  pytorch-dp is not in the reference tree, so the clip / noise rule is 'parity unpinned' (the oracle
restates the documented algorithm: per-sample clip to C = 1.0, noise multiplier 1.3); the per-sample gradients under
it are pinned to the reference's model class (tests/golden/dp_ref.npz, tests/test_oracle_train.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import train_oracle as O  # noqa: E402
from primia_amd import _lib, resnet_spec as rs  # noqa: E402
from primia_amd._lib import call, query  # noqa: E402
from primia_amd.engine import ResNet18Engine  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,relu,res", [(64, 1, 0), (128, 1, 1), (512, 0, 0), (256, 1, 1)])
@pytest.mark.parametrize("N,H", [(3, 6), (130, 5)])
def test_groupnorm_kernels(cuda, dtype, C, relu, res, N, H):
    """N = 3: the reductions run over pixel slabs + a finalize launch; N = 130 (>= 128): one block per sample with
    the finalize folded in (gn_sample_reduce_kernel)."""
    G = 32
    HW = H * H
    g = torch.Generator().manual_seed(C)
    rnd = lambda t: t.to(dtype).float()
    y = rnd(torch.randn(N, C, H, H, generator=g) * 2 + 0.3)
    r = rnd(torch.randn(N, C, H, H, generator=g)) if res else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    yr = y.clone().requires_grad_(True)
    zr = F.group_norm(yr, G, gamma, beta, 1e-5)
    if res:
        zr = zr + r
    if relu:
        zr = F.relu(zr)
    dz = rnd(torch.randn(zr.shape, generator=g))
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(N * HW, C).contiguous().to(dtype).to(cuda)
    back = lambda t: t.float().cpu().view(N, H, H, C).permute(0, 3, 1, 2)
    dt = _lib.dtype_code(dtype)
    wsb = query("primia_gn_workspace_bytes", N, C, G)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=cuda)
    z = torch.empty(N * HW, C, dtype=dtype, device=cuda)
    sm, si = torch.empty(N * G, device=cuda), torch.empty(N * G, device=cuda)
    call("primia_gn_fwd", nhwc(y), nhwc(r) if res else None, z, gamma.to(cuda), beta.to(cuda), sm, si, N, HW, C, G, 1e-5,
         relu, ws, wsb, dt)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel(back(z), zr.detach()) < tol
    # backward incl. per-sample affine gradients
    per_g, per_b = [], []
    for n in range(N):
        yn = y[n:n + 1].clone().requires_grad_(True)
        gn_, bn_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        zn = F.group_norm(yn, G, gn_, bn_, 1e-5)
        if res:
            zn = zn + r[n:n + 1]
        if relu:
            zn = F.relu(zn)
        zn.backward(dz[n:n + 1])
        per_g.append(gn_.grad)
        per_b.append(bn_.grad)
    zr.backward(dz)
    dy = torch.empty_like(z)
    psg, psb = torch.empty(N, C, device=cuda), torch.empty(N, C, device=cuda)
    call("primia_gn_bwd", nhwc(y), z if relu else None, nhwc(dz), dy, None, gamma.to(cuda), sm, si, psg, psb, N, HW, C, G,
         relu, ws, wsb, dt)
    assert rel(back(dy), yr.grad) < (2e-5 if dtype == torch.float32 else 2e-2)
    assert rel(psg, torch.stack(per_g)) < (2e-5 if dtype == torch.float32 else 1e-2)
    assert rel(psb, torch.stack(per_b)) < (2e-5 if dtype == torch.float32 else 1e-2)
    if relu and not res:
        # the same backward with the ReLU mask recomputed from y instead of read from z: identical bits
        dy2 = torch.empty_like(z)
        psg2, psb2 = torch.empty(N, C, device=cuda), torch.empty(N, C, device=cuda)
        call("primia_gn_relu_bwd", nhwc(y), nhwc(dz), dy2, gamma.to(cuda), beta.to(cuda), sm, si, psg2, psb2, N, HW, C, G,
             ws, wsb, dt)
        assert torch.equal(dy2, dy) and torch.equal(psg2, psg) and torch.equal(psb2, psb)


def test_dp_sgd_gradient_matches_oracle(cuda):
    """fp32 engine, GroupNorm ResNet-18, batch 4 at 64x64: per-sample norms, clip factors and the
    noised clipped gradient vs the oracle's per-sample (batch-of-1) restatement."""
    batch, size = 4, 64
    torch.manual_seed(9)
    spec = rs.resnet18_spec(3, 3, size, "max")
    sd = rs.init_state_dict(spec, "group")
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda, norm="group")
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, 3, (batch,), generator=g)
    noise_flat = torch.randn(eng.P, generator=g)
    noise, off = {}, 0
    for k, s in eng.p_entries:
        n = int(torch.Size(s).numel())
        noise[k] = noise_flat[off:off + n].view(s)
        off += n
    # plain (non-private) step on the GroupNorm network first
    logits = eng.forward(x.to(cuda)).cpu()
    eng.loss_backward(y.to(cuda))
    ologits, _, ograds = O.train_step({k: v.clone() for k, v in sd.items()}, x, y, 0.0, 0.0)
    assert rel(logits, ologits) < 1e-5
    for k in ("conv1.weight", "bn1.weight", "layer3.0.downsample.0.weight", "layer4.1.bn2.bias", "fc.weight"):
        assert rel(eng.gviews[k], ograds[k]) < 1e-2, k
    # DP-SGD gradient
    eng.forward(x.to(cuda))
    eng.dp_loss_backward(y.to(cuda), max_grad_norm=1.0, noise_multiplier=1.3, noise=noise_flat.to(cuda))
    want, norms, clip = O.dp_gradients({k: v.clone() for k, v in sd.items()}, x, y, 1.0, 1.3, noise)
    got_norms = eng.dp_stats["sq_norms"].sqrt().cpu()
    assert torch.allclose(got_norms, norms, rtol=5e-3), (got_norms, norms)
    assert torch.allclose(eng.dp_stats["clip"].cpu().double(), clip, rtol=5e-3)
    assert (clip < 1).any(), "test should exercise clipping"
    for k, _ in eng.p_entries:
        # noise (std 1.3/4 per element) dominates most tensors: compare on the full vector
        assert rel(eng.gviews[k], want[k]) < 5e-3, k
    # without noise the result is the mean of clipped per-sample gradients: norm <= C
    eng.forward(x.to(cuda))
    eng.dp_loss_backward(y.to(cuda), 1.0, 0.0, noise=torch.zeros(eng.P, device=cuda))
    assert eng.grads.double().norm().item() <= 1.0 + 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(3, 16, 64, 64, 3, 1, 1), (2, 16, 64, 128, 3, 2, 1), (3, 8, 256, 256, 3, 1, 1),
                                  (2, 8, 256, 512, 1, 2, 0), (3, 32, 4, 64, 7, 2, 3), (5, 14, 128, 128, 3, 1, 1),
                                  # enough (image pairs x slabs) for the whole-images-per-half norm pass of the patch
                                  # kernel: 4 sub-patches per image and an odd batch; one sub-patch per image; ragged 12x12
                                  (33, 16, 256, 256, 3, 1, 1), (40, 7, 512, 256, 3, 1, 1), (17, 12, 256, 512, 3, 1, 1),
                                  # 7x7 images: "ghost" norms from two Gram matrices per sample (dp_ghost.hip)
                                  (12, 7, 512, 512, 3, 1, 1), (5, 7, 64, 128, 3, 1, 1), (7, 7, 256, 64, 3, 1, 1),
                                  (11, 14, 256, 512, 3, 2, 1), (3, 14, 64, 64, 3, 2, 1), (9, 14, 256, 256, 3, 1, 1),
                                  # whole images per block in the per-tap kernel (stride-2 / 1x1 layers): 784 / 196 / 49
                                  # pixels per image (13 / 4 / 1 stages, the last one ragged), several images per block
                                  (70, 56, 64, 128, 3, 2, 1), (37, 28, 128, 256, 3, 2, 1), (9, 14, 256, 512, 3, 2, 1),
                                  (300, 14, 256, 512, 1, 2, 0), (5, 28, 128, 256, 1, 2, 0)])
def test_persample_sqnorm_matches_slab_norms(cuda, dtype, case):
    """primia_conv2d_wgrad_persample_sqnorm (norms summed inside the wgrad kernels) against the explicit
    per-sample slabs of primia_conv2d_wgrad_persample + primia_persample_sqnorm, for every kernel family
    (patch, per-tap, LDS-DMA, stem)."""
    from primia_amd._lib import ConvDesc

    N, H, C, K, R, s, p = case
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(H * C + K)
    d = ConvDesc.make(N, H, H, C, K, R, R, s, p)
    x = (torch.randn(N * H * H, C, generator=g)).to(dtype).to(cuda)
    if R == 7:
        x[:, 3] = 0  # 4th stem channel is padding
    dy = torch.randn(N * d.Ho * d.Wo, K, generator=g).to(dtype).to(cuda)
    n = query("primia_conv_wfwd_elems", d)
    slab = torch.zeros(N, n, device=cuda)
    call("primia_conv2d_wgrad_persample", d, x, dy, slab, dt)
    sq_ref = torch.zeros(N, dtype=torch.float64, device=cuda)
    call("primia_persample_sqnorm", slab, N, n, sq_ref)
    sq = torch.full((N,), 0.5, dtype=torch.float64, device=cuda)   # accumulates on top of what is there
    call("primia_conv2d_wgrad_persample_sqnorm", d, x, dy, sq, dt)
    assert rel(sq - 0.5, sq_ref) < 1e-6
    assert rel(sq_ref, (slab.double() ** 2).sum(1)) < 1e-10


@pytest.mark.parametrize("dtype,batch,size", [(torch.float32, 4, 64), (torch.bfloat16, 8, 64), (torch.bfloat16, 130, 32)])
def test_groupnorm_stem_fused_matches_the_chain(cuda, dtype, batch, size):
    """Stem tail gn1 -> relu -> maxpool as one op each way (primia_gn_relu_maxpool_fwd / _bwd, z and dz never written)
    against the three-op chain (primia_gn_fwd, primia_maxpool3x3s2_fwd / _bwd, primia_gn_relu_bwd): pooled activations
    and argmax codes are bit-equal (one value expression); the gradients agree to rounding — the fused backward adds
    the up-to-four window gradients of an element in fp32 where the chain rounds their sum to the storage type."""
    spec = rs.resnet18_spec(3, 3, size, "max")
    torch.manual_seed(21)
    sd = rs.init_state_dict(spec, "group")
    g = torch.Generator().manual_seed(22)
    x = torch.randn(batch, 3, size, size, generator=g).to(cuda)
    y = torch.randint(0, 3, (batch,), generator=g).to(cuda)
    out = []
    for fused in (True, False):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=dtype, device=cuda, norm="group")
        eng.gn_stem_fused = fused
        eng.load_state_dict(sd)
        eng.forward(x)
        eng.loss_backward(y)
        torch.cuda.synchronize()
        assert eng._stem_fused_gn == fused
        out.append((eng.t["pool.out"].clone(), eng.pool_argmax.clone(), eng.logits.clone(), eng.t["stem.dy"].clone(),
                    eng.grads.clone(), {k: eng.gviews[k].clone() for k in ("conv1.weight", "bn1.weight", "bn1.bias")}))
    a, b = out
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel(a[3], b[3]) < tol
    for k in a[5]:
        assert rel(a[5][k], b[5][k]) < tol, k
    assert rel(a[4], b[4]) < tol


@pytest.mark.parametrize("case", [(140, 16, 64, 64), (9, 24, 64, 64), (300, 8, 64, 64), (20, 14, 128, 128)])
def test_kept_per_sample_tiles_give_the_clipped_sum(cuda, case):
    """DP-SGD for layers with small per-sample gradients: primia_conv2d_wgrad_persample_sqnorm_keep stores every sample's
    tile while adding its squares; primia_conv_wgrad_clipped_sum = the ordered reduce of the kept tiles weighted by the
    clip factors.  Against the explicit per-sample slabs (primia_conv2d_wgrad_persample): norms to 1e-6, the clipped sum
    to fp32 rounding — both walk whole images per half-block (large batches) or one block per image."""
    from primia_amd._lib import ConvDesc

    N, H, C, K = case
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    d = ConvDesc.make(N, H, H, C, K, 3, 3, 1, 1)
    need = query("primia_conv_wgrad_persample_slab_bytes", d, dt)
    assert need > 0
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn(N * H * H, C, generator=g).relu().to(dtype).to(cuda)
    dy = (torch.randn(N * H * H, K, generator=g) * 1e-2).to(dtype).to(cuda)
    ne = query("primia_conv_wfwd_elems", d)
    slab = torch.zeros(N, ne, device=cuda)
    call("primia_conv2d_wgrad_persample", d, x, dy, slab, dt)
    sq_ref = (slab.double() ** 2).sum(1)
    clip = (torch.rand(N, generator=g) * 0.9 + 0.1).to(cuda)
    want = (slab * clip[:, None]).sum(0)
    keep = torch.full((need // 4,), float("nan"), device=cuda)
    sq = torch.zeros(N, dtype=torch.float64, device=cuda)
    call("primia_conv2d_wgrad_persample_sqnorm_keep", d, x, dy, sq, keep, need, dt)
    assert rel(sq, sq_ref) < 1e-6
    acc = torch.full((ne,), float("nan"), device=cuda)
    call("primia_conv_wgrad_clipped_sum", d, keep, clip, acc, dt)
    assert rel(acc, want) < 2e-6


def test_stem_kept_per_sample_tiles_give_the_clipped_sum(cuda):
    from primia_amd._lib import ConvDesc

    N, S = 6, 64
    dtype = torch.bfloat16
    dt = _lib.dtype_code(dtype)
    need = query("primia_stem_conv_wgrad_persample_slab_bytes", N, S, S)
    assert need == N * 64 * 256 * 4
    g = torch.Generator().manual_seed(3)
    xp = torch.zeros(N, S + 6, S + 8, 4)
    xp[:, 3:3 + S, 3:3 + S, :3] = torch.randn(N, S, S, 3, generator=g)
    xp = xp.to(dtype).to(cuda)
    dy = (torch.randn(N * (S // 2) ** 2, 64, generator=g) * 1e-2).to(dtype).to(cuda)
    keep = torch.full((need // 4,), float("nan"), device=cuda)
    sq = torch.zeros(N, dtype=torch.float64, device=cuda)
    call("primia_stem_conv_wgrad_persample_sqnorm_keep", xp, dy, sq, keep, need, N, S, S, dt)
    sq_ref = torch.zeros(N, dtype=torch.float64, device=cuda)
    call("primia_stem_conv_wgrad_persample_sqnorm", xp, dy, sq_ref, N, S, S, dt)
    assert rel(sq, sq_ref) < 1e-9
    tiles = keep.view(N, 64 * 256)
    assert rel((tiles.double() ** 2).sum(1), sq_ref) < 1e-6
    clip = (torch.rand(N, generator=g) * 0.9 + 0.1).to(cuda)
    acc = torch.full((64 * 256,), float("nan"), device=cuda)
    call("primia_stem_conv_wgrad_clipped_sum", keep, clip, acc, N)
    assert rel(acc, (tiles * clip[:, None]).sum(0)) < 2e-6
    # and against the batched kernel on the row-scaled dy (what the engine did before): bf16 rounding of the scaled rows
    dys = dy.clone()
    call("primia_scale_rows", dys, clip, N, dys.numel() // N, dt)
    wsb = query("primia_stem_conv_wgrad_ws_bytes", N, S, S)
    ws = torch.empty(wsb // 4, device=cuda)
    acc2 = torch.zeros(64 * 256, device=cuda)
    call("primia_stem_conv_wgrad_ws", xp, dys, acc2, ws, wsb, N, S, S, dt)
    assert rel(acc, acc2) < 1e-2


def test_dp_step_with_kept_tiles_matches_the_two_pass_form(cuda):
    """bf16 DP-SGD engine step with the kept-tile path (stem + layer1) against the same step with dp_keep off:
    same norms and clip factors (to fp64 summation order), clipped gradient within the bf16 rounding of the scaled rows."""
    batch, size = 130, 32
    spec = rs.resnet18_spec(3, 3, size, "max")
    torch.manual_seed(31)
    sd = rs.init_state_dict(spec, "group")
    g = torch.Generator().manual_seed(32)
    x = torch.randn(batch, 3, size, size, generator=g).to(cuda)
    y = torch.randint(0, 3, (batch,), generator=g).to(cuda)
    out = []
    for keep in (True, False):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda, norm="group")
        eng.dp_keep = keep
        eng.load_state_dict(sd)
        eng.forward(x)
        eng.dp_loss_backward(y, 1.0, 0.0, noise=torch.zeros(eng.P, device=cuda))
        torch.cuda.synchronize()
        assert bool(eng._dp_keep_buffers()) == keep
        out.append((eng.dp_stats["sq_norms"].clone(), eng.dp_stats["clip"].clone(), eng.grads.clone()))
    a, b = out
    assert rel(a[0], b[0]) < 1e-9 and rel(a[1], b[1]) < 1e-6
    assert rel(a[2], b[2]) < 2e-2


@pytest.mark.parametrize("dtype,batch,size", [(torch.float32, 4, 64), (torch.bfloat16, 8, 64), (torch.bfloat16, 130, 32)])
def test_groupnorm_relu_masks_give_the_same_bits(cuda, dtype, batch, size):
    """GroupNorm residual layers on 1-bit ReLU masks (primia_gn_fwd_mask / primia_gn_bwd_mask, identity blocks through
    primia_conv2d_dgrad_masked_acc: z is not read and the masked gradient not written) against the z-reading chain
    (primia_gn_fwd / primia_gn_bwd): logits, every gradient and the DP-SGD clipped sums bit for bit."""
    spec = rs.resnet18_spec(3, 3, size, "max")
    torch.manual_seed(31)
    sd = rs.init_state_dict(spec, "group")
    g = torch.Generator().manual_seed(32)
    x = torch.randn(batch, 3, size, size, generator=g).to(cuda)
    y = torch.randint(0, 3, (batch,), generator=g).to(cuda)
    out = []
    for masks in (True, False):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=dtype, device=cuda, norm="group")
        eng.gn_relu_masks = masks
        eng.load_state_dict(sd)
        eng.forward(x)
        eng.loss_backward(y)
        assert (len(eng.relu_masks) == 8) == masks
        plain = (eng.logits.clone(), eng.grads.clone())
        eng.forward(x)
        eng.dp_loss_backward(y, 1.0, 0.0, noise=torch.zeros(eng.P, device=cuda))
        out.append(plain + (eng.grads.clone(), eng.dp_stats["sq_norms"].clone()))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a, b)
    # (the fp32 norm pass adds its squares with atomics: equal to fp64 rounding, not to the bit)
    assert torch.allclose(out[0][3], out[1][3], rtol=1e-10, atol=0.0)


PS_KERNEL = {21: "ghost 7x7", 22: "ghost 14->7 stride 2", 24: "patch, block per image", 25: "patch, whole images per half",
             26: "per-tap, whole images per block", 13: "dma", 14: "generic", 15: "stem"}


def _norm_pass_kernels(eng):
    """{conv name: 'kept' | id of the kernel primia_conv2d_wgrad_persample_sqnorm dispatches to}."""
    kept = eng._dp_keep_buffers()
    return {c.name: ("kept" if c.name in kept else query("primia_conv_wgrad_persample_kernel_id", eng.convs[c.name].desc, eng.dt))
            for c in eng.spec.convs}


def test_dp_step_bf16_at_224_against_oracle(cuda):
    """BASELINE configs[3] at the resolution and dtype `bench.py --dp` times (224x224, bf16; batch 16 so that the oracle's
    per-sample loop stays in seconds): layer4 is 7x7 here, so the Gram-matrix norm kernels (dp_ghost.hip), the kept-tile
    clipped sums (stem, layer1, layer2) and the per-tap whole-image norm pass (transition blocks) are the paths taken —
    asserted — and their per-sample norms, clip factors and clipped sum are held to O.dp_gradients
    (reference rule: train.py:325-334).  Two oracles: the fp32 one (what the reference computes) and the same with
    nothing but bf16 rounding at the engine's storage points, which separates storage format from kernel error."""
    batch, size = 16, 224
    torch.manual_seed(41)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"), "group")
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda, norm="group")
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, 3, (batch,), generator=g)
    eng.forward(x.to(cuda))
    # at batch 16 every layer's per-sample tiles would fit the keep budget; keep what a batch-256 step keeps (stem,
    # layer1, layer2) so that layer3 / layer4 take the norm-pass kernels the benchmark runs
    for k in [k for k in eng._dp_keep_buffers() if k.startswith(("layer3", "layer4"))]:
        del eng._dp_keep[k]
    kern = _norm_pass_kernels(eng)
    assert kern["conv1"] == "kept" and all(kern[f"layer{l}.{b}.conv{c}"] == "kept" for l in (1, 2) for b in (0, 1)
                                           for c in (1, 2) if (l, b, c) != (2, 0, 1)), kern
    assert [kern[k] for k in ("layer4.0.conv2", "layer4.1.conv1", "layer4.1.conv2")] == [21, 21, 21], kern
    assert kern["layer4.0.conv1"] == 22, kern
    for k in ("layer2.0.conv1", "layer2.0.downsample.0", "layer3.0.conv1", "layer3.0.downsample.0", "layer4.0.downsample.0"):
        assert kern[k] == 26, (k, kern[k])
    fresh = lambda: {k: v.clone() for k, v in sd.items()}
    # the clipping norm: the oracle's median per-sample norm, so that half of the samples are clipped and half are not
    _, norms_any, _ = O.dp_gradients(fresh(), x, y, 1.0, 0.0, None)
    C = float(norms_any.median())
    eng.forward(x.to(cuda))
    eng.dp_loss_backward(y.to(cuda), C, 0.0, noise=torch.zeros(eng.P, device=cuda))
    got_norms = eng.dp_stats["sq_norms"].sqrt().cpu()
    got_clip = eng.dp_stats["clip"].cpu().double()
    want32, norms32, clip32 = O.dp_gradients(fresh(), x, y, C, 0.0, None)
    want16, norms16, clip16 = O.dp_gradients(fresh(), x, y, C, 0.0, None, bf16_storage=True)
    e32 = ((got_norms - norms32).abs() / norms32).max().item()
    e16 = ((got_norms - norms16).abs() / norms16).max().item()
    o16 = ((norms16 - norms32).abs() / norms32).max().item()
    print(f"per-sample norms: engine vs fp32 oracle {e32:.3e}, vs bf16-storage oracle {e16:.3e}, oracle bf16 vs fp32 {o16:.3e}")
    # measured on MI355X: 3.2e-3 / 1.8e-3 (the fp32 and the bf16-storage oracle are 2.6e-3 apart themselves)
    assert e16 < 5e-3 and e32 < 5e-3, (e16, e32, o16)
    assert (clip32 < 1).any() and (clip32 == 1).any(), "test should exercise clipped and unclipped samples"
    assert torch.allclose(got_clip, clip16, rtol=5e-3) and torch.allclose(got_clip, clip32, rtol=5e-3)
    flat = lambda d: torch.cat([d[k].reshape(-1).double() for k, _ in eng.p_entries])
    gvec = torch.cat([eng.gviews[k].reshape(-1).double().cpu() for k, _ in eng.p_entries])
    d32, d16, oo = rel(gvec, flat(want32)), rel(gvec, flat(want16)), rel(flat(want16), flat(want32))
    print(f"clipped mean gradient (all 62 tensors): engine vs fp32 {d32:.3e}, vs bf16-storage oracle {d16:.3e}, "
          f"oracle bf16 vs fp32 {oo:.3e}")
    # the engine is held to the band the storage format alone opens (as the BatchNorm path's bf16 gradient test does)
    assert d16 < 1.25 * oo + 0.02 and d32 < 1.25 * oo + 0.02, (d32, d16, oo)
    assert gvec.norm().item() <= C * 1.02


def test_dp_norm_pass_at_batch_256_against_explicit_slabs(cuda):
    """The DP bench's own size (batch 256, 224x224, bf16): every layer's contribution to ||g_n||^2 as the norm pass
    produced it (Gram-matrix, kept-tile, whole-images-per-half patch and per-tap kernels) against the squares of the
    EXPLICIT per-sample gradient slabs (primia_conv2d_wgrad_persample) formed from the same (x, dy) buffers."""
    from collections import OrderedDict

    batch, size = 256, 224
    torch.manual_seed(43)
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda, norm="group")
    eng.init_weights()
    g = torch.Generator().manual_seed(44)
    x = torch.randn(batch, 3, size, size, generator=g).to(cuda)
    y = torch.randint(0, 3, (batch,), generator=g).to(cuda)
    eng.forward(x)
    kern = _norm_pass_kernels(eng)
    assert {kern[f"layer3.{b}.conv{c}"] for b, c in ((0, 2), (1, 1), (1, 2))} == {25}, kern    # whole images per half
    assert kern["layer4.1.conv1"] == 21 and kern["layer4.0.conv1"] == 22 and kern["layer3.0.conv1"] == 26
    eng.dp_trace, eng.dp_keep_operands = OrderedDict(), True
    eng.forward(x)
    # C so large that every clip factor is exactly 1: the in-place row scaling of dy that follows the norm pass is x 1.0
    eng.dp_loss_backward(y, 1e12, 0.0, noise=torch.zeros(eng.P, device=cuda))
    torch.cuda.synchronize()
    assert torch.all(eng.dp_stats["clip"] == 1.0)
    prev, inc = None, {}
    for name, cum in eng.dp_trace.items():
        inc[name] = cum - prev if prev is not None else cum.clone()
        prev = cum
    x4 = torch.zeros(batch, size, size, 4, dtype=torch.bfloat16, device=cuda)
    x4[..., :3] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    worst = 0.0
    for name, xin, dy in eng.dp_operands:
        d = eng.convs[name].desc
        ne = query("primia_conv_wfwd_elems", d)
        slab = torch.zeros(batch, ne, device=cuda)
        call("primia_conv2d_wgrad_persample", d, x4.view(-1, 4) if name == "conv1" else xin, dy, slab, eng.dt)
        ref = (slab.double() ** 2).sum(1)
        e = ((inc[name] - ref).abs() / ref).max().item()
        worst = max(worst, e)
        assert e < 2e-5, (name, kern[name], e)
        del slab
    print(f"batch-256 norm pass vs explicit slabs: worst per-sample relative error {worst:.2e}")


def test_sibling_engine_keeps_the_dp_parameters(cuda):
    """A ragged or MixUp-halved batch runs on ResNet18Engine.sibling(n): it must be clipped and noised like every other
    batch — the DP-SGD parameters set on the root engine after construction (train.py) travel with every sibling() call."""
    eng = ResNet18Engine(4, 3, 3, 64, "max", dtype=torch.float32, device=cuda, norm="group")
    eng.init_weights()
    sib = eng.sibling(2)
    assert sib.dp_params is None
    eng.dp_params = {"max_grad_norm": 0.05, "noise_multiplier": 0.0}
    sib = eng.sibling(2)
    assert sib.dp_params == eng.dp_params and sib.norm == "group"
    eng.class_weight = torch.tensor([1.0, 2.0, 0.5], device=cuda)          # CrossEntropyLoss(weight=...) of train.py:316-319
    assert eng.sibling(2).class_weight is eng.class_weight
    eng.class_weight = None
    assert eng.sibling(2).class_weight is None
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(2, 3, 64, 64, generator=g).to(cuda), torch.randint(0, 3, (2,), generator=g).to(cuda)
    sib.forward(x)
    sib.loss_backward(y)                       # -> dp_loss_backward(**dp_params)
    assert sib.dp_stats["clip"].numel() == 2 and float(sib.dp_stats["clip"].max()) < 1.0
    assert float(eng.grads.norm()) <= 0.05 * 1.001          # the shared gradient arena holds the CLIPPED mean
