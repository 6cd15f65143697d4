"""GPU parity of the whole training step (forward, loss, backward, optimizer) against
 (a) the golden vectors minted from the reference's torchlib/models.py (fp32 engine, 1e-5), and
 (b) the CPU oracle run live on identical batches (bf16 engine, bf16-level tolerance)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import train_oracle as O  # noqa: E402
from primia_amd import resnet_spec as rs  # noqa: E402
from primia_amd.engine import ResNet18Engine  # noqa: E402

CASES = {
    # name: pooling, optimizer, lr, wd, class weights, soft targets
    "sgd_hard_64": ("max", "SGD", 1e-2, 5e-4, [0.5, 1.0, 2.0], False),
    "adam_soft_64": ("avg", "Adam", 1e-3, 5e-4, [0.5, 1.0, 2.0], True),
    "sgd_hard_224": ("max", "SGD", 1e-4, 5e-4, None, False),
}


def summary(t, n=8):
    f = t.detach().double().flatten().cpu()
    return np.array([f.norm().item(), f.sum().item()] + f[:n].tolist() + [0.0] * max(0, n - f.numel()))


def check_summary(got, want, rtol, what, abs_norm=0.0, abs_val=0.0):
    # [0] = L2 norm, [1] = sum, [2:] = leading values.  Norm to rtol; leading values to rtol of the
    # tensor's scale (norm / sqrt(n) is not stored, so use max |leading| as the scale).
    # abs_norm / abs_val: extra absolute slack (post-step weights inherit lr * gradient error).
    assert abs(got[0] - want[0]) <= rtol * max(abs(want[0]), 1e-12) + abs_norm, f"{what}: norm {got[0]} vs {want[0]}"
    scale = max(np.abs(want[2:]).max(), 1e-12)
    assert np.abs(got[2:] - want[2:]).max() <= 20 * rtol * scale + 1e-7 + abs_val, f"{what}: leading values"


@pytest.mark.parametrize("name", ["sgd_hard_64", "adam_soft_64", "sgd_hard_224"])
def test_fp32_engine_matches_reference_golden(cuda, golden_dir, name):
    pooling, optimizer, lr, wd, cw, soft = CASES[name]
    gold = np.load(os.path.join(golden_dir, f"train_{name}.npz"))
    seed, batch, size, steps = [int(v) for v in gold["meta"]]
    eng = ResNet18Engine(batch, 3, 3, size, pooling, dtype=torch.float32, device=cuda)
    torch.manual_seed(seed)
    eng.init_weights()  # consumes the RNG exactly like the reference constructor
    if cw:
        eng.class_weight = torch.tensor(cw, dtype=torch.float32, device=cuda)
    g = torch.Generator().manual_seed(seed + 1)
    for step in range(steps):
        x = torch.randn(batch, 3, size, size, generator=g)
        if soft:
            y = torch.rand(batch, 3, generator=g)
            y = y / y.sum(1, keepdim=True)
        else:
            y = torch.randint(0, 3, (batch,), generator=g)
        chaotic = optimizer != "SGD" and step > 0
        sd_before = eng.state_dict() if chaotic else None
        logits = eng.forward(x.to(cuda))
        loss = eng.loss_backward(y.to(cuda), soft=soft)
        if chaotic:
            # Adam's update is lr * m / (sqrt(v) + 1e-8): at step 0 it is lr * g / (|g| + 1e-8), discontinuous
            # at g = 0, so gradient components that are zero up to rounding move by +-lr depending on the
            # platform's rounding, and the TRAJECTORY after an Adam step is not comparable with the
            # reference's beyond ~1e-2 (observed 2e-3 .. 2e-2 on logits as kernels changed summation
            # order).  From step 1 on, the Adam case therefore checks every step against the ORACLE run
            # on the engine's own current weights and the same batch — the same bounds as step 0, without
            # trajectory chaos.  (The Adam kernel is held to 1e-5 on identical gradients in test_gpu_ops.)
            osd = {k: v.clone() for k, v in sd_before.items()}
            keys = O.param_keys(osd)
            for k in keys:
                osd[k].requires_grad_(True)
            ol = O.forward(osd, x, True, pooling, size)
            cwt = torch.tensor(cw) if cw else None
            oloss = O.cross_entropy_one_hot(ol, y, cwt) if soft else torch.nn.functional.cross_entropy(ol, y, cwt)
            oloss.backward()
            err = (logits.cpu().double() - ol.detach().double()).norm() / ol.detach().double().norm()
            assert err < 1e-5, f"step {step} logits vs oracle on the same weights: {err}"
            assert abs(loss.item() - oloss.item()) <= 1e-5 * abs(oloss.item())
            for k, _ in eng.p_entries:
                gg, go = eng.gviews[k].cpu().double().flatten(), osd[k].grad.double().flatten()
                assert (gg - go).norm() <= 1e-2 * go.norm() + 1e-9, f"step {step} grad {k} vs oracle"
            eng.adam_step(lr, (0.5, 0.99), 1e-8, wd)
            continue
        # Step 0 and every SGD step are pinned to the reference-derived golden vectors.
        rtol = 1e-5
        want = gold[f"s{step}.logits"]
        err = np.linalg.norm(logits.double().cpu().numpy() - want) / np.linalg.norm(want)
        assert err < rtol, f"step {step} logits rel err {err}"
        wl = float(gold[f"s{step}.loss"])
        assert abs(loss.item() - wl) <= rtol * abs(wl)
        # Gradients: every backward kernel is held to 1e-5 against autograd on identical inputs in
        # test_gpu_ops.py.  End to end, a gradient is NOT a continuous function of rounding: one
        # ReLU whose pre-activation is zero up to fp32 rounding (BN output ~ 0 added to an exactly
        # zero identity) flips its mask between platforms, which moves every upstream gradient by
        # ~1/sqrt(pixels per channel) of one element (measured on this network at batch 4: a single
        # flip in layer3.1 -> 2e-3 on all earlier layers, while all forward tensors agree to 5e-6).
        # So the end-to-end bound on gradients is 1e-2; logits, loss and post-step weights keep 1e-5.
        gtol = 1e-2
        for k, _ in eng.p_entries:
            check_summary(summary(eng.gviews[k]), gold[f"s{step}.grad.{k}"], gtol, f"step {step} grad {k}")
        if optimizer == "SGD":
            eng.sgd_step(lr, wd)
        else:
            eng.adam_step(lr, (0.5, 0.99), 1e-8, wd)
        sd = eng.state_dict()
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                assert int(v) == step + 1
            elif optimizer == "SGD":
                # p' = p - lr*(g + wd*p): 1e-5 on the weights plus lr times the gradient slack above
                gk = f"s{step}.grad.{k}"
                gn, gv = (gold[gk][0], np.abs(gold[gk][2:]).max()) if gk in gold.files else (0.0, 0.0)
                check_summary(summary(v), gold[f"s{step}.post.{k}"], 1e-5, f"step {step} post {k}",
                              abs_norm=lr * gtol * gn, abs_val=20 * lr * gtol * gv)
            else:
                # first Adam update = +-lr per component: components with |g| ~ 1e-8 may land on either side
                check_summary(summary(v), gold[f"s{step}.post.{k}"], 5e-3, f"step {step} post {k}")


def _nchw(t, N):
    hw = int(round((t.shape[0] // N) ** 0.5))
    return t.float().cpu().view(N, hw, hw, t.shape[1]).permute(0, 3, 1, 2).contiguous()


def _engine_relu_masks(eng, N):
    return {n: _nchw(v, N) > 0 for n, v in eng.t.items() if n == "stem.z" or n.endswith(".a1") or n.endswith(".out")
            and n != "pool.out"}


@pytest.mark.parametrize("batch,size,seed", [(4, 224, 42), (8, 64, 42), (8, 64, 7)])
def test_fp32_gradients_hold_1e5_once_relu_branches_agree(cuda, batch, size, seed):
    """The 1e-5 bar on GRADIENTS, demonstrated instead of argued.  A ReLU gradient is discontinuous where the
    pre-activation is zero up to rounding, so two fp32 implementations may take different branches at a handful of
    the ~10^7 ReLU sites (each flip moves every upstream gradient by ~1e-3).  The test
      1. counts the sites where the engine's mask differs from the oracle's (= the reference's, bit for bit):
         at most 8 of millions;
      2. re-runs the oracle with the ENGINE's masks: all 62 gradient tensors then agree within 1e-5 of the exact
         (float64) gradient of that network, and within 1.5e-5 of the torch-CPU fp32 oracle — which is itself
         ~5e-6 away from the float64 gradient."""
    torch.manual_seed(seed)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"))
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, 3, (batch,), generator=g)
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda)
    eng.fuse_stem = False            # keep z = relu(bn1(y)) so that the stem's mask can be read back
    eng.load_state_dict(sd)
    logits = eng.forward(x.to(cuda)).cpu()
    eng.loss_backward(y.to(cuda))
    masks = _engine_relu_masks(eng, batch)
    assert len(masks) == 17

    def oracle(dtype, use_masks):
        osd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k in O.param_keys(osd):
            osd[k].requires_grad_(True)
        taps = {}
        ol = O.forward(osd, x.to(dtype), True, "max", size, taps=taps, relu_masks=masks if use_masks else None)
        torch.nn.functional.cross_entropy(ol, y).backward()
        return ol.detach(), osd, taps

    ol, _, taps = oracle(torch.float32, False)
    flips = sum(int((masks[n] != (taps[n].detach() > 0)).sum()) for n in masks)
    assert flips <= 8, f"{flips} ReLU sites take the other branch"
    assert (logits - ol).norm() / ol.norm() < 1e-5
    _, osd32, _ = oracle(torch.float32, True)
    _, osd64, _ = oracle(torch.float64, True)
    for k, _ in eng.p_entries:
        mine = eng.gviews[k].double().cpu().flatten()
        exact, ref32 = osd64[k].grad.flatten(), osd32[k].grad.double().flatten()
        assert (mine - exact).norm() <= 1e-5 * exact.norm(), f"{k}: {((mine - exact).norm() / exact.norm()).item():.2e}"
        assert (mine - ref32).norm() <= 1.5e-5 * ref32.norm(), f"{k}: {((mine - ref32).norm() / ref32.norm()).item():.2e}"


def _bn_ref(sd, yt, name, B):
    """float64 training BatchNorm of the STORED tensor yt: (xhat, invstd, gamma, bn(y))."""
    yv = _nchw(yt, B).double()
    mean = yv.mean((0, 2, 3), keepdim=True)
    invstd = (yv.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5).rsqrt()
    gam = sd[name + ".weight"].double().view(1, -1, 1, 1)
    bet = sd[name + ".bias"].double().view(1, -1, 1, 1)
    xh = (yv - mean) * invstd
    return xh, invstd, gam, xh * gam + bet


def _bn_bwd_ref(g, xh, invstd, gam):
    """(dy, dgamma, dbeta) of training BatchNorm for the output gradient g, float64."""
    m = g.numel() / g.shape[1]
    dbeta, dgamma = g.sum((0, 2, 3)), (g * xh).sum((0, 2, 3))
    return gam * invstd * (g - dbeta.view(1, -1, 1, 1) / m - xh * dgamma.view(1, -1, 1, 1) / m), dgamma, dbeta


def _full_size_links(eng, sd, x, B, idxs, stem=True):
    """Every link of the step at full size on the ENGINE'S OWN operands (what the kernels read and wrote), float64 / fp32
    references formed here.  Returns {label: (relative error, bound)}; the caller asserts."""
    import torch.nn.functional as F

    def rel(a, b):
        return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()

    errs = {}
    blocks = eng.spec.blocks
    for i in idxs:
        blk = blocks[i]
        p = blk.prefix
        xin = eng.t["pool.out"] if i == 0 else eng.t[blocks[i - 1].prefix + ".out"]
        layers = [(blk.conv1, xin, "y1", "dy1", None), (blk.conv2, eng.t[p + ".a1"], "y2", "dy2", "da1")]
        if blk.down is not None:
            layers.append((blk.down, xin, "yd", "dyd", None))
        for c, xt, yn, dyn, dxn in layers:
            d = eng.convs[c.name].desc
            xc, yc, dyc = _nchw(xt, B), _nchw(eng.t[f"{p}.{yn}"], B), _nchw(eng.t[f"{p}.{dyn}"], B)
            w = sd[c.name + ".weight"].bfloat16().float()
            errs[f"{c.name} forward"] = (rel(yc, F.conv2d(xc, w, None, d.stride, d.pad)), 3e-3)
            gref = torch.nn.grad.conv2d_weight(xc, w.shape, dyc, d.stride, d.pad)
            errs[f"{c.name} wgrad"] = (rel(eng.gviews[c.name + ".weight"].float().cpu(), gref), 1e-4)
            if dxn is not None:
                dxref = torch.nn.grad.conv2d_input(xc.shape, w, dyc, d.stride, d.pad)
                errs[f"{c.name} dgrad"] = (rel(_nchw(eng.t[f"{p}.{dxn}"], B), dxref), 3e-3)
        # the BatchNorm passes of the block: forward a1 = relu(bn1(y1)), out = relu(bn2(y2) + residual) with batch statistics
        # (one bf16 rounding: 3e-3); backward of bn1 from the buffers it read (da1 = conv2's data gradient, y1) and wrote
        # (dy1, dgamma, dbeta); backward of bn2 (and of the downsample branch's BatchNorm) from the gradient w.r.t. the
        # block's output AS ITS BACKWARD PASS READ IT (engine.taps: the buffer is consumed in place afterwards), the ReLU
        # mask of the stored output and y2 / yd.  These are the passes whose reductions round 5 moved into the write-backs of
        # the data-gradient kernels (head_bnsums, pair_bnsums, acc_bnsums) — dy 3e-3, dgamma / dbeta 1e-4, as bn1.
        b1, b2 = rs.bn_name(blk.conv1.name), rs.bn_name(blk.conv2.name)
        xh1, invstd1, gam1, z1 = _bn_ref(sd, eng.t[p + ".y1"], b1, B)
        a1 = _nchw(eng.t[p + ".a1"], B).double()
        errs[f"{p} bn1 forward"] = (rel(a1, z1.clamp_min(0)), 3e-3)
        dyref, dgamma, dbeta = _bn_bwd_ref(_nchw(eng.t[p + ".da1"], B).double() * (a1 > 0), xh1, invstd1, gam1)
        errs[f"{p} bn1 backward dy"] = (rel(_nchw(eng.t[p + ".dy1"], B), dyref), 3e-3)
        errs[f"{p} bn1 dbeta"] = (rel(eng.gviews[b1 + ".bias"].cpu(), dbeta), 1e-4)
        errs[f"{p} bn1 dgamma"] = (rel(eng.gviews[b1 + ".weight"].cpu(), dgamma), 1e-4)
        del xh1, z1, dyref
        xh2, invstd2, gam2, z2 = _bn_ref(sd, eng.t[p + ".y2"], b2, B)
        if blk.down is None:
            res = _nchw(xin, B).double()
        else:   # the downsample branch's BatchNorm output is rounded to the storage type before the add
            bd = rs.bn_name(blk.down.name)
            xhd, invstdd, gamd, zd = _bn_ref(sd, eng.t[p + ".yd"], bd, B)
            res = zd.float().bfloat16().double()
        outv = _nchw(eng.t[p + ".out"], B).double()
        errs[f"{p} bn2 forward"] = (rel(outv, (z2 + res).clamp_min(0)), 3e-3)
        g = _nchw(eng.taps[p + ".dout_in"], B).double() * (outv > 0)
        dyref, dgamma, dbeta = _bn_bwd_ref(g, xh2, invstd2, gam2)
        errs[f"{p} bn2 backward dy"] = (rel(_nchw(eng.t[p + ".dy2"], B), dyref), 3e-3)
        errs[f"{p} bn2 dbeta"] = (rel(eng.gviews[b2 + ".bias"].cpu(), dbeta), 1e-4)
        errs[f"{p} bn2 dgamma"] = (rel(eng.gviews[b2 + ".weight"].cpu(), dgamma), 1e-4)
        if blk.down is not None:
            dyref, dgamma, dbeta = _bn_bwd_ref(g, xhd, invstdd, gamd)
            errs[f"{p} downsample bn backward dy"] = (rel(_nchw(eng.t[p + ".dyd"], B), dyref), 3e-3)
            errs[f"{p} downsample bn dbeta"] = (rel(eng.gviews[bd + ".bias"].cpu(), dbeta), 1e-4)
            errs[f"{p} downsample bn dgamma"] = (rel(eng.gviews[bd + ".weight"].cpu(), dgamma), 1e-4)
            del xhd, zd
        del xh2, z2, g, dyref, outv, res
    if not stem:
        return errs
    # ---- the stem's tail: conv1 <- bn1 <- relu <- maxpool, from the pool's output gradient (complete once layer1.0.conv1's
    # accumulating data gradient has run; nothing overwrites it), the pooled activation, the argmax codes and conv1's stored
    # output.  A window's gradient lands on its argmax element; bn1's sums (formed at pooled resolution inside that data
    # gradient's write-back, acc_bnsums mode 3) and conv1's weight gradient, whose dy tiles the fused kernel forms on the fly
    # in the storage type (primia_stem_bwd_fused).  dgamma there uses xhat = (p - beta) / gamma read off the STORED pooled
    # value p: p carries one bf16 rounding (half an ulp, uniform, zero mean), so sum g * xhat carries per channel a random
    # error of standard deviation sigma_c = sqrt(sum (g * ulp(p) / sqrt(12) / gamma)^2) — measured 4.7e-3 of |dgamma| over
    # the 64 channels at batch 256, because dgamma is itself a sum of 800 k terms of both signs.  The bound is therefore that
    # noise model: every channel within 5 sigma_c (+ 1e-4 |dgamma_c|), no common sign (a bias would show as a mean z-score
    # away from 0), 2e-2 overall.  16 channels at a time: 1.6 GB per float64 tensor otherwise.
    hw, ph = eng.stem_hw, eng.pool_hw
    assert eng.stem_bwd_fused_active
    gp_all = eng.t["pool.dout"].float().cpu().view(B, ph, ph, 64)
    pv_all = eng.t["pool.out"].float().cpu().view(B, ph, ph, 64)
    code_all = eng.pool_argmax.cpu().view(B, ph, ph, 64).long()
    y_all = eng.t["stem.y"].cpu().view(B, hw, hw, 64)
    # forward: pooled = maxpool(relu(bn1(y))) and the argmax element holds that maximum
    dy_ref = torch.empty(B, 64, hw, hw, dtype=torch.bfloat16)
    dbeta_all, dgamma_all = torch.empty(64, dtype=torch.float64), torch.empty(64, dtype=torch.float64)
    sigma_all = torch.empty(64, dtype=torch.float64)
    fwd_num = fwd_den = 0.0
    oo = torch.arange(ph)
    for c0 in range(0, 64, 16):
        cs = slice(c0, c0 + 16)
        yv = y_all[..., cs].double().permute(0, 3, 1, 2)                     # [B, 16, hw, hw]
        mean = yv.mean((0, 2, 3), keepdim=True)
        invstd = (yv.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5).rsqrt()
        gam, bet = sd["bn1.weight"][cs].double().view(1, -1, 1, 1), sd["bn1.bias"][cs].double().view(1, -1, 1, 1)
        xh = (yv - mean) * invstd
        z = (xh * gam + bet).clamp_min(0)
        pooled = F.max_pool2d(z, 3, 2, 1)
        pv = pv_all[..., cs].double().permute(0, 3, 1, 2)
        fwd_num += float((pv - pooled).pow(2).sum())
        fwd_den += float(pooled.pow(2).sum())
        code = code_all[..., cs].permute(0, 3, 1, 2)
        hh = 2 * oo.view(1, 1, ph, 1) - 1 + code // 3
        ww = 2 * oo.view(1, 1, 1, ph) - 1 + code % 3
        assert int(hh.min()) >= 0 and int(hh.max()) < hw and int(ww.min()) >= 0 and int(ww.max()) < hw
        nn_ = torch.arange(B).view(B, 1, 1, 1).expand_as(code)
        cc = torch.arange(16).view(1, 16, 1, 1).expand_as(code)
        # the argmax element IS the window maximum (of the values as stored: within one bf16 rounding of the float64 z)
        zmax = z[nn_, cc, hh, ww]
        assert float((zmax - pooled).abs().max()) <= 2.0 ** -7 * float(pooled.abs().max())
        gwin = gp_all[..., cs].double().permute(0, 3, 1, 2) * (pv > 0)
        ulp = torch.where(pv > 0, torch.exp2(torch.floor(torch.log2(pv.clamp_min(1e-30))) - 7), torch.zeros_like(pv))
        sigma_all[cs] = ((gwin * ulp / 12 ** 0.5 / gam) ** 2).sum((0, 2, 3)).sqrt()
        dz = torch.zeros_like(yv)
        dz.index_put_((nn_, cc, hh, ww), gwin, accumulate=True)
        dyc, dgamma, dbeta = _bn_bwd_ref(dz, xh, invstd, gam)
        dy_ref[:, cs] = dyc.float().bfloat16()
        dbeta_all[cs], dgamma_all[cs] = dbeta, dgamma
        del yv, xh, z, dz, dyc
    errs["stem pooled forward"] = ((fwd_num / fwd_den) ** 0.5, 3e-3)
    errs["stem bn1 dbeta"] = (rel(eng.gviews["bn1.bias"].cpu(), dbeta_all), 1e-4)
    dg = eng.gviews["bn1.weight"].double().cpu()
    errs["stem bn1 dgamma (xhat from the stored pooled value)"] = (rel(dg, dgamma_all), 2e-2)
    zs = (dg - dgamma_all) / sigma_all.clamp_min(1e-30)
    worst = float(((dg - dgamma_all).abs() / (5 * sigma_all + 1e-4 * dgamma_all.abs())).max())
    errs["stem bn1 dgamma, worst channel / (5 sigma of the stored value's rounding)"] = (worst, 1.0)
    errs["stem bn1 dgamma, |mean z-score| over the channels (bias)"] = (abs(float(zs.mean())), 0.75)
    xq = x.bfloat16().float()
    gref = torch.nn.grad.conv2d_weight(xq, sd["conv1.weight"].shape, dy_ref.float(), 2, 3)
    errs["conv1 wgrad (dy tiles formed on the fly)"] = (rel(eng.gviews["conv1.weight"].float().cpu(), gref), 1e-3)
    return errs


def _assert_links(errs):
    bad = {k: v for k, v in errs.items() if not v[0] <= v[1]}
    table = "\n".join(f"  {k:58s} {v[0]:.2e}  (bound {v[1]:.0e})" for k, v in errs.items())
    print("full-size links, relative errors:\n" + table)
    assert not bad, "\n".join(f"{k}: {v[0]:.2e} > {v[1]:.0e}" for k, v in bad.items())


def test_bf16_full_size_step_against_oracle(cuda):
    """BASELINE configs[1] exactly (batch 256, 3x224x224, bf16 storage / fp32 accumulate), the kernels the benchmark times:
      1. logits within 3e-2 and loss within 2e-3 of the fp32 CPU oracle on the same batch;
      2. every link of the step at full size on the ENGINE'S OWN operands (_full_size_links): every convolution — forward
         output within 3e-3 of conv2d(x, w) in fp32 (= one bf16 rounding of the result), weight gradient within 1e-4 of
         conv2d_weight(x, dy), data gradient (where it lands in a buffer of its own; the conv1 + downsample pair of the
         transition blocks: test_transition_dgrad_pair_at_full_size) within 3e-3; every block's BatchNorm forward passes
         (batch statistics, ReLU, residual / downsample branch) within 3e-3; the backward pass of EVERY BatchNorm — bn1 of
         each block, bn2 of each block (residual and downsample branch; their reductions ride in the data-gradient
         write-backs of the head, the transition pairs and layer1's accumulate forms since round 5), the stem's bn1 through
         the max-pool and conv1's weight gradient behind it — dy within 3e-3, dgamma / dbeta within 1e-4, from the buffers
         the kernels read and wrote;
      3. end-to-end gradients against the fp32 oracle: bf16 STORAGE makes them deviate by 25-50 % per tensor on this
         randomly initialised network — the CPU oracle with nothing but bf16 rounding at the engine's storage points
         (O.forward(bf16_storage=True)) is as far from the fp32 oracle as the engine is — so the bound is that band,
         and the engine must not be further from the fp32 gradients than the bf16-storage oracle is, by more than 25 %."""
    import torch.nn.functional as F

    B, S = 256, 224
    torch.manual_seed(42)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, S, "max"))
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, 3, S, S, generator=g)
    y = torch.randint(0, 3, (B,), generator=g)
    eng = ResNet18Engine(B, 3, 3, S, "max", dtype=torch.bfloat16, device=cuda)
    eng.load_state_dict(sd)
    eng.taps = {}
    logits = eng.forward(x.to(cuda)).float().cpu()
    loss = eng.loss_backward(y.to(cuda)).item()
    # the fused reductions are the paths taken
    assert eng.dgrad_bnsums and eng.pair_bnsums and eng.acc_bnsums and eng.head_bnsums
    assert set(eng._bwd_sum_bufs) >= {"head", "layer1.0.conv1.acc", "layer1.1.conv1.acc", "layer2.0.conv1.pair",
                                      "layer3.0.conv1.pair", "layer4.0.conv1.pair"}
    ologits, oloss, ograds = O.train_step(sd, x, y, 0.0, 0.0)
    assert (logits - ologits).norm() / ologits.norm() < 3e-2
    assert abs(loss - oloss.item()) < 2e-3 * abs(oloss.item())
    # ---- 2. per layer, full size, the engine's own bf16 operands -------------------------------------------------
    _assert_links(_full_size_links(eng, sd, x, B, range(len(eng.spec.blocks))))
    # ---- 3. end to end ------------------------------------------------------------------------------------------------
    osd = {k: v.clone() for k, v in sd.items()}
    for k in O.param_keys(osd):
        osd[k].requires_grad_(True)
    F.cross_entropy(O.forward(osd, x, True, "max", S, bf16_storage=True), y).backward()

    def rel(a, b):
        return ((a.double().flatten() - b.double().flatten()).norm() / b.double().norm()).item()

    for k, _ in eng.p_entries:
        e_engine, e_model = rel(eng.gviews[k].cpu(), ograds[k]), rel(osd[k].grad, ograds[k])
        assert e_engine < 0.6, f"{k}: {e_engine:.2f}"
        assert e_engine < 1.25 * e_model + 0.02, f"{k}: engine {e_engine:.3f} vs bf16-storage model {e_model:.3f}"


@pytest.mark.parametrize("c64_blocks,c64_stages", [(300, 4), (437, 3)])
def test_bf16_full_size_layer1_links_on_other_block_counts(cuda, c64_blocks, c64_stages):
    """The same full-size links for layer1 and the stem with conv3x3_c64_kernel on a NON-default number of persistent blocks
    (and ring depth): at 512 blocks every block walks the same number of patches; other counts give ragged ranges, so the
    kernel's counted waits on its LDS-DMA ring (halo pieces, old rows, BatchNorm rows, mask words one iteration ahead) see
    other phase relations at batch 256 than the default run above — the size at which round 5's asm-store hazard showed."""
    from primia_amd import _lib

    B, S = 256, 224
    torch.manual_seed(42)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, S, "max"))
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, 3, S, S, generator=g)
    y = torch.randint(0, 3, (B,), generator=g)
    _lib.set_option("c64_blocks", c64_blocks)
    _lib.set_option("c64_stages", c64_stages)
    try:
        eng = ResNet18Engine(B, 3, 3, S, "max", dtype=torch.bfloat16, device=cuda)
        eng.load_state_dict(sd)
        eng.taps = {}
        eng.forward(x.to(cuda))
        eng.loss_backward(y.to(cuda))
        torch.cuda.synchronize()
        _assert_links(_full_size_links(eng, sd, x, B, [0, 1]))
    finally:
        _lib.set_option("c64_blocks", 512)      # (the table's defaults, csrc/options.h)
        _lib.set_option("c64_stages", 4)


def test_bf16_engine_tracks_oracle(cuda):
    """bf16 storage / fp32 accumulate: compare with the fp32 oracle on the same batch.
    Tolerance: logits 3e-2 of their norm, loss 2e-2, per-tensor gradient cosine > 0.9 (the
    earliest layers see bf16 rounding of 20 layers of backward; batch 8 makes BN noisy)."""
    batch, size = 8, 64
    torch.manual_seed(123)
    spec = rs.resnet18_spec(3, 3, size, "max")
    sd = rs.init_state_dict(spec)
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, 3, (batch,), generator=g)
    logits = eng.forward(x.to(cuda)).cpu()
    loss = eng.loss_backward(y.to(cuda)).item()
    ologits, oloss, ograds = O.train_step(sd, x, y, 0.0, 0.0)
    assert (logits - ologits).norm() / ologits.norm() < 3e-2
    assert abs(loss - oloss.item()) < 2e-2 * abs(oloss.item())
    worst = 1.0
    for k, _ in eng.p_entries:
        a, b = eng.gviews[k].double().cpu().flatten(), ograds[k].double().flatten()
        if b.norm() < 1e-8:
            continue
        cos = (a @ b / (a.norm() * b.norm())).item()
        worst = min(worst, cos)
        assert cos > 0.9, f"{k}: cosine {cos}"
        assert abs(a.norm() / b.norm() - 1) < 0.2, k


def test_state_dict_roundtrip_and_eval(cuda):
    batch, size = 2, 64
    torch.manual_seed(1)
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda)
    eng.init_weights()
    sd = eng.state_dict()
    assert list(sd.keys()) == rs.state_dict_keys(eng.spec)
    eng2 = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.float32, device=cuda)
    eng2.load_state_dict(sd)
    x = torch.randn(batch, 3, size, size)
    eng.eval()
    eng2.eval()
    a = eng.forward(x.to(cuda)).cpu()
    b = eng2.forward(x.to(cuda)).cpu()
    assert torch.equal(a, b)
    ref = O.forward({k: v.clone() for k, v in sd.items()}, x, training=False, input_size=size)
    assert (a - ref).norm() / ref.norm() < 1e-5
    with pytest.raises(KeyError):
        eng.load_state_dict({"conv1.weight": sd["conv1.weight"]})


@pytest.mark.parametrize("kind,dtype", [("SGD", torch.bfloat16), ("Adam", torch.float32)])
def test_sibling_engine_trains_the_same_model_at_another_batch_size(cuda, kind, dtype):
    """ResNet18Engine.sibling(n): another batch size on the SAME parameters, running statistics, kernel-layout weight
    copies and optimizer state — what the reference's local loop needs when MixUp halves a batch with probability
    mixup_prob (torchlib/utils.py:1262-1267).  Steps of 8, 4, 8 samples through root + sibling against two separate
    engines that hand the model and the optimizer over as state dicts: the same bits."""
    from primia_amd.optim import EngineOptimizer

    B, S = 8, 64
    torch.manual_seed(21)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, S, "max"))
    g = torch.Generator().manual_seed(22)
    seq = [(torch.randn(n, 3, S, S, generator=g).to(cuda), torch.randint(0, 3, (n,), generator=g).to(cuda)) for n in (B, B // 2, B)]
    kw = dict(lr=0.05, weight_decay=1e-4) if kind == "SGD" else dict(lr=1e-3, weight_decay=5e-4, betas=(0.5, 0.99))
    root = ResNet18Engine(B, 3, 3, S, "max", dtype=dtype, device=cuda)
    root.load_state_dict(sd)
    opt = EngineOptimizer(root, kind, **kw)
    for x, y in seq:
        eng = root.sibling(x.shape[0])
        assert (eng is root) == (x.shape[0] == B) and eng.flat.data_ptr() == root.flat.data_ptr()
        eng.forward(x)
        eng.loss_backward(y)
        opt.step(eng)
    got, got_opt = root.state_dict(), opt.state_dict()
    engines = {n: ResNet18Engine(n, 3, 3, S, "max", dtype=dtype, device=cuda) for n in (B, B // 2)}
    cur, cur_opt = sd, None
    for x, y in seq:
        e = engines[x.shape[0]]
        e.load_state_dict(cur)
        o = EngineOptimizer(e, kind, **kw)
        if cur_opt is not None:
            o.load_state_dict(cur_opt)
        e.forward(x)
        e.loss_backward(y)
        o.step()
        cur, cur_opt = e.state_dict(), o.state_dict()
    for k in cur:
        assert torch.equal(got[k].cpu(), cur[k].cpu()), k
    if kind == "Adam":
        assert got_opt["state"][0]["step"] == 3 == cur_opt["state"][0]["step"]
        assert torch.equal(got_opt["state"][5]["exp_avg_sq"], cur_opt["state"][5]["exp_avg_sq"])
    with pytest.raises(ValueError):
        opt.step(engines[B])                      # not a sibling


@pytest.mark.parametrize("batch,size", [(160, 64), (5, 96)])
def test_eval_stem_as_one_pass_equals_the_chain(cuda, batch, size):
    """Eval mode, bf16: conv1 -> bn1 -> relu -> maxpool as ONE kernel over the input (primia_stem_conv_pool_eval; the
    engine takes it from 160 images per batch) against primia_stem_conv_fwd -> primia_bn_fwd_eval ->
    primia_maxpool3x3s2_fwd: same pooled tensor, same logits, bit for bit, on running statistics that are not (0, 1)."""
    torch.manual_seed(11)
    spec = rs.resnet18_spec(3, 3, size, "max")
    sd = rs.init_state_dict(spec)
    sd["bn1.running_mean"] = torch.randn(64) * 0.2
    sd["bn1.running_var"] = torch.rand(64) + 0.3
    sd["bn1.weight"] = torch.rand(64) + 0.5
    sd["bn1.weight"][3] *= -1
    sd["bn1.bias"] = torch.randn(64) * 0.3
    x = torch.randn(batch, 3, size, size).to(cuda)
    outs = []
    for one_pass in (True, False):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda)
        eng.load_state_dict(sd)
        eng.eval()
        eng.stem_eval_one_pass_min_batch = 1 if one_pass else 10 ** 9
        assert eng._stem_eval_one_pass() == one_pass
        eng.t["stem.y"].fill_(float("nan"))
        logits = eng.forward(x).clone()
        assert bool(torch.isnan(eng.t["stem.y"].float()).all()) == one_pass      # conv1's output: never written
        outs.append((logits, eng.t["pool.out"].clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[1][0])
    assert float(outs[0][1].float().abs().max()) > 0


@pytest.mark.parametrize("dtype,batch,size", [(torch.bfloat16, 16, 64), (torch.float32, 4, 64), (torch.bfloat16, 6, 96)])
def test_training_step_is_bitwise_deterministic(cuda, dtype, batch, size):
    """Every weight gradient leaves through per-block partial tiles added in a fixed order (primia_conv2d_wgrad_ws,
    primia_stem_conv_wgrad_ws), every BatchNorm reduction is two-level in a fixed order: the same weights and batches
    give bit-identical weights after several SGD steps, run after run (no floating-point atomics on the path)."""
    spec = rs.resnet18_spec(3, 3, size, "max")
    torch.manual_seed(7)
    sd = rs.init_state_dict(spec)
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(batch, 3, size, size, generator=g).to(cuda) for _ in range(2)]
    ys = [torch.randint(0, 3, (batch,), generator=g).to(cuda) for _ in range(2)]
    outs = []
    for _ in range(2):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=dtype, device=cuda)
        eng.load_state_dict(sd)
        for i in range(4):
            eng.forward(xs[i % 2])
            eng.loss_backward(ys[i % 2])
            eng.sgd_step(1e-2, 5e-4)
        torch.cuda.synchronize()
        outs.append((eng.flat.clone(), eng.grads.clone(), eng.loss.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("batch,size,fused_tail", [(8, 64, False), (8, 64, True), (16, 224, True)])
def test_weight_gradient_accumulators_are_overwritten(cuda, batch, size, fused_tail):
    """The bf16 step has no fill launch: every convolution's weight-gradient accumulator — conv1's included, where its weight
    gradient runs on the padded input (primia_stem_bwd_fused) — is WRITTEN by an ordered reduce, never accumulated into.
    Poison the whole arena (and the workspace) with NaN between two steps on the same batch: same gradients, bit for bit."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(batch, 3, size, size, generator=g).to(cuda)
    y = torch.randint(0, 3, (batch,), generator=g).to(cuda)
    eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda)
    torch.manual_seed(5)
    eng.init_weights()
    eng.fuse_sgd_tail = fused_tail
    eng.forward(x)
    eng.loss_backward(y)
    assert eng.stem_bwd_fused_active and eng._acc_zero_only_stem
    ref = eng.grads.clone()
    assert bool(torch.isfinite(ref).all()) and float(ref.abs().max()) > 0
    eng.dw_acc.fill_(float("nan"))
    eng.wgrad_ws.fill_(float("nan"))
    eng._grads.fill_(float("nan"))
    eng.forward(x)
    eng.loss_backward(y)
    assert torch.equal(eng.grads, ref)
    # ... and the weights the fused tail makes from them are finite
    eng.forward(x)
    eng.loss_backward(y)
    eng.sgd_step(1e-2, 5e-4)
    assert bool(torch.isfinite(eng.flat).all())
    for c in eng.convs.values():
        assert bool(torch.isfinite(c.w_fwd.float()).all()), c.spec.name


def test_full_size_forward_and_step_are_repeatable(cuda):
    """BASELINE configs[1] size (batch 256, 224x224, bf16): the same batch gives bit-identical activations forward
    after forward, and the same gradients step after step.  (Small shapes never put two blocks of the 64->64 kernel on
    one CU; at this size they compete for the LDS, which once let a write-back read a result row before the wave that
    computes it had finished writing it.)"""
    B, S = 256, 224
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, S, S, generator=g).to(cuda)
    y = torch.randint(0, 3, (B,), generator=g).to(cuda)
    eng = ResNet18Engine(B, 3, 3, S, "max", dtype=torch.bfloat16, device=cuda)
    torch.manual_seed(42)
    eng.init_weights()
    names = ["pool.out", "layer1.0.y1", "layer1.0.out", "layer1.1.y2", "layer1.1.out", "layer2.1.out", "layer4.1.out"]
    ref = ref_g = None
    for _ in range(10):
        eng.forward(x)
        eng.loss_backward(y)
        torch.cuda.synchronize()
        cur = {n: eng.t[n].clone() for n in names}
        if ref is None:
            ref, ref_g = cur, eng.grads.clone()
            continue
        for n in names:
            assert torch.equal(ref[n], cur[n]), n
        assert torch.equal(ref_g, eng.grads)


@pytest.mark.parametrize("batch,size", [(16, 64), (64, 224)])
def test_weight_gradients_on_a_second_stream_give_the_same_bits(cuda, batch, size):
    """engine.wgrad_overlap: the weight gradients are leaves of the backward graph, so they may
    run on a second stream (1: beside the BatchNorm chain, joined before the next data gradient; 2: free-running until
    the finalize).  Same kernels on the same operands: weights, gradients and loss after three steps are bit-equal to
    the one-stream schedule."""
    spec = rs.resnet18_spec(3, 3, size, "max")
    torch.manual_seed(11)
    sd = rs.init_state_dict(spec)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(batch, 3, size, size, generator=g).to(cuda)
    y = torch.randint(0, 3, (batch,), generator=g).to(cuda)
    outs = []
    for mode in (0, 1, 2, 3):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda)
        eng.wgrad_overlap = mode
        eng.load_state_dict(sd)
        for _ in range(3):
            eng.forward(x)
            eng.loss_backward(y)
            eng.sgd_step(1e-2, 5e-4)
        torch.cuda.synchronize()
        outs.append((eng.flat.clone(), eng.grads.clone(), eng.loss.clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,batch,size", [(torch.bfloat16, 8, 64), (torch.float32, 4, 64), (torch.bfloat16, 16, 224)])
def test_fused_sgd_tail_gives_the_same_bits(cuda, dtype, batch, size):
    """engine.fuse_sgd_tail (primia_conv_sgd_step_many + primia_sgd_step_ranges) against the three unfused passes
    (finalize -> primia_sgd_step -> weight refresh): gradients, master weights and both kernel-layout copies of every
    convolution bit for bit, over two steps; a reader that comes between backward and step (materialize_grads, Adam)
    finds the gradients the unfused pass would have written."""
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(batch, 3, size, size, generator=g).to(cuda) for _ in range(2)]
    ys = [torch.randint(0, 3, (batch,), generator=g).to(cuda) for _ in range(2)]

    def run(fused):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=dtype, device=cuda)
        torch.manual_seed(3)
        eng.init_weights()
        eng.fuse_sgd_tail = fused
        for x, y in zip(xs, ys):
            eng.forward(x)
            eng.loss_backward(y)
            assert eng._grads_pending == fused
            eng.sgd_step(1e-2, 5e-4)
            assert not eng._grads_pending
        return eng

    a, b = run(False), run(True)
    assert torch.equal(a.grads, b.grads)
    assert torch.equal(a.flat, b.flat)
    for name, ca in a.convs.items():
        cb = b.convs[name]
        assert torch.equal(ca.w_fwd.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                           cb.w_fwd.view(torch.int16 if dtype == torch.bfloat16 else torch.int32)), name
        if ca.w_dgrad is not None:
            assert torch.equal(ca.w_dgrad.float(), cb.w_dgrad.float()), name
    # a reader between backward and step
    b.forward(xs[0]); b.loss_backward(ys[0])
    a.forward(xs[0]); a.loss_backward(ys[0])
    assert b._grads_pending
    b.materialize_grads()
    assert not b._grads_pending and torch.equal(a.grads, b.grads)
    b.forward(xs[1]); b.loss_backward(ys[1])
    a.forward(xs[1]); a.loss_backward(ys[1])
    a.adam_step(1e-3)
    b.adam_step(1e-3)
    assert torch.equal(a.flat, b.flat)


def test_transition_dgrad_pair_at_full_size(cuda):
    """conv1 (3x3 / 2) + downsample (1x1 / 2) data gradients of the three transition blocks in one pass
    (primia_conv2d_dgrad_pair) at batch 256 on the engine's weight copies, against conv2d_input in fp32: 3e-3 (one bf16
    rounding of the sum).  In the training step this buffer is consumed in place by the block before, so the full-size
    step test cannot look at it afterwards."""
    from primia_amd._lib import call

    B, S = 256, 224
    torch.manual_seed(44)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, S, "max"))
    eng = ResNet18Engine(B, 3, 3, S, "max", dtype=torch.bfloat16, device=cuda)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(45)
    for blk in eng.spec.blocks:
        if blk.down is None:
            continue
        c1, cd = eng.convs[blk.conv1.name], eng.convs[blk.down.name]
        d = c1.desc
        M = B * d.Ho * d.Wo
        dy1 = torch.randn(M, d.K, generator=g).bfloat16().to(cuda)
        dyd = torch.randn(M, d.K, generator=g).bfloat16().to(cuda)
        dx = torch.empty(B * d.H * d.W, d.C, dtype=torch.bfloat16, device=cuda)
        call("primia_conv2d_dgrad_pair", c1.desc, dy1, c1.w_dgrad, cd.desc, dyd, cd.w_dgrad, dx, eng.dt)
        w1 = sd[blk.conv1.name + ".weight"].bfloat16().float()
        wd = sd[blk.down.name + ".weight"].bfloat16().float()
        xs = (B, d.C, d.H, d.W)
        ref = (torch.nn.grad.conv2d_input(xs, w1, _nchw(dy1, B), d.stride, d.pad)
               + torch.nn.grad.conv2d_input(xs, wd, _nchw(dyd, B), cd.desc.stride, cd.desc.pad))
        got = _nchw(dx, B)
        assert (got - ref).norm() <= 3e-3 * ref.norm(), f"{blk.prefix}: {((got - ref).norm() / ref.norm()).item():.2e}"


@pytest.mark.parametrize("batch,size", [(6, 64), (16, 224)])
def test_transition_forward_pair_gives_the_same_bits(cuda, batch, size):
    """conv1 (3x3 / 2) + downsample (1x1 / 2) of a transition block in one launch (primia_conv2d_fwd_stats_pair) against
    the two single launches (primia_conv2d_fwd_stats): outputs and per-tile BatchNorm partials bit for bit (ragged last
    tiles included), and the whole step with and without the pairing."""
    from primia_amd._lib import call, query

    g = torch.Generator().manual_seed(11)
    x = torch.randn(batch, 3, size, size, generator=g).to(cuda)
    y = torch.randint(0, 3, (batch,), generator=g).to(cuda)
    res = []
    for pair in (True, False):
        eng = ResNet18Engine(batch, 3, 3, size, "max", dtype=torch.bfloat16, device=cuda)
        torch.manual_seed(5)
        eng.init_weights()
        eng.fwd_pair = pair
        eng.forward(x)
        eng.loss_backward(y)
        res.append((eng.logits.clone(), eng.grads.clone(), eng))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    eng = res[0][2]
    blocks = eng.spec.blocks
    for i, blk in enumerate(blocks):
        if blk.down is None:
            continue
        c1, cd = eng.convs[blk.conv1.name], eng.convs[blk.down.name]
        assert query("primia_conv_fwd_pair_ok", c1.desc, cd.desc, eng.dt) == 1
        xin = eng.t[blocks[i - 1].prefix + ".out"]
        y1, yd = torch.empty_like(eng.t[blk.prefix + ".y1"]), torch.empty_like(eng.t[blk.prefix + ".yd"])
        s1, sd = torch.zeros_like(c1.sums), torch.zeros_like(cd.sums)
        call("primia_conv2d_fwd_stats", c1.desc, xin, c1.w_fwd, y1, s1, eng.dt)
        call("primia_conv2d_fwd_stats", cd.desc, xin, cd.w_fwd, yd, sd, eng.dt)
        p1, pd = torch.empty_like(y1), torch.empty_like(yd)
        q1, qd = torch.zeros_like(s1), torch.zeros_like(sd)
        call("primia_conv2d_fwd_stats_pair", c1.desc, xin, c1.w_fwd, p1, q1, cd.desc, cd.w_fwd, pd, qd, eng.dt)
        assert torch.equal(p1.view(torch.int16), y1.view(torch.int16)) and torch.equal(pd.view(torch.int16), yd.view(torch.int16))
        assert torch.equal(q1, s1) and torch.equal(qd, sd), blk.prefix


def test_schedule_switches_of_round_6_give_the_same_bits(cuda):
    """engine.wgrad_flush_last (a stage's last same-shape weight gradient runs as soon as its dy exists instead of at the end of
    the backward pass — batch 256's layer4 groups 2 of its 3 layers; here a batch where the grouping leaves a remainder too)
    only moves a launch: gradients and weights after two steps are bit-equal.  Siblings are evicted least-recently-USED."""
    g = torch.Generator().manual_seed(17)
    x = torch.randn(32, 3, 224, 224, generator=g).to(cuda)
    y = torch.randint(0, 3, (32,), generator=g).to(cuda)
    outs = []
    for flush in (True, False):
        eng = ResNet18Engine(32, 3, 3, 224, "max", dtype=torch.bfloat16, device=cuda, options={"wgrad_flush_last": flush})
        torch.manual_seed(5)
        eng.init_weights()
        for _ in range(2):
            eng.forward(x)
            eng.loss_backward(y)
            eng.sgd_step(1e-2, 5e-4)
        outs.append((eng.flat.clone(), eng.grads.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    root = ResNet18Engine(4, 3, 3, 64, "max", dtype=torch.bfloat16, device=cuda)
    root.max_siblings = 3
    for n in (1, 2, 3):
        root.sibling(n)
    root.sibling(1)                  # a hit: batch 1 becomes the most recently used
    root.sibling(5)                  # evicts batch 2, not batch 1
    assert list(root._siblings) == [3, 1, 5]


def test_torchlib_models_resnet18_builds_the_engine(cuda):
    """`from torchlib.models import resnet18` with the keyword arguments /root/reference/train.py:259-268 passes (+ the
    batch size the engine needs): the same network, state-dict compatible with the reference's 122 keys."""
    from torchlib.models import resnet18

    torch.manual_seed(3)
    m = resnet18(pretrained=False, num_classes=3, in_channels=3, adptpool=False, input_size=64, pooling="max", batch_size=4,
                 dtype=torch.float32, device=cuda)
    assert isinstance(m, ResNet18Engine) and list(m.state_dict().keys()) == rs.state_dict_keys(m.spec)
    x = torch.randn(4, 3, 64, 64).to(cuda)
    m.eval()
    ref = O.forward({k: v.clone().cpu() for k, v in m.state_dict().items()}, x.cpu(), training=False, input_size=64)
    out = m.forward(x).cpu()
    assert (out - ref).norm() / ref.norm() < 1e-5
    g = resnet18(num_classes=3, input_size=64, pooling="max", batch_size=2, norm_layer=object(), dtype=torch.float32, device=cuda)
    assert g.norm == "group"
