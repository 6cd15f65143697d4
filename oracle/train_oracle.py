"""ORACLE (test infrastructure, not product code) — CPU restatement of PriMIA's plaintext training
step for ResNet-18.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product package (primia_amd/) never does.

What it restates (reference = /root/reference, torch 1.4 semantics):
  * the network  — torchlib/models.py:238-284 (BasicBlock), :345-423 (ResNet ctor), :466-482
    (_forward_impl), :499-516 (resnet18) with the arguments of train.py:257-266
    (num_classes=3, adptpool=False -> AvgPool2d(input_size/32), pooling max|avg);
  * the losses   — nn.CrossEntropyLoss(weight, mean) (train.py:335-340) and
    Cross_entropy_one_hot (torchlib/utils.py:404-441);
  * the optimizers — torch.optim.SGD(lr, weight_decay) without momentum and torch-1.4 Adam
    (train.py:280-303);
  * LearningRateScheduler (torchlib/utils.py:37-89) and MixUp (torchlib/utils.py:327-400).

It runs on torch-CPU fp32 kernels (ATen) — the same native code a PySyft VirtualWorker executes for
the reference — written functionally over a reference-compatible state dict.

Parity pin: validated in this container against the reference's own torchlib/models.py loaded
from /root/reference (tests/golden/make_train_golden.py): identical logits, loss, gradients and
post-step parameters on seeded inputs; the resulting vectors are committed under tests/golden/.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BLOCKS = [("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2)]


def _bn(x, sd, prefix, training, momentum=0.1, eps=1e-5):
    if (prefix + ".running_mean") not in sd:
        # norm_layer = GroupNorm(32, C) (torchlib/models.py:355,362-364): the BN-free network of the DP
        # configuration (train.py:308 rejects BatchNorm under the PrivacyEngine)
        return F.group_norm(x, 32, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)
    # F.batch_norm updates running stats in place when training (models.py:261-264 via nn.BatchNorm2d)
    out = F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                       sd[prefix + ".bias"], training, momentum, eps)
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return out


class _StoreBF16(torch.autograd.Function):
    """A tensor that is STORED in bf16: the value is rounded on the way forward and its gradient on the way back
    (what a bf16 activation / activation-gradient buffer does to an fp32-accumulated result)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def forward(sd, x, training=True, pooling="max", input_size=224, taps=None, relu_masks=None, bf16_storage=False):
    """ResNet._forward_impl (models.py:466-482).  `sd` maps reference state-dict keys to tensors
    (parameters may require grad).  `taps`, if a dict, receives named intermediates.
    `bf16_storage` (tests only): model the bf16 engine's storage format — the input, every convolution output, every
    block activation and the compute copies of the conv weights are rounded to bf16 where the engine keeps them in
    bf16 buffers (fp32 accumulation, BatchNorm statistics and the master weights stay fp32).
    `relu_masks` (tests only): {site: bool NCHW mask} — at the named ReLU sites ("stem.z", "<block>.a1",
    "<block>.out") the activation is `pre * mask` instead of relu(pre).  With the masks another implementation
    produced, the backward pass follows that implementation's branch at pre-activations that are zero up to
    rounding, which makes gradients comparable at fp32 accuracy (a ReLU gradient is discontinuous there)."""

    def tap(name, v):
        if taps is not None:
            if v.requires_grad:
                v.retain_grad()
            taps[name] = v
        return v

    def relu(name, pre):
        if relu_masks is not None and name in relu_masks:
            return pre * relu_masks[name].to(pre.dtype)
        return F.relu(pre)

    q = _StoreBF16.apply if bf16_storage else (lambda t: t)
    x = tap("stem.y", q(F.conv2d(q(x), q(sd["conv1.weight"]), None, 2, 3)))
    x = tap("stem.z", relu("stem.z", _bn(x, sd, "bn1", training)))
    if pooling == "max":
        x = F.max_pool2d(x, 3, 2, 1)
    else:
        x = F.avg_pool2d(x, 3, 2, 1)
    x = tap("pool.out", q(x))
    for lname, planes, stride in BLOCKS:
        for bi in range(2):
            p = f"{lname}.{bi}"
            s = stride if bi == 0 else 1
            identity = x
            out = tap(p + ".y1", q(F.conv2d(x, q(sd[p + ".conv1.weight"]), None, s, 1)))
            out = tap(p + ".a1", q(relu(p + ".a1", _bn(out, sd, p + ".bn1", training))))
            out = tap(p + ".y2", q(F.conv2d(out, q(sd[p + ".conv2.weight"]), None, 1, 1)))
            out = _bn(out, sd, p + ".bn2", training)
            if (p + ".downsample.0.weight") in sd:
                identity = q(F.conv2d(x, q(sd[p + ".downsample.0.weight"]), None, s, 0))
                identity = _bn(identity, sd, p + ".downsample.1", training)
            x = tap(p + ".out", q(relu(p + ".out", out + identity)))
    x = F.avg_pool2d(x, int(input_size / 32))
    x = torch.flatten(x, 1)
    return F.linear(x, sd["fc.weight"], sd["fc.bias"])


def cross_entropy_one_hot(output, target, weight=None, reduction="mean"):
    """Cross_entropy_one_hot.forward (torchlib/utils.py:417-441)."""
    per = (torch.sum(weight * target, dim=1) if weight is not None else 1.0) * torch.sum(
        -target * F.log_softmax(output, dim=1), dim=1)
    return torch.mean(per) if reduction == "mean" else torch.sum(per)


def sgd_step(params, grads, lr, weight_decay):
    """torch.optim.SGD.step with momentum=0 (train.py:280-282)."""
    with torch.no_grad():
        for p, g in zip(params, grads):
            d_p = g.add(p, alpha=weight_decay) if weight_decay != 0 else g
            p.add_(d_p, alpha=-lr)


def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch-1.4 torch.optim.Adam.step (L2-coupled weight decay)."""
    import math

    b1, b2 = betas
    state["step"] = state.get("step", 0) + 1
    st = state["step"]
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            if i not in state:
                state[i] = (torch.zeros_like(p), torch.zeros_like(p))
            m, v = state[i]
            if weight_decay != 0:
                g = g.add(p, alpha=weight_decay)
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(1 - b2 ** st)).add_(eps)
            p.addcdiv_(m, denom, value=-(lr / (1 - b1 ** st)))


def param_keys(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def train_step(sd, x, target, lr, weight_decay, class_weight=None, soft=False, pooling="max",
               optimizer="SGD", opt_state=None, betas=(0.9, 0.999), taps=None):
    """One iteration of the reference's batch loop (torchlib/utils.py:1168-1173 / :1255-1270):
    zero_grad, forward, loss, backward, optimizer.step.  Mutates `sd` in place.
    Returns (logits, loss, grads dict)."""
    keys = param_keys(sd)
    for k in keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    logits = forward(sd, x, True, pooling, x.shape[-1], taps)
    if soft:
        loss = cross_entropy_one_hot(logits, target, class_weight)
    else:
        loss = F.cross_entropy(logits, target, weight=class_weight, reduction="mean")
    loss.backward()
    grads = OrderedDict((k, sd[k].grad.detach().clone()) for k in keys)
    for k in keys:
        sd[k].requires_grad_(False)
    if optimizer == "SGD":
        sgd_step([sd[k] for k in keys], [grads[k] for k in keys], lr, weight_decay)
    else:
        adam_step([sd[k] for k in keys], [grads[k] for k in keys], opt_state, lr, betas, 1e-8, weight_decay)
    return logits.detach(), loss.detach(), grads


class LearningRateScheduler:
    """torchlib/utils.py:37-89 (log_linear / log_cosine with optional restarts)."""

    def __init__(self, total_epochs, log_start_lr, log_end_lr, schedule_plan="log_linear", restarts=None):
        if restarts == 0:
            restarts = None
        self.total_epochs = total_epochs if not restarts else total_epochs / (restarts + 1)
        if schedule_plan == "log_linear":
            self.calc_lr = lambda epoch: np.power(
                10, ((log_end_lr - log_start_lr) / self.total_epochs) * epoch + log_start_lr)
        elif schedule_plan == "log_cosine":
            self.calc_lr = lambda epoch: np.power(
                10, (np.cos(np.pi * (epoch / self.total_epochs)) / 2.0 + 0.5) * abs(log_start_lr - log_end_lr)
                + log_end_lr)
        else:
            raise NotImplementedError(schedule_plan)

    def get_lr(self, epoch):
        return self.calc_lr(epoch % self.total_epochs)


def fedavg_plain(state_dicts, weights=None):
    """aggregation(..., secure=False) (torchlib/utils.py:1062-1072, 1087-1090): per key (skipping
    num_batches_tracked) sum_k w_k * theta_k, divided by K when unweighted."""
    out = OrderedDict()
    K = len(state_dicts)
    for key in state_dicts[0]:
        if "num_batches_tracked" in key:
            continue
        stack = [sd[key] * (weights[i] if weights else 1) for i, sd in enumerate(state_dicts)]
        s = torch.sum(torch.stack(stack), dim=0)
        out[key] = s if weights else s / K
    return out


def fix_encode(x, base=10, precision_fractional=16):
    """FixedPrecisionTensor.fix_precision (precision.py:117-132): float32 multiply, .long()."""
    return (x * base ** precision_fractional).long()


def fix_decode(q, base=10, precision_fractional=16):
    """float_precision (precision.py:134-144)."""
    return q.float() / (base ** precision_fractional)


def fedavg_secure(state_dicts, weights=None, precision_fractional=16, base=10):
    """aggregation(..., secure=True) (torchlib/utils.py:1046-1060, 1079-1085): encode each
    (weighted) tensor to the 2^64 ring, share, sum shares, reconstruct, decode.  Additive sharing
    commutes with ring addition, so the reconstructed sum equals the wrapping sum of encodings."""
    out = OrderedDict()
    K = len(state_dicts)
    for key in state_dicts[0]:
        if "num_batches_tracked" in key:
            continue
        enc = [fix_encode(sd[key] * (weights[i] if weights else 1), base, precision_fractional)
               for i, sd in enumerate(state_dicts)]
        s = torch.sum(torch.stack(enc), dim=0)  # int64, wraps
        dec = fix_decode(s, base, precision_fractional)
        out[key] = dec if weights else dec / K
    return out


def dp_gradients(sd, x, target, max_grad_norm=1.0, noise_multiplier=1.3, noise=None, pooling="max", bf16_storage=False,
                 return_per_sample=False):
    """DP-SGD gradient as pytorch-dp 0.1b1's PrivacyEngine computes it for the optimizer
    (parameter values: train.py:325-334): per-sample gradients of each sample's own loss, flat L2
    clip to C with factor min(1, C / (norm + 1e-6)), sum, + N(0, (noise_multiplier*C)^2), / batch.
    pytorch-dp is not in the reference tree (environment_torch.yml:136): the clip / noise rule is restated from its
    published algorithm ("parity unpinned" for that rule).  The per-sample gradients under it ARE pinned: the
    GroupNorm network makes samples independent, so they are batch-of-1 gradients, and
    tests/golden/make_train_golden.py (mint_dp) checks them against the reference's model class
    (norm_layer=GroupNorm) differentiated by torch.func.vmap(grad) — tests/golden/dp_ref.npz.
    `noise`: dict key -> standard-normal tensor (explicit randomness).  Returns (grads, norms, clip).
    `bf16_storage` (tests only): the forward pass rounds to bf16 where the bf16 engine stores (see forward())."""
    keys = param_keys(sd)
    B = x.shape[0]
    per = []
    for n in range(B):
        for k in keys:
            sd[k].requires_grad_(True)
            sd[k].grad = None
        logits = forward(sd, x[n:n + 1], True, pooling, x.shape[-1], bf16_storage=bf16_storage)
        F.cross_entropy(logits, target[n:n + 1]).backward()
        per.append(OrderedDict((k, sd[k].grad.detach().clone()) for k in keys))
    for k in keys:
        sd[k].requires_grad_(False)
    norms = torch.stack([torch.sqrt(sum((g[k].double() ** 2).sum() for k in keys)) for g in per])
    clip = torch.clamp(max_grad_norm / (norms + 1e-6), max=1.0)
    out = OrderedDict()
    for k in keys:
        s = sum(clip[n].to(per[n][k].dtype) * per[n][k] for n in range(B))
        if noise is not None:
            s = s + noise[k] * (noise_multiplier * max_grad_norm)
        out[k] = s / B
    if return_per_sample:
        return out, norms, clip, per
    return out, norms, clip
