"""Mint the training golden vectors from the REFERENCE's own model code.

Run in the build container only (needs /root/reference):
    python tests/golden/make_train_golden.py

It imports /root/reference/torchlib/models.py (with a 2-line stub for the unused `syft.Plan`
import), runs seeded forward / backward / optimizer steps on torch-CPU, checks that
oracle/train_oracle.py reproduces every number bit-for-bit, and stores small fixtures
(seeds, logits, losses, per-tensor norms and leading values — never whole models).
The reference ships no tests or known-answer vectors, so these are the pin.
"""
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import train_oracle as O  # noqa: E402


def load_reference_models():
    stub = types.ModuleType("syft")
    stub.Plan = object
    sys.modules["syft"] = stub
    spec = importlib.util.spec_from_file_location("ref_models", "/root/reference/torchlib/models.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def extract(path, names, glob):
    """Execute exactly the named top-level definitions of a reference file in place (the file itself cannot be
    imported: it needs syft / albumentations / torchvision)."""
    import ast

    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), glob)
    return glob


def load_reference_utils():
    from typing import Optional

    g = {"torch": torch, "np": np, "Optional": Optional}
    extract("/root/reference/torchlib/utils.py", ["LearningRateScheduler", "Cross_entropy_one_hot"], g)
    return g


REF_UTILS = None


def mint_dp(ref_models):
    """T10 (DP-SGD): the per-sample gradients are taken from the REFERENCE's model class with its norm_layer hook
    (torchlib/models.py:355,362-364 -> GroupNorm(32, C)), differentiated per sample by torch.func.vmap(grad) — an
    implementation independent of the oracle's batch-of-1 loop.  The clip / sum / divide rule on top of them is
    pytorch-dp 0.1b1's (not in the tree: that part stays restated from its published algorithm)."""
    from torch.func import functional_call, grad, vmap

    seed, batch, size, C = 11, 6, 64, 1.0
    torch.manual_seed(seed)
    model = ref_models.resnet18(pretrained=False, num_classes=3, in_channels=3, adptpool=False, input_size=size,
                                pooling="max", norm_layer=lambda c: torch.nn.GroupNorm(32, c))
    model.train()
    sd = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, 3, (batch,), generator=g)
    params = {k: v.detach() for k, v in model.named_parameters()}

    def loss_fn(p, xi, yi):
        return torch.nn.functional.cross_entropy(functional_call(model, p, (xi[None],)), yi[None])

    per = vmap(grad(loss_fn), in_dims=(None, 0, 0))(params, x, y)
    norms = torch.sqrt(sum((per[k].double().flatten(1) ** 2).sum(1) for k in per))
    clip = torch.clamp(C / (norms + 1e-6), max=1.0)
    clipped = {k: (per[k].double() * clip.view(-1, *[1] * (per[k].dim() - 1))).sum(0) / batch for k in per}
    want, onorms, oclip = O.dp_gradients({k: v.clone() for k, v in sd.items()}, x, y, C, 0.0, None)
    assert torch.allclose(onorms, norms, rtol=1e-5) and torch.allclose(oclip, clip, rtol=1e-5)
    for k in clipped:
        assert (clipped[k] - want[k].double()).norm() <= 1e-5 * clipped[k].norm(), k
    out = {"meta": np.array([seed, batch, size]), "C": np.array(C), "norms": norms.numpy(), "clip": clip.numpy(),
           "x_sum": np.array([x.double().sum().item(), x.double().abs().sum().item()]),
           "init.conv1.weight": summary(sd["conv1.weight"]), "init.fc.weight": summary(sd["fc.weight"])}
    for k in clipped:
        out["grad." + k] = summary(clipped[k])
        out["sample0." + k] = summary(per[k][0])
    path = os.path.join(HERE, "dp_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def summary(t, n=8):
    f = t.detach().double().flatten()
    return np.array([f.norm().item(), f.sum().item()] + f[:n].tolist() + [0.0] * max(0, n - f.numel()))


def case(ref_models, name, seed, batch, size, pooling, optimizer, lr, wd, class_weight, soft, steps):
    torch.manual_seed(seed)
    model = ref_models.resnet18(pretrained=False, num_classes=3, in_channels=3, adptpool=False,
                                input_size=size, pooling=pooling)
    model.train()
    sd = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    g = torch.Generator().manual_seed(seed + 1)
    cw = torch.tensor(class_weight, dtype=torch.float32) if class_weight else None
    if optimizer == "SGD":
        opt = torch.optim.SGD(model.parameters(), lr=lr, weight_decay=wd)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=lr, betas=(0.5, 0.99), weight_decay=wd)
    opt_state = {}
    out = {"meta": np.array([seed, batch, size, steps])}
    for step in range(steps):
        x = torch.randn(batch, 3, size, size, generator=g)
        if soft:
            y = torch.rand(batch, 3, generator=g)
            y = y / y.sum(1, keepdim=True)
        else:
            y = torch.randint(0, 3, (batch,), generator=g)
        # ---- reference ------------------------------------------------------------------------
        opt.zero_grad()
        pred = model(x)
        if soft:
            # the reference's own Cross_entropy_one_hot (torchlib/utils.py:404-441), executed from its file
            loss = REF_UTILS["Cross_entropy_one_hot"](weight=cw)(pred, y)
        else:
            loss = torch.nn.CrossEntropyLoss(weight=cw, reduction="mean")(pred, y)
        loss.backward()
        ref_grads = OrderedDict((k, p.grad.detach().clone()) for k, p in model.named_parameters())
        opt.step()
        # ---- oracle on the same inputs ------------------------------------------------------------
        logits, oloss, ograds = O.train_step(sd, x, y, lr, wd, cw, soft, pooling, optimizer, opt_state,
                                             betas=(0.5, 0.99))
        assert torch.equal(logits, pred.detach()), "oracle logits differ from reference"
        assert torch.equal(oloss, loss.detach()), "oracle loss differs from reference"
        for k in ref_grads:
            assert torch.equal(ograds[k], ref_grads[k]), f"oracle grad {k} differs"
        rsd = model.state_dict()
        for k in rsd:
            if optimizer == "SGD":
                assert torch.equal(sd[k], rsd[k]), f"oracle post-step {k} differs"
            else:
                # The oracle restates torch-1.4 Adam (mul_/add_/addcdiv_); torch 2.x in this container
                # uses lerp_/foreach forms that round differently in the last bit.
                assert torch.allclose(sd[k].float(), rsd[k].float(), rtol=1e-5, atol=1e-7), f"post-step {k}"
        out[f"s{step}.logits"] = pred.detach().numpy().astype(np.float64)
        out[f"s{step}.loss"] = np.array(loss.item())
        out[f"s{step}.x_sum"] = np.array([x.double().sum().item(), x.double().abs().sum().item()])
        for k in ref_grads:
            out[f"s{step}.grad.{k}"] = summary(ref_grads[k])
        for k in rsd:
            if not k.endswith("num_batches_tracked"):
                out[f"s{step}.post.{k}"] = summary(sd[k])  # == reference for SGD; 1.4-style Adam otherwise
    path = os.path.join(HERE, f"train_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    global REF_UTILS
    REF_UTILS = load_reference_utils()
    ref = load_reference_models()
    mint_dp(ref)
    # the reference's own config values: seed 42 / lr 1e-4 / wd 5e-4 (pneumonia-resnet-pretrained.ini)
    case(ref, "sgd_hard_224", 42, 4, 224, "max", "SGD", 1e-4, 5e-4, None, False, 2)
    case(ref, "sgd_hard_64", 42, 8, 64, "max", "SGD", 1e-2, 5e-4, [0.5, 1.0, 2.0], False, 2)
    case(ref, "adam_soft_64", 7, 8, 64, "avg", "Adam", 1e-3, 5e-4, [0.5, 1.0, 2.0], True, 2)
    # LearningRateScheduler (torchlib/utils.py:37-89): the reference's class, executed from its file
    lrs = {}
    for plan in ("log_linear", "log_cosine"):
        for restarts in (0, 1, 3):
            ref_s = REF_UTILS["LearningRateScheduler"](40, -4, -5, schedule_plan=plan, restarts=restarts)
            mine = O.LearningRateScheduler(40, -4, -5, schedule_plan=plan, restarts=restarts)
            vals = np.array([ref_s.get_lr(e) for e in range(40)])
            assert np.array_equal(vals, np.array([mine.get_lr(e) for e in range(40)])), (plan, restarts)
            lrs[f"{plan}.r{restarts}"] = vals
    lrs["r0"], lrs["r1"] = lrs["log_linear.r0"], lrs["log_linear.r1"]
    # soft cross entropy alone: weighted / unweighted, both reductions
    g = torch.Generator().manual_seed(5)
    out, tgt = torch.randn(6, 3, generator=g), torch.rand(6, 3, generator=g)
    tgt = tgt / tgt.sum(1, keepdim=True)
    lrs["ce.out"], lrs["ce.target"] = out.numpy(), tgt.numpy()
    for wname, w in (("w", torch.tensor([0.5, 1.0, 2.0])), ("nw", None)):
        for red in ("mean", "sum"):
            o = out.clone().requires_grad_(True)
            loss = REF_UTILS["Cross_entropy_one_hot"](reduction=red, weight=w)(o, tgt)
            loss.backward()
            o2 = out.clone().requires_grad_(True)
            mine = O.cross_entropy_one_hot(o2, tgt, w, red)
            mine.backward()
            assert torch.equal(mine, loss.detach()) and torch.equal(o.grad, o2.grad), (wname, red)
            lrs[f"ce.{wname}.{red}.loss"] = np.array(loss.item(), dtype=np.float32)
            lrs[f"ce.{wname}.{red}.grad"] = o.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "lr_schedule.npz"), **lrs)


if __name__ == "__main__":
    main()
