"""CPU: the training oracle against the golden vectors minted from the reference's own
torchlib/models.py (tests/golden/make_train_golden.py), and the host-side structure code."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import train_oracle as O
from primia_amd import resnet_spec as rs


def _summary(t, n=8):
    f = t.detach().double().flatten()
    return np.array([f.norm().item(), f.sum().item()] + f[:n].tolist() + [0.0] * max(0, n - f.numel()))


def replay(golden_dir, name, pooling, optimizer, lr, wd, cw, soft):
    """Re-run the oracle from the seeds stored in the fixture; yields per-step results."""
    gold = np.load(os.path.join(golden_dir, f"train_{name}.npz"))
    seed, batch, size, steps = [int(v) for v in gold["meta"]]
    torch.manual_seed(seed)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, pooling))
    g = torch.Generator().manual_seed(seed + 1)
    cwt = torch.tensor(cw, dtype=torch.float32) if cw else None
    st = {}
    for step in range(steps):
        x = torch.randn(batch, 3, size, size, generator=g)
        if soft:
            y = torch.rand(batch, 3, generator=g)
            y = y / y.sum(1, keepdim=True)
        else:
            y = torch.randint(0, 3, (batch,), generator=g)
        assert np.allclose(gold[f"s{step}.x_sum"], [x.double().sum().item(), x.double().abs().sum().item()])
        logits, loss, grads = O.train_step(sd, x, y, lr, wd, cwt, soft, pooling, optimizer, st, betas=(0.5, 0.99))
        yield gold, step, x, y, sd, logits, loss, grads


CASES = {
    "sgd_hard_64": ("max", "SGD", 1e-2, 5e-4, [0.5, 1.0, 2.0], False),
    "adam_soft_64": ("avg", "Adam", 1e-3, 5e-4, [0.5, 1.0, 2.0], True),
    "sgd_hard_224": ("max", "SGD", 1e-4, 5e-4, None, False),
}


@pytest.mark.parametrize("name", ["sgd_hard_64", "adam_soft_64", "sgd_hard_224"])
def test_oracle_matches_reference_golden(golden_dir, name):
    for gold, step, x, y, sd, logits, loss, grads in replay(golden_dir, name, *CASES[name]):
        # same ATen kernels, same order as the reference -> expect (near) bit equality; allow 1e-6
        assert np.allclose(logits.numpy(), gold[f"s{step}.logits"], rtol=1e-6, atol=1e-7)
        assert abs(loss.item() - float(gold[f"s{step}.loss"])) <= 1e-6 * abs(float(gold[f"s{step}.loss"]))
        for k, g in grads.items():
            assert np.allclose(_summary(g), gold[f"s{step}.grad.{k}"], rtol=1e-5, atol=1e-9), k
        for k, v in sd.items():
            if not k.endswith("num_batches_tracked"):
                assert np.allclose(_summary(v), gold[f"s{step}.post.{k}"], rtol=1e-6, atol=1e-9), k


def test_dp_gradients_against_reference_model(golden_dir):
    """T10: per-sample norms, clip factors and the clipped mean gradient of the oracle's batch-of-1 loop against
    the fixture minted from the reference's model class (norm_layer=GroupNorm) differentiated by vmap(grad)."""
    gold = np.load(os.path.join(golden_dir, "dp_ref.npz"))
    seed, batch, size = [int(v) for v in gold["meta"]]
    torch.manual_seed(seed)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"), "group")
    assert np.allclose(_summary(sd["conv1.weight"]), gold["init.conv1.weight"], rtol=1e-6)
    assert np.allclose(_summary(sd["fc.weight"]), gold["init.fc.weight"], rtol=1e-6)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, 3, (batch,), generator=g)
    assert np.allclose(gold["x_sum"], [x.double().sum().item(), x.double().abs().sum().item()])
    grads, norms, clip = O.dp_gradients(sd, x, y, float(gold["C"]), 0.0, None)
    assert np.allclose(norms.numpy(), gold["norms"], rtol=1e-5)
    assert np.allclose(clip.numpy(), gold["clip"], rtol=1e-5)
    for k, gk in grads.items():
        want = gold["grad." + k]
        assert np.allclose(_summary(gk), want, rtol=1e-4, atol=1e-5 * want[0] + 1e-12), k


def test_spec_matches_reference_inventory():
    spec = rs.resnet18_spec()
    keys = rs.state_dict_keys(spec)
    assert len(keys) == 122
    n_params = sum(int(np.prod(s)) for _, s in rs.param_entries(spec))
    n_buf = sum(int(np.prod(s)) for _, s in rs.buffer_entries(spec))
    assert n_params == 11178051
    assert n_params + n_buf == 11187651   # FedAvg payload, §8a T9
    assert len([k for k in keys if not k.endswith("num_batches_tracked")]) == 102


def test_lr_schedule(golden_dir):
    gold = np.load(os.path.join(golden_dir, "lr_schedule.npz"))
    from primia_amd.torchlib_compat import LearningRateScheduler

    # every value below was produced by the reference's own class (tests/golden/make_train_golden.py)
    for plan in ("log_linear", "log_cosine"):
        for restarts in (0, 1, 3):
            for cls in (LearningRateScheduler, O.LearningRateScheduler):
                s = cls(40, -4, -5, schedule_plan=plan, restarts=restarts)
                got = np.array([s.get_lr(e) for e in range(40)])
                assert np.array_equal(got, gold[f"{plan}.r{restarts}"]), (plan, restarts, cls)
    assert abs(gold["r0"][0] - 1e-4) < 1e-12
    with pytest.raises(NotImplementedError):
        LearningRateScheduler(40, -4, -5, schedule_plan="step")

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0}]

    lr = LearningRateScheduler(40, -4, -5).adjust_learning_rate(Opt, 7)
    assert lr == gold["log_linear.r0"][7] and all(g["lr"] == lr for g in Opt.param_groups)


def test_soft_cross_entropy_matches_reference_class(golden_dir):
    """Cross_entropy_one_hot (torchlib/utils.py:404-441): losses and gradients minted by executing the reference class."""
    gold = np.load(os.path.join(golden_dir, "lr_schedule.npz"))
    out, tgt = torch.from_numpy(gold["ce.out"]), torch.from_numpy(gold["ce.target"])
    for wname, w in (("w", torch.tensor([0.5, 1.0, 2.0])), ("nw", None)):
        for red in ("mean", "sum"):
            o = out.clone().requires_grad_(True)
            loss = O.cross_entropy_one_hot(o, tgt, w, red)
            loss.backward()
            assert loss.item() == gold[f"ce.{wname}.{red}.loss"].item()
            assert np.array_equal(o.grad.numpy(), gold[f"ce.{wname}.{red}.grad"])


def test_fedavg_oracle_plain_and_secure():
    torch.manual_seed(0)
    sds = []
    for k in range(3):
        sds.append(OrderedDict(a=torch.randn(5, 4) * 0.1, b=torch.randn(7), **{"bn.num_batches_tracked": torch.tensor(3)}))
    plain = O.fedavg_plain(sds)
    assert torch.allclose(plain["a"], (sds[0]["a"] + sds[1]["a"] + sds[2]["a"]) / 3)
    assert "bn.num_batches_tracked" not in plain
    w = [0.2, 0.3, 0.5]
    wp = O.fedavg_plain(sds, w)
    assert torch.allclose(wp["b"], sum(wi * sd["b"] for wi, sd in zip(w, sds)))
    for pf in (16, 3):
        sec = O.fedavg_secure(sds, None, pf)
        assert torch.allclose(sec["a"], plain["a"], atol=10.0 ** -pf * 3 + 1e-7)


def _fedavg_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "fedavg_ref.npz"))
    keys = [k.split("/", 1)[1] for k in z.files if k.startswith("in0/")]
    sds = [OrderedDict((k, torch.from_numpy(z[f"in{c}/{k}"])) for k in keys) for c in range(3)]
    return z, keys, sds


def test_fedavg_oracle_matches_reference_aggregation(golden_dir):
    """aggregation() of torchlib/utils.py:1000-1092 was EXECUTED (plain and secure, unweighted and weighted, K = 3
    workers sharing between themselves) when tests/golden/fedavg_ref.npz was minted; the oracle must agree bit for bit."""
    z, keys, sds = _fedavg_fixture(golden_dir)
    w = [0.2, 0.3, 0.5]
    for tag, fn in (("plain.u.p0", lambda: O.fedavg_plain(sds)), ("plain.w.p0", lambda: O.fedavg_plain(sds, w)),
                    ("secure.u.p3", lambda: O.fedavg_secure(sds, None, 3)), ("secure.u.p16", lambda: O.fedavg_secure(sds, None, 16)),
                    ("secure.w.p3", lambda: O.fedavg_secure(sds, w, 3)), ("secure.w.p16", lambda: O.fedavg_secure(sds, w, 16))):
        got = fn()
        for k in keys:
            if k.endswith("num_batches_tracked"):
                assert k not in got
                continue
            assert np.array_equal(got[k].numpy(), z[f"{tag}/{k}"]), (tag, k)
