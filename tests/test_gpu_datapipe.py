"""GPU: MixUp / To_one_hot / calc_mean_std through the C ABI against the reference-derived vectors (bit-exact
for MixUp and one-hot, 1e-6 for the statistics) and against the oracle at the training batch size."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import datapipe_oracle as D  # noqa: E402
from primia_amd import datapipe as P  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "datapipe.npz"))


def test_mixup_matches_reference_vectors(cuda):
    for name in sorted({k.split(".")[1] for k in GOLD.files if k.startswith("mixup.")}):
        L, lam, p, seed, as_tuple = GOLD[f"mixup.{name}.meta"]
        x, y = torch.from_numpy(GOLD[f"mixup.{name}.x"]).to(cuda), torch.from_numpy(GOLD[f"mixup.{name}.y"]).to(cuda)
        xin = tuple(t.unsqueeze(0) for t in x) if as_tuple else x
        yin = tuple(t.unsqueeze(0) for t in y) if as_tuple else y
        random.seed(int(seed))
        ox, oy = P.MixUp(λ=None if lam < 0 else float(lam), p=float(p))((xin, yin))
        assert torch.equal(ox.cpu(), torch.from_numpy(GOLD[f"mixup.{name}.out_x"])), name
        assert torch.equal(oy.cpu(), torch.from_numpy(GOLD[f"mixup.{name}.out_y"])), name


def test_one_hot_and_mean_std_match_reference_vectors(cuda):
    for name in ("int", "list", "scalar_tensor", "vector"):
        arg = GOLD[f"onehot.{name}.in"]
        arg = int(arg) if name == "int" else (arg.tolist() if name == "list" else torch.from_numpy(arg))
        assert torch.equal(P.To_one_hot(3)(arg).cpu(), torch.from_numpy(GOLD[f"onehot.{name}.out"])), name
    for name in ("rgb", "gray", "other"):
        m, s = P.calc_mean_std(torch.from_numpy(GOLD[f"meanstd.{name}.data"]))
        # tolerance: the kernel accumulates sums in fp64, torch uses a float Welford pass
        assert torch.allclose(m.cpu(), torch.from_numpy(GOLD[f"meanstd.{name}.mean"]), rtol=1e-6, atol=1e-7), name
        assert torch.allclose(s.cpu(), torch.from_numpy(GOLD[f"meanstd.{name}.std"]), rtol=1e-6, atol=1e-7), name


@pytest.mark.parametrize("L", [256, 255])
def test_mixup_full_batch_bit_exact_vs_oracle(cuda, L):
    """BASELINE batch: 256 x 3 x 224 x 224 (and an odd batch with a pass-through sample)."""
    g = torch.Generator().manual_seed(L)
    x = torch.randn(L, 3, 224, 224, generator=g)
    y = D.to_one_hot(torch.randint(0, 3, (L,), generator=g), 3)
    ox, oy = D.mixup(x, y, 0.3141, 0.0)
    px, py = P.MixUp(λ=0.3141, p=0.0)((x.to(cuda), y.to(cuda)))
    assert torch.equal(px.cpu(), ox) and torch.equal(py.cpu(), oy)


def test_mean_std_of_a_large_dataset(cuda):
    g = torch.Generator().manual_seed(3)
    data = torch.rand(64, 3, 224, 224, generator=g) * torch.tensor([1.0, 2.0, 0.5]).view(1, 3, 1, 1) + 0.1
    m, s = P.calc_mean_std(data)
    om, os_ = D.calc_mean_std(data.double())
    assert torch.allclose(m.cpu().double(), om, rtol=1e-6) and torch.allclose(s.cpu().double(), os_, rtol=1e-6)


def test_train_with_mixup_matches_oracle_step(cuda):
    """torchlib_compat.train with args.mixup (utils.py:1249-1267): one-hot + MixUp on the device, soft-label
    loss; one SGD step against the oracle run on the oracle-mixed batch (fp32 engine, 1e-5 / 1e-2 bounds as
    in test_gpu_train_step)."""
    import types

    from oracle import train_oracle as O
    from primia_amd import resnet_spec as rs
    from primia_amd.engine import ResNet18Engine
    from primia_amd.torchlib_compat import train

    B, size, lam = 4, 64, 0.37
    torch.manual_seed(11)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, size, "max"))
    eng = ResNet18Engine(B, 3, 3, size, "max", dtype=torch.float32, device=cuda)
    eng.load_state_dict(sd)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2 * B, 3, size, size, generator=g)       # mixup_prob = 1.0 doubles the loader's batch
    t = torch.randint(0, 3, (2 * B,), generator=g)
    args = types.SimpleNamespace(mixup=True, mixup_lambda=lam, mixup_prob=1.0, optimizer="SGD", weight_decay=5e-4,
                                 log_interval=1)
    random.seed(0)
    train(args, eng, cuda, [(x.to(cuda), t.to(cuda))], {"lr": 1e-2}, 1, None, 3, verbose=False)
    ox, oy = D.mixup(x, D.to_one_hot(t, 3), lam, 1.0, rng=random.Random(0))
    osd = {k: v.clone() for k, v in sd.items()}
    O.train_step(osd, ox, oy, 1e-2, 5e-4, None, soft=True, pooling="max")
    got = eng.state_dict()
    for k in ("fc.weight", "layer4.1.conv2.weight", "layer1.0.bn1.weight", "conv1.weight", "bn1.running_mean"):
        a, b = got[k].double(), osd[k].double()
        assert (a - b).norm() <= 1e-4 * b.norm() + 1e-7, k


def test_federated_registration_matches_reference_loop(cuda):
    """primia_amd.datapipe.register_federated (the blend in primia_mixup) against the output of the reference's own
    registration loop (torchlib/utils.py:694-734, executed when tests/golden/datapipe.npz was minted): bit-exact."""
    for name in sorted({k.split(".")[1] for k in GOLD.files if k.startswith("reg.")}):
        n, reps, mix, lam, p, seed = GOLD[f"reg.{name}.meta"]
        xs = torch.from_numpy(GOLD[f"reg.{name}.x"]).to(cuda)
        ys = P.To_one_hot(3)(torch.from_numpy(GOLD[f"reg.{name}.labels"]))
        orders = GOLD[f"reg.{name}.orders"].tolist()
        mixer = P.MixUp(λ=None if lam < 0 else float(lam), p=float(p)) if mix else None
        random.seed(int(seed))
        d, t = P.register_federated(xs, ys, orders, mixer)
        assert torch.equal(d.cpu(), torch.from_numpy(GOLD[f"reg.{name}.data"])), name
        assert torch.equal(t.cpu(), torch.from_numpy(GOLD[f"reg.{name}.targets"])), name


def test_class_weights_match_reference_function(cuda):
    from types import SimpleNamespace

    for name, fed in (("fed_soft", True), ("vanilla_hard", False), ("fed_empty", True)):
        nw = int(GOLD[f"cw.{name}.n"][0])
        loaders = {f"w{w}": [(None, torch.from_numpy(t).to(cuda)) for t in GOLD[f"cw.{name}.w{w}"]] for w in range(nw)}
        args = SimpleNamespace(train_federated=fed, mixup=fed, weight_classes=True, batch_size=4)
        arg = loaders if fed else (list(loaders.values())[0] if loaders else [])
        with __import__("warnings").catch_warnings():
            __import__("warnings").simplefilter("ignore")
            cw = P.calc_class_weights(args, arg, 3)
        assert torch.equal(cw.cpu(), torch.from_numpy(GOLD[f"cw.{name}.cw"])), name


def test_registered_loader_feeds_soft_targets_and_class_weights_into_the_step(cuda):
    """The reference's default INI in federated mode (mixup = yes): the loader a client trains on yields soft targets of
    registered (blended) samples, and the engine's soft-label loss takes the class weights."""
    from types import SimpleNamespace

    from primia_amd.engine import ResNet18Engine
    from primia_amd.imagefolder import DeviceLoader, register

    args = SimpleNamespace(train_federated=True, mixup=True, weight_classes=True, mixup_lambda=None, mixup_prob=0.9,
                           repetitions_dataset=2, batch_size=4)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(6, 3, 32, 32, generator=g).to(cuda)
    y = torch.randint(0, 3, (6,), generator=g).to(cuda)
    random.seed(5)
    data, tg = register([x, x], y, args, 3, seed=1)        # two walks over the dataset (repetitions_dataset = 2)
    assert data.shape == (12, 3, 32, 32) and tg.shape == (12, 3)
    assert torch.allclose(tg.sum(1), torch.ones(12, device=cuda), atol=1e-6)
    assert (tg.max(1)[0] < 1.0).any()                       # some targets are genuinely mixed
    loader = DeviceLoader(data, tg, 4, True, 1)
    cw = P.calc_class_weights(args, {"w": loader}, 3)
    eng = ResNet18Engine(4, 3, 3, 32, "max", dtype=torch.float32, device=cuda)
    torch.manual_seed(0)
    eng.init_weights()
    eng.class_weight = cw.to(cuda).float().contiguous()
    xb, tb = next(iter(loader))
    logits = eng.forward(xb)
    loss = eng.loss_backward(tb, soft=True)
    ls = torch.log_softmax(logits.float(), 1)
    want = ((tb * eng.class_weight).sum(1) * (-(tb * ls).sum(1))).mean()       # Cross_entropy_one_hot (utils.py:404-441)
    assert abs(loss.item() - want.item()) < 1e-5 * max(1.0, abs(want.item()))
