"""CPU: the data-path oracle (oracle/datapipe_oracle.py) against the vectors minted from the reference's
own MixUp / To_one_hot / calc_mean_std (tests/golden/make_datapipe_golden.py), plus the host-side splitters."""
import os
import random

import numpy as np
import torch

from oracle import datapipe_oracle as D

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "datapipe.npz"))


def _mix_cases():
    return sorted({k.split(".")[1] for k in GOLD.files if k.startswith("mixup.")})


def test_mixup_oracle_matches_reference_vectors():
    assert len(_mix_cases()) == 8
    for name in _mix_cases():
        L, lam, p, seed, as_tuple = GOLD[f"mixup.{name}.meta"]
        x, y = torch.from_numpy(GOLD[f"mixup.{name}.x"]), torch.from_numpy(GOLD[f"mixup.{name}.y"])
        xin = tuple(t.unsqueeze(0) for t in x) if as_tuple else x
        yin = tuple(t.unsqueeze(0) for t in y) if as_tuple else y
        ox, oy = D.mixup(xin, yin, None if lam < 0 else float(lam), float(p), rng=random.Random(int(seed)))
        assert torch.equal(ox, torch.from_numpy(GOLD[f"mixup.{name}.out_x"])), name
        assert torch.equal(oy, torch.from_numpy(GOLD[f"mixup.{name}.out_y"])), name


def test_one_hot_and_mean_std_oracle_match_reference_vectors():
    for name in ("int", "list", "scalar_tensor", "vector"):
        arg = GOLD[f"onehot.{name}.in"]
        arg = int(arg) if name == "int" else (arg.tolist() if name == "list" else torch.from_numpy(arg))
        assert torch.equal(D.to_one_hot(arg, 3), torch.from_numpy(GOLD[f"onehot.{name}.out"])), name
    for name in ("rgb", "gray", "other"):
        m, s = D.calc_mean_std(torch.from_numpy(GOLD[f"meanstd.{name}.data"]))
        assert torch.equal(m, torch.from_numpy(GOLD[f"meanstd.{name}.mean"])), name
        assert torch.equal(s, torch.from_numpy(GOLD[f"meanstd.{name}.std"])), name


def test_splitters():
    from primia_amd.datapipe import iid_round_robin_split, label_skew_split

    n, k = 5163, 3          # the reference's training set size, three hospitals
    parts = iid_round_robin_split(n, k)
    assert parts == D.iid_round_robin_split(n, k)
    assert sorted(sum(parts, [])) == list(range(n)) and [len(p) for p in parts] == [1721, 1721, 1721]
    rng = random.Random()
    rng.seed(0)
    ref = list(range(n))
    rng.shuffle(ref)
    assert parts[1][:5] == ref[1::3][:5]
    labels = np.random.default_rng(1).integers(0, 3, size=n)
    shards = label_skew_split(labels, 8, alpha=0.3, seed=5)
    assert sorted(sum(shards, [])) == list(range(n))                  # a partition
    assert shards == label_skew_split(labels, 8, alpha=0.3, seed=5)    # deterministic
    props = np.array([[np.mean(labels[s] == c) if len(s) else 0.0 for c in range(3)] for s in shards])
    assert props.std(axis=0).max() > 0.15                              # visibly non-IID
    iid = np.array([[np.mean(labels[s] == c) for c in range(3)] for s in iid_round_robin_split(n, 8)])
    assert iid.std(axis=0).max() < 0.03


def test_exchange_mean_std_oracle():
    means = [torch.tensor([0.5, 0.25]), torch.tensor([0.7, 0.35]), torch.tensor([0.1003, 0.9])]
    stds = [torch.tensor([0.2, 0.2]), torch.tensor([0.3, 0.1]), torch.tensor([0.25, 0.15])]
    m, s = D.exchange_mean_std(means, stds)
    # fixed point with 3 fractional digits: every term is truncated to 1e-3 before the sum
    assert torch.allclose(m, torch.tensor([(0.5 + 0.7 + 0.1) / 3, (0.25 + 0.35 + 0.9) / 3]), atol=1e-6)
    assert torch.allclose(s, torch.tensor([0.25, 0.15]), atol=1e-6)


def test_exchange_mean_std_against_reference_tensors(golden_dir):
    """tests/golden/mean_std_ref.npz was produced by the reference's FixedPrecisionTensor / AdditiveSharingTensor
    classes running utils.py:764-794's chain (make_secure_ref_golden.py mint_mean_std)."""
    import os

    gold = np.load(os.path.join(golden_dir, "mean_std_ref.npz"))
    for tag in ("w2c1", "w3c3", "w5c3"):
        means = [torch.from_numpy(v) for v in gold[tag + "/means"]]
        stds = [torch.from_numpy(v) for v in gold[tag + "/stds"]]
        m, s = D.exchange_mean_std(means, stds)
        assert np.array_equal(m.numpy(), gold[tag + "/mean"]) and np.array_equal(s.numpy(), gold[tag + "/std"]), tag


def test_iid_split_against_reference_script(golden_dir):
    """The dealing statements of distribute_data.py, executed at mint time (make_datapipe_golden.reference_split)."""
    import os

    from primia_amd.datapipe import iid_round_robin_split

    gold = np.load(os.path.join(golden_dir, "datapipe.npz"))
    for n_items, nw in ((5163, 3), (100, 8), (7, 2)):
        want = gold[f"split.{n_items}.{nw}"]
        for fn in (D.iid_round_robin_split, iid_round_robin_split):
            parts = fn(n_items, nw)
            got = np.array([len(s) for s in parts] + [v for s in parts for v in s[:8]])
            assert np.array_equal(got, want), (n_items, nw)
