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


def _reg_cases():
    return sorted({k.split(".")[1] for k in GOLD.files if k.startswith("reg.")})


def test_federated_registration_oracle_matches_reference_loop():
    """setup_pysyft's registration loop (torchlib/utils.py:694-734), executed from the reference's file when the
    fixture was minted: MixUp pairing with the previous unmixed sample, the random() draw order, repetitions."""
    assert _reg_cases() == ["mix_always", "mix_fixed", "mix_rand_lambda", "onehot_only"]
    for name in _reg_cases():
        n, reps, mix, lam, p, seed = GOLD[f"reg.{name}.meta"]
        xs = torch.from_numpy(GOLD[f"reg.{name}.x"])
        ys = D.to_one_hot(torch.from_numpy(GOLD[f"reg.{name}.labels"]), 3)
        orders = GOLD[f"reg.{name}.orders"].tolist()
        d, t = D.register_federated(list(xs), list(ys), orders, bool(mix), None if lam < 0 else float(lam), float(p),
                                    rng=random.Random(int(seed)))
        assert torch.equal(d, torch.from_numpy(GOLD[f"reg.{name}.data"])), name
        assert torch.equal(t, torch.from_numpy(GOLD[f"reg.{name}.targets"])), name
        assert d.shape[0] == int(n) * int(reps)


def test_class_weights_oracle_matches_reference_function():
    for name, soft in (("fed_soft", True), ("vanilla_hard", False), ("fed_empty", True)):
        nw = int(GOLD[f"cw.{name}.n"][0])
        loaders = [[(None, torch.from_numpy(t)) for t in GOLD[f"cw.{name}.w{w}"]] for w in range(nw)]
        cw = D.calc_class_weights(loaders, 4, 3, soft)
        assert torch.equal(cw, torch.from_numpy(GOLD[f"cw.{name}.cw"])), name


def test_augmentation_oracle_properties():
    """oracle/augment_oracle.py: identities and invariants of the restated image arithmetic (its parity with cv2's
    binaries is unpinned — this checks the restatement against the definitions it follows)."""
    from oracle import augment_oracle as A

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    ident = A.inverse_affine_matrix((56 * 0.5 + 0.5, 40 * 0.5 + 0.5), 0, (0, 0), 1.0, 0)
    assert np.array_equal(A.affine_nearest(img, ident), img)
    sq = rng.integers(0, 256, size=(41, 41), dtype=np.uint8)
    # (about the geometric centre W / 2; torchvision 0.5 passes W / 2 + 0.5, which shifts a rotation by one pixel)
    rot = A.inverse_affine_matrix((20.5, 20.5), 90, (0, 0), 1.0, 0)
    assert np.array_equal(A.affine_nearest(sq, rot), np.rot90(sq, 1)) or np.array_equal(A.affine_nearest(sq, rot), np.rot90(sq, -1))
    assert np.array_equal(A.resize_crop(img, 40, 0, 0, 40)[:, :, 0].shape, (40, 40))
    assert np.array_equal(A.resize_crop(sq, 41, 0, 0, 41), sq)                       # same size: identity
    assert np.array_equal(A.box_blur(np.full((9, 9), 77, np.uint8), 5), np.full((9, 9), 77, np.uint8))
    g1 = A.gamma_table(1.0).astype(int)       # (i / 255) * 255 truncated: within one level of the identity, monotone
    assert np.abs(g1 - np.arange(256)).max() <= 1 and (np.diff(g1) >= 0).all()
    assert (A.gamma_table(0.8).astype(int) >= A.gamma_table(1.2).astype(int)).all()
    assert np.array_equal(A.brightness_table(1.0, 0.0), np.arange(256, dtype=np.uint8))
    flat = np.full((64, 64), 90, np.uint8)
    out = A.clahe_plane(flat, 1.0)
    assert out.min() == out.max()                                                    # a flat image stays flat
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 2, (64, 1))
    eq = A.clahe_plane(ramp, 40.0)
    assert eq.max() > ramp.max() and eq.shape == ramp.shape                          # contrast stretched
