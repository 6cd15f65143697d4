"""CPU: host-side mirror of the reference interface (Arguments, worker CSV, schedules)."""
import argparse
import configparser
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

from primia_amd.torchlib_compat import Arguments, read_websocket_config, matthews_corrcoef  # noqa: E402


def cmd(**kw):
    base = dict(training_name=None, save_file="x.csv", train_federated=True, unencrypted_aggregation=False,
                visdom=False, data_dir="data/train", cuda=True, websockets=False, encrypted_inference=False)
    base.update(kw)
    return argparse.Namespace(**base)


def test_arguments_from_reference_style_ini():
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "configs/torch/pneumonia-resnet-pretrained.ini"))
    a = Arguments(cmd(), cfg, mode="train", verbose=False)
    assert (a.batch_size, a.train_resolution, a.inference_resolution) == (200, 224, 224)
    assert a.optimizer == "Adam" and (a.beta1, a.beta2) == (0.5, 0.99)
    assert a.lr == 1e-4 and a.end_lr == 1e-5 and a.weight_decay == 5e-4 and a.seed == 42
    assert a.sync_every_n_batch == 3 and a.repetitions_dataset == 5 and a.epochs == 8  # 40 / 5 repetitions
    assert a.precision_fractional == 16 and a.weighted_averaging is False and a.keep_optim_dict is False
    assert a.mixup and a.mixup_prob == 0.9 and a.mixup_lambda is None and a.name == "default"
    # inference mode never reads the federated section and forces the train-only switches off
    b = Arguments(cmd(encrypted_inference=True), cfg, mode="inference", verbose=False)
    assert b.encrypted_inference and not b.train_federated and not hasattr(b, "sync_every_n_batch")
    # a required key that is missing raises, as configparser does for the reference
    cfg.remove_option("config", "weight_decay")
    with pytest.raises(configparser.NoOptionError):
        Arguments(cmd(), cfg, mode="train", verbose=False)


def test_mixup_prob_one_doubles_batch():
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "configs/torch/pneumonia-resnet-pretrained.ini"))
    cfg.set("augmentation", "mixup_prob", "1.0")
    assert Arguments(cmd(train_federated=False), cfg, verbose=False).batch_size == 400


def test_worker_csv():
    w = read_websocket_config(os.path.join(ROOT, "configs/websetting/config.csv"))
    assert [v["id"] for v in w.values()] == ["alice", "bob", "charlie", "crypto_provider"]
    assert w[1] == {"id": "alice", "host": "127.0.0.1", "port": 8777}
    inf = read_websocket_config(os.path.join(ROOT, "configs/websetting/config_inference.csv"))
    assert [v["id"] for v in inf.values()] == ["data_owner", "model_owner", "crypto_provider"]


def test_mcc_matches_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    g = torch.Generator().manual_seed(0)
    t = torch.randint(0, 3, (200,), generator=g).tolist()
    p = torch.randint(0, 3, (200,), generator=g).tolist()
    assert abs(matthews_corrcoef(t, p, 3) - sk.matthews_corrcoef(t, p)) < 1e-12


def _opener_worker(rank, port, tmp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from primia_amd.secure import DistOpener

    g = torch.Generator().manual_seed(5)
    s = [torch.randint(-2 ** 63, 2 ** 63 - 1, (1000,), generator=g, dtype=torch.int64) for _ in range(2)]
    op = DistOpener()
    got = op.open([s[0] if rank == 0 else None, s[1] if rank == 1 else None])
    torch.save(torch.equal(got, s[0] + s[1]), os.path.join(tmp, f"o{rank}.pt"))
    dist.destroy_process_group()


def test_dist_opener_two_parties(tmp_path):
    mp.spawn(_opener_worker, args=(29700 + os.getpid() % 1000, str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(os.path.join(str(tmp_path), f"o{r}.pt")) for r in range(2))


def test_torchlib_shim_and_checkpoint_argument_pickle(tmp_path):
    """A checkpoint's pickled `args` names torchlib.utils.Arguments (utils.py:1489): the shim package resolves
    it to the HIP-backed class, both for files we write and for a reference-style pickle."""
    import pickle

    import torch

    import torchlib.utils as tu
    from primia_amd.torchlib_compat import Arguments

    assert tu.Arguments is Arguments and Arguments.__module__ == "torchlib.utils"
    a = Arguments.__new__(Arguments)
    a.batch_size, a.lr, a.model = 8, 1e-3, "resnet-18"
    blob = pickle.dumps(a)
    assert b"torchlib.utils" in blob
    b = pickle.loads(blob)
    assert isinstance(b, Arguments) and b.batch_size == 8 and b.model == "resnet-18"
    p = tmp_path / "ckpt.pt"
    torch.save({"epoch": 3, "args": a, "val_mean_std": torch.zeros(2, 3)}, p)
    st = torch.load(p, weights_only=False)
    assert isinstance(st["args"], Arguments) and st["epoch"] == 3
    from torchlib.run_websocket_server import read_websocket_config  # noqa: F401
    from torchlib.utils import LearningRateScheduler, MixUp, To_one_hot, train_federated  # noqa: F401


def test_stats_table_matches_reference_rendering():
    """The validation table against the text rendered by the reference's own stats_table
    (tests/golden/make_metrics_golden.py), and test()'s objective = 100 * MCC."""
    import os

    import numpy as np
    from sklearn import metrics as mt

    from primia_amd.torchlib_compat import matthews_corrcoef, stats_table

    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
    for name, names in (("named", ["normal", "bacterial", "viral"]), ("numbered", None)):
        t, p = gold[f"{name}.target"], gold[f"{name}.pred"]
        cm = mt.confusion_matrix(t, p)
        rep = mt.classification_report(t, p, output_dict=True, zero_division=0)
        text = stats_table(cm, rep, roc_auc=0.8123, matthews_coeff=0.5678, class_names=names, epoch=7)
        assert text == bytes(gold[f"{name}.table"]).decode("utf-8")
        assert abs(matthews_corrcoef(t.tolist(), p.tolist(), 3) - mt.matthews_corrcoef(t, p)) < 1e-12


def test_distribute_data_tool_deals_like_the_reference_script(tmp_path):
    """tools/distribute_data.py: the IID deal is the reference's (random.seed(0) shuffle, i::num_workers — pinned in
    tests/golden/datapipe.npz against distribute_data.py itself), the label-skew deal assigns every image exactly once."""
    import importlib.util
    import os

    from oracle import datapipe_oracle as D

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("distribute_data", os.path.join(root, "tools", "distribute_data.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    src, tst = tmp_path / "train", tmp_path / "test"
    names = []
    for ci, c in enumerate(("a_normal", "b_bact", "c_viral")):
        for d in (src, tst):
            (d / c).mkdir(parents=True)
        for k in range(5 + 2 * ci):
            (src / c / f"{c}_{k}.png").write_bytes(b"x")
            names.append(f"{c}_{k}.png")
        (tst / c / f"t_{c}.png").write_bytes(b"y")
    out = tmp_path / "sim"
    shards = mod.main(["--train_data_src", str(src), "--test_data_src", str(tst), "--num_workers", "3", "--out", str(out), "-s"])
    assert shards == D.iid_round_robin_split(len(names), 3)
    got = sorted(f for w in range(3) for _, _, fs in os.walk(out / f"worker{w + 1}") for f in fs)
    assert got == sorted(names)
    assert sorted(os.listdir(out / "validation")) == ["a_normal", "b_bact", "c_viral"]
    out2 = tmp_path / "skew"
    shards2 = mod.main(["--train_data_src", str(src), "--test_data_src", "", "--num_workers", "4", "--out", str(out2),
                        "--label_skew", "0.3", "-s"])
    assert sorted(i for s in shards2 for i in s) == list(range(len(names)))
    assert len({len(s) for s in shards2}) > 1            # skewed shards are uneven
