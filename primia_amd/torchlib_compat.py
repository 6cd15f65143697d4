"""Host-side mirror of the pieces of PriMIA's `torchlib/utils.py` that sit on the hot path
(SURVEY.md §8b): same names, arguments and error behaviour, running on the HIP engine.

This file holds control flow only (schedules, sync bookkeeping, argument plumbing); every tensor
operation is a HIP kernel behind primia_amd.engine / primia_amd.fed.
"""
from typing import Optional

import numpy as np


class LearningRateScheduler:
    """Epoch -> learning rate on a log10 scale (torchlib/utils.py:37-89).

    One cycle covers T = total_epochs / (restarts + 1) epochs and the schedule is periodic in T, so `restarts`
    warm restarts simply replay the cycle.  Within a cycle, at position e:
        log_linear   log10 lr = log_start + (log_end - log_start) / T * e
        log_cosine   log10 lr = log_end + |log_start - log_end| * (1 + cos(pi * e / T)) / 2
    The expressions are evaluated in the reference's operation order so that the doubles agree to the last bit
    (tests/golden/lr_schedule.npz is minted by executing the reference's class)."""

    PLANS = ("log_linear", "log_cosine")

    def __init__(self, total_epochs: int, log_start_lr: float, log_end_lr: float,
                 schedule_plan: str = "log_linear", restarts: Optional[int] = None):
        if schedule_plan not in self.PLANS:
            raise NotImplementedError(f"learning rate schedule {schedule_plan!r} is not one of {self.PLANS}")
        self.schedule_plan = schedule_plan
        self.log_start_lr, self.log_end_lr = log_start_lr, log_end_lr
        self.total_epochs = total_epochs / (restarts + 1) if restarts else total_epochs

    def calc_lr(self, epoch):
        T, lo, hi = self.total_epochs, self.log_start_lr, self.log_end_lr
        if self.schedule_plan == "log_linear":
            exponent = ((hi - lo) / T) * epoch + lo
        else:
            exponent = (np.cos(np.pi * (epoch / T)) / 2.0 + 0.5) * abs(lo - hi) + hi
        return np.power(10, exponent)

    def get_lr(self, epoch):
        # periodic in T: the reference's range check behind the same modulo can never fire
        return self.calc_lr(epoch % self.total_epochs)

    def adjust_learning_rate(self, optimizer, epoch: int):
        """Set and return the epoch's rate; `optimizer` has torch's `param_groups` or a plain `lr` attribute."""
        lr = self.get_lr(epoch)
        groups = getattr(optimizer, "param_groups", None)
        if groups is None:
            optimizer.lr = lr
        else:
            for g in groups:
                g["lr"] = lr
        return lr


# ---------------------------------------------------------------------------------------------
# Arguments (torchlib/utils.py:92-267): the INI + CLI attribute bag, same names and fallbacks.
# ---------------------------------------------------------------------------------------------
_REQ = object()
# (section, key, attribute, kind, fallback)  — _REQ = no fallback, configparser raises if absent
_CONFIG_KEYS = [
    ("config", "batch_size", "batch_size", "int", _REQ),
    ("config", "test_batch_size", "test_batch_size", "int", _REQ),
    ("config", "train_resolution", "train_resolution", "int", _REQ),
    ("config", "validation_split", "validation_split", "int", _REQ),
    ("config", "epochs", "epochs", "int", _REQ),
    ("config", "lr", "lr", "float", _REQ),
    ("config", "deterministic", "deterministic", "bool", _REQ),
    ("config", "restarts", "restarts", "int", _REQ),
    ("config", "seed", "seed", "int", 1),
    ("config", "test_interval", "test_interval", "int", 1),
    ("config", "log_interval", "log_interval", "int", 10),
    ("config", "optimizer", "optimizer", "str", _REQ),
    ("config", "differentially_private", "differentially_private", "bool", False),
    ("config", "model", "model", "str", _REQ),
    ("config", "pooling_type", "pooling_type", "str", "max"),
    ("config", "pretrained", "pretrained", "bool", _REQ),
    ("config", "weight_decay", "weight_decay", "float", _REQ),
    ("config", "weight_classes", "weight_classes", "bool", _REQ),
    ("augmentation", "rotation", "rotation", "float", _REQ),
    ("augmentation", "translate", "translate", "float", _REQ),
    ("augmentation", "scale", "scale", "float", _REQ),
    ("augmentation", "shear", "shear", "float", _REQ),
    ("albumentations", "overall_prob", "albu_prob", "float", _REQ),
    ("albumentations", "individual_probs", "individual_albu_probs", "float", _REQ),
    ("albumentations", "noise_std", "noise_std", "float", _REQ),
    ("albumentations", "noise_prob", "noise_prob", "float", _REQ),
] + [("albumentations", k, k, "bool", _REQ) for k in (
    "clahe", "randomgamma", "randombrightness", "blur", "elastic", "optical_distortion", "grid_distortion",
    "grid_shuffle", "hsv", "invert", "cutout", "shadow", "fog", "sun_flare", "solarize", "equalize",
    "grid_dropout")] + [
    ("augmentation", "mixup", "mixup", "bool", _REQ),
    ("augmentation", "mixup_prob", "mixup_prob", "float", _REQ),
    ("augmentation", "mixup_lambda", "mixup_lambda", "float", None),
]
_FEDERATED_KEYS = [
    ("federated", "sync_every_n_batch", "sync_every_n_batch", "int", _REQ),
    ("federated", "wait_interval", "wait_interval", "float", 0.1),
    ("federated", "keep_optim_dict", "keep_optim_dict", "bool", _REQ),
    ("federated", "repetitions_dataset", "repetitions_dataset", "int", _REQ),
    ("federated", "weighted_averaging", "weighted_averaging", "bool", _REQ),
    ("federated", "precision_fractional", "precision_fractional", "float", 16),
]


def _cfg_get(config, section, key, kind, fallback):
    getter = {"int": config.getint, "float": config.getfloat, "bool": config.getboolean, "str": config.get}[kind]
    if fallback is _REQ:
        return getter(section, key)
    return getter(section, key, fallback=fallback)


class Arguments:
    # pickles as torchlib.utils.Arguments, the path the reference's checkpoints use (utils.py:1489); the
    # `torchlib` shim package at the repository root resolves it back to this class
    __module__ = "torchlib.utils"
    """Same attributes, fallbacks and side effects as the reference's Arguments."""

    def __init__(self, cmd_args, config, mode: str = "train", verbose: bool = True):
        assert mode in ["train", "inference"], "no other mode known"
        self.name = cmd_args.training_name if getattr(cmd_args, "training_name", None) else "default"
        self.save_file = getattr(cmd_args, "save_file", "model_weights/completed_trainings.csv")
        for section, key, attr, kind, fb in _CONFIG_KEYS:
            setattr(self, attr, _cfg_get(config, section, key, kind, fb))
        self.inference_resolution = config.getint("config", "inference_resolution", fallback=self.train_resolution)
        if self.train_resolution != self.inference_resolution:
            from warnings import warn

            warn("We are not supporting different train and inference resolutions although it works for some "
                 "scenarios.", category=UserWarning)
        self.end_lr = config.getfloat("config", "end_lr", fallback=self.lr)
        assert self.optimizer in ["SGD", "Adam"], "Unknown optimizer"
        if self.optimizer == "Adam":
            self.beta1 = config.getfloat("config", "beta1", fallback=0.9)
            self.beta2 = config.getfloat("config", "beta2", fallback=0.999)
        assert self.model in ["simpleconv", "resnet-18", "vgg16"]
        if self.mixup and self.mixup_prob == 1.0:
            self.batch_size *= 2
            print("Doubled batch size because of mixup")
        train = mode == "train"
        self.train_federated = cmd_args.train_federated if train else False
        self.unencrypted_aggregation = cmd_args.unencrypted_aggregation if train else False
        if self.train_federated:
            for section, key, attr, kind, fb in _FEDERATED_KEYS:
                setattr(self, attr, _cfg_get(config, section, key, kind, fb))
            if self.repetitions_dataset > 1:
                self.epochs = int(self.epochs / self.repetitions_dataset)
                if verbose:
                    print("Number of epochs was decreased to {:d} because of {:d} repetitions of dataset".format(
                        self.epochs, self.repetitions_dataset))
        self.visdom = cmd_args.visdom if train else False
        self.encrypted_inference = cmd_args.encrypted_inference if mode == "inference" else False
        self.data_dir = cmd_args.data_dir
        self.cuda = cmd_args.cuda
        self.websockets = cmd_args.websockets if train else False
        if self.websockets:
            assert self.train_federated, "If you use websockets it must be federated"
        self.num_threads = config.getint("system", "num_threads", fallback=0)

    @classmethod
    def from_namespace(cls, args):
        obj = cls.__new__(cls)
        for attr in dir(args):
            if not attr.startswith("__") and not callable(getattr(args, attr)):
                setattr(obj, attr, getattr(args, attr))
        return obj

    def from_previous_checkpoint(self, cmd_args):
        self.visdom = False
        if hasattr(cmd_args, "encrypted_inference"):
            self.encrypted_inference = cmd_args.encrypted_inference
        self.cuda = cmd_args.cuda
        self.websockets = getattr(cmd_args, "websockets", False)

    def incorporate_cmd_args(self, cmd_args):
        """Only attributes this object already has are overridden (torchlib/utils.py:282-292); command-line flags
        without a counterpart here are ignored."""
        for attr in [a for a in dir(self) if not a.startswith("__") and not callable(getattr(self, a))]:
            if hasattr(cmd_args, attr):
                setattr(self, attr, getattr(cmd_args, attr))

    def __str__(self):
        width = max(len(k) for k in vars(self))
        return "\n".join("{:>{w}}: {}".format(k, v, w=width) for k, v in sorted(vars(self).items()))


def read_websocket_config(path: str):
    """torchlib/run_websocket_server.py:6-8 — `read_csv(path, header=None, index_col=0).to_dict()`:
    the CSV has one ROW per field (id / host / port) and one COLUMN per worker, so the result is
    {column_number: {"id": ..., "host": ..., "port": ...}} with column numbers starting at 1."""
    import csv

    with open(path, newline="") as f:
        rows = [r for r in csv.reader(f) if r]
    out = {}
    for row in rows:
        field = row[0].strip()
        for col, value in enumerate(row[1:], start=1):
            v = value.strip()
            out.setdefault(col, {})[field] = int(v) if v.isdigit() else v
    return out


# ---------------------------------------------------------------------------------------------
# Federated epoch for clients that live in THIS process (one engine per client), the layout the
# reference uses with VirtualWorkers (torchlib/utils.py:936-1233).  The per-rank / RCCL form of the
# same control flow is primia_amd.fed.federated_epoch.
# ---------------------------------------------------------------------------------------------
def aggregation(local_model, models, workers, crypto_provider, args, test_params, weights=None, secure=True):
    """torchlib/utils.py:1000-1092 on flat arenas.  `models` maps worker id -> engine,
    `local_model` is the engine receiving the average.  Returns local_model."""
    import torch

    from . import _lib

    ids = [w if isinstance(w, str) else w.id for w in workers]
    shapes = {models[i].flat.numel() for i in ids}
    assert len(shapes) == 1 and local_model.flat.numel() == next(iter(shapes)), "Shape mismatch BEFORE sending and getting"
    n = local_model.flat.numel()
    out = local_model.flat
    if secure:
        pf = getattr(args, "precision_fractional", 16)
        scale = float(10 ** pf)
        acc = torch.zeros(n, dtype=torch.int64, device=out.device)
        q = torch.empty(n, dtype=torch.int64, device=out.device)
        tmp = torch.empty_like(out)
        for i in ids:
            tmp.copy_(models[i].flat)
            if weights:
                _lib.call("primia_scale", tmp, n, float(weights[i]))
            _lib.call("primia_fx_encode", tmp, q, n, scale)
            _lib.call("primia_ring_add", acc, q, acc, n, n)
        _lib.call("primia_fx_decode", acc, out, n, scale)
    else:
        tmp = torch.zeros_like(out)
        for i in ids:
            _lib.call("primia_axpy", tmp, models[i].flat, n, float(weights[i]) if weights else 1.0)
        out.copy_(tmp)
    if not weights:
        _lib.call("primia_divide", out, n, float(len(ids)))
    # `fresh_state_dict` carries no num_batches_tracked (utils.py:1040) and is a plain dict, so load_state_dict takes
    # BatchNorm's version-1 path, which under the reference's torch 1.4 (environment_torch.yml:98) re-creates the
    # counter as 0; send_new_models then hands that 0 to every worker model with the rest of the state dict.
    for b in local_model.num_batches_tracked:
        local_model.num_batches_tracked[b] = 0
    local_model.refresh_weights()
    return local_model


def send_new_models(local_model, models):
    """torchlib/utils.py:1095-1105: every listed worker model takes the averaged arena."""
    for worker, m in models.items():
        if worker == "local_model":
            continue
        m.flat.copy_(local_model.flat)
        m.num_batches_tracked.update(local_model.num_batches_tracked)
        m.refresh_weights()
    return models


def _as_optimizer(engine, opt, args):
    """Host loops take the optimizer object of primia_amd.optim; a bare {"lr": ...} (older callers, tests) is wrapped
    around the engine WITHOUT touching its state."""
    from .optim import EngineOptimizer

    if isinstance(opt, EngineOptimizer):
        return opt
    steps, state = engine.opt_steps, engine.opt_state
    o = EngineOptimizer.from_args(engine, args, lr=opt["lr"])
    engine.opt_steps, engine.opt_state = steps, state
    return o


def secure_aggregation_epoch(args, models, device, train_loaders, optimizers, epoch, loss_fns, crypto_provider,
                             weights=None, test_params=None, verbose=True, privacy_engines=None):
    """torchlib/utils.py:1108-1233.  train_loaders: {worker: iterable of (data, target)}; optimizers: {worker_id:
    EngineOptimizer}.  Unless `keep_optim_dict`, every worker's optimizer is RE-CREATED at the start of the epoch and
    after every mid-epoch sync with `lr = args.lr` (utils.py:1131-1145,1208-1218) — the reference thereby discards both
    the optimizer state and the rate its scheduler had just written into the old object (train.py:421-428); the new
    objects are stored back into `optimizers`, as the reference does."""
    import numpy as np

    from .optim import EngineOptimizer

    def wid(w):
        return w if isinstance(w, str) else w.id

    for w in list(optimizers):
        optimizers[w] = (EngineOptimizer.from_args(models[w], args) if not args.keep_optim_dict
                         else _as_optimizer(models[w], optimizers[w], args))
    avg_loss = []
    num_batches = {wid(w): len(tl) for w, tl in train_loaders.items()}
    loaders = {w: iter(tl) for w, tl in train_loaders.items()}
    secure = not args.unencrypted_aggregation
    for batch_idx in range(max(num_batches.values())):
        for w, it in loaders.items():
            i = wid(w)
            if batch_idx >= num_batches[i]:
                continue
            optimizers[i].zero_grad()
            data, target = next(it)
            sib = getattr(models[i], "sibling", None)    # (the ragged final batch of a client's loader)
            eng = models[i] if sib is None else sib(data.shape[0])
            eng.forward(data)
            loss = eng.loss_backward(target, soft=getattr(loss_fns.get(i), "soft", False) if loss_fns else False)
            optimizers[i].step() if eng is models[i] else optimizers[i].step(eng)
            avg_loss.append(loss.item())
        if batch_idx > 0 and batch_idx % args.sync_every_n_batch == 0:
            models["local_model"] = aggregation(models["local_model"], models, train_loaders.keys(), crypto_provider,
                                                args, test_params, weights=weights, secure=secure)
            send_new_models(models["local_model"],
                            {w: m for w, m in models.items() if w in num_batches and num_batches[w] > batch_idx})
            if not args.keep_optim_dict:
                for w in list(optimizers):
                    optimizers[w] = EngineOptimizer.from_args(models[w], args)
    models["local_model"] = aggregation(models["local_model"], models, train_loaders.keys(), crypto_provider, args,
                                        test_params, weights=weights, secure=secure)
    models = send_new_models(models["local_model"], models)
    return models, float(np.mean(avg_loss))


def train_federated(args, model, device, train_loaders, optimizer, epoch, loss_fn, crypto_provider,
                    test_params=None, vis_params=None, verbose=True, privacy_engines=None):
    """torchlib/utils.py:936-988."""
    total_batches = sum(len(tl) for tl in train_loaders.values())
    w_dict = None
    if args.weighted_averaging:
        w_dict = {(w if isinstance(w, str) else w.id): len(tl) / total_batches for w, tl in train_loaders.items()}
    model, avg_loss = secure_aggregation_epoch(args, model, device, train_loaders, optimizer, epoch, loss_fn,
                                               crypto_provider, test_params=test_params, weights=w_dict)
    if verbose:
        print("Train Epoch: {} \tLoss: {:.6f}".format(epoch, avg_loss))
    return model


def train(args, model, device, train_loader, optimizer, epoch, loss_fn, num_classes=3, vis_params=None,
          verbose=True):
    """torchlib/utils.py:1236-1292 — local (non-federated) epoch, incl. the on-device MixUp of :1249-1267."""
    losses = []
    optimizer = _as_optimizer(model, optimizer, args)
    if getattr(args, "mixup", False):
        from .datapipe import MixUp, To_one_hot

        mixup = MixUp(λ=args.mixup_lambda, p=args.mixup_prob)
        oh_converter = To_one_hot(num_classes)
    for batch_idx, (data, target) in enumerate(train_loader):
        soft = False
        if getattr(args, "mixup", False):
            target = oh_converter(target)
            data, target = mixup((data, target))
            soft = True
        # MixUp mixes the two halves of a batch with probability mixup_prob and passes it on whole otherwise: consecutive
        # steps see B or B / 2 samples (:1262-1267).  A sibling engine serves the other size on the same parameters.
        eng = model if data.shape[0] == getattr(model, "N", data.shape[0]) else model.sibling(data.shape[0])
        optimizer.zero_grad()
        eng.forward(data)
        loss = eng.loss_backward(target, soft=soft)
        optimizer.step() if eng is model else optimizer.step(eng)
        if batch_idx % args.log_interval == 0:
            losses.append(loss.item())
            if verbose:
                print("Train Epoch: {} [{}/{}]\tLoss: {:.6f}".format(epoch, batch_idx, len(train_loader), losses[-1]))
    return model


def matthews_corrcoef(y_true, y_pred, num_classes):
    """Multi-class MCC (what sklearn.metrics.matthews_corrcoef computes) from the confusion matrix."""
    import numpy as np

    c = np.zeros((num_classes, num_classes), dtype=np.float64)
    for t, p in zip(y_true, y_pred):
        c[int(t), int(p)] += 1
    t_k, p_k, n, tr = c.sum(1), c.sum(0), c.sum(), np.trace(c)
    num = tr * n - t_k @ p_k
    den = np.sqrt(n * n - p_k @ p_k) * np.sqrt(n * n - t_k @ t_k)
    return float(num / den) if den > 0 else 0.0


def stats_table(conf_matrix, report, roc_auc=0.0, matthews_coeff=0.0, class_names=None, epoch=0):
    """The validation table of torchlib/utils.py:1295-1351: one row per class (recall, precision, F1, support,
    confusion-matrix row), macro / weighted averages, then micro recall, MCC and ROC AUC; `fancy_grid`."""
    from tabulate import tabulate

    pct = lambda v: "{:.1f} %".format(v * 100.0)
    n = conf_matrix.shape[0]
    label = lambda i: class_names[i] if class_names else i
    rows = []
    for i in range(n):
        e = report[str(i)]
        rows.append([label(i), pct(e["recall"]), pct(e["precision"]), pct(e["f1-score"]), e["support"]]
                    + [conf_matrix[i, j] for j in range(conf_matrix.shape[1])])
    for title, key in (("Overall (macro)", "macro avg"), ("Overall (weighted)", "weighted avg")):
        e = report[key]
        rows.append([title, pct(e["recall"]), pct(e["precision"]), pct(e["f1-score"]), e["support"]])
    rows.append(["Overall stats", "micro recall", "matthews coeff", "AUC ROC score"])
    rows.append(["", pct(report["accuracy"]), "{:.3f}".format(matthews_coeff), "{:.3f}".format(roc_auc)])
    headers = ["Epoch {:d}".format(epoch), "Recall", "Precision", "F1 score", "n total"] + [label(i) for i in range(n)]
    return tabulate(rows, headers=headers, tablefmt="fancy_grid")


def test(args, model, device, val_loader, epoch, loss_fn, num_classes, verbose=True, vis_params=None,
         class_names=None):
    """torchlib/utils.py:1354-1467 (plaintext branch): mean batch loss, ROC AUC (one-vs-one on the min-shifted,
    row-normalised logits), MCC; returns (loss, 100 * MCC) — the objective train.py maximises — and prints the
    reference's table when verbose."""
    from warnings import warn

    import numpy as np
    import torch

    model.eval()
    nll, preds, tgts, scores = [], [], [], []
    for data, target in val_loader:
        # (a ragged final batch — every validation sample counts — runs on a sibling engine of that size; eval-mode
        # BatchNorm makes the logits independent of how the samples are batched)
        logits = (model.sibling(data.shape[0]) if hasattr(model, "sibling") else model).forward(data)
        ls = torch.log_softmax(logits, dim=1)
        nll.append(-ls.gather(1, target.view(-1, 1)).reshape(-1).double().cpu())
        scores.append(logits.detach().float().cpu().numpy().copy())
        preds += logits.argmax(1).tolist()
        tgts += target.tolist()
    model.train()
    # the reference averages the loss per batch of test_batch_size, then over batches (:1392-1412)
    nll = torch.cat(nll)
    tbs = max(1, int(getattr(args, "test_batch_size", 1) or 1))
    test_loss = float(np.mean([float(nll[i:i + tbs].mean()) for i in range(0, nll.numel(), tbs)]))
    total_scores = np.concatenate(scores)
    total_scores -= total_scores.min(axis=1)[:, np.newaxis]
    total_scores = total_scores / total_scores.sum(axis=1)[:, np.newaxis]
    try:
        from sklearn import metrics as mt

        roc_auc = mt.roc_auc_score(np.asarray(tgts), total_scores, multi_class="ovo")
    except ValueError:
        warn("ROC AUC score could not be calculated and was set to zero.", category=UserWarning)
        roc_auc = 0.0
    mcc = matthews_corrcoef(tgts, preds, num_classes)
    if verbose:
        from sklearn import metrics as mt

        cm = mt.confusion_matrix(tgts, preds, labels=list(range(num_classes)))
        rep = mt.classification_report(tgts, preds, labels=list(range(num_classes)), output_dict=True, zero_division=0)
        if "accuracy" not in rep:   # sklearn omits it when `labels` is a superset of the observed classes
            rep["accuracy"] = float(np.mean(np.asarray(preds) == np.asarray(tgts)))
        print(stats_table(cm, rep, roc_auc=roc_auc, matthews_coeff=mcc, class_names=class_names, epoch=epoch))
    return test_loss, 100.0 * mcc


def save_model(model, optim, path, args, epoch, val_mean_std):
    """torchlib/utils.py:1470-1493 — same checkpoint keys, so reference tooling reads it."""
    import os

    import torch

    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    def sd(o):
        return o.state_dict() if hasattr(o, "state_dict") else dict(o)

    opt_state_dict = {name: sd(o) for name, o in optim.items()} if args.train_federated else sd(optim)
    model = model["local_model"] if isinstance(model, dict) else model
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optim_state_dict": opt_state_dict,
                "args": args, "val_mean_std": val_mean_std}, path)


class Cross_entropy_one_hot:
    """torchlib/utils.py:404-437 — cross entropy on soft (one-hot / mixed) targets, optionally class-weighted:
    mean_n( sum_c(weight_c t_nc) * sum_c(-t_nc log_softmax(o)_nc) ).  In the training loops the engine computes exactly this
    inside loss_backward (primia_xent_soft); the object carries `soft = True` (what the loops read) and `weight`, and is
    callable on (logits, targets) for host-side uses such as the validation loss."""
    soft = True

    def __init__(self, reduction="mean", weight=None):
        if reduction not in ("mean", "sum"):
            raise NotImplementedError("reduction must be mean or sum")
        self.reduction, self.weight = reduction, weight

    def __call__(self, output, target):
        import torch

        per = torch.sum(-target * torch.log_softmax(output.float(), dim=1), dim=1)
        if self.weight is not None:
            per = torch.sum(self.weight.to(target) * target, dim=1) * per
        return per.mean() if self.reduction == "mean" else per.sum()

    forward = __call__


def save_config_results(args, score, timestamp, table):
    """torchlib/utils.py:859-874 — one row per run (every non-callable attribute of `args`, the timestamp, the best
    validation score) appended to the CSV `table`."""
    from os.path import isfile

    import pandas as pd

    members = [a for a in dir(args) if not callable(getattr(args, a)) and not a.startswith("__")]
    if not isfile(table):
        print("Configuration table does not exist - Creating new")
        df = pd.DataFrame(columns=members)
    else:
        df = pd.read_csv(table)
    row = dict(zip(members, [getattr(args, x) for x in members]))
    row["timestamp"] = timestamp
    row["best_validation_score"] = score
    df = pd.concat([df, pd.DataFrame([row])], ignore_index=True)      # (DataFrame.append of pandas < 2)
    df.to_csv(table, index=False)


def resnet18(pretrained=False, progress=True, in_channels=3, pooling="avg", num_classes=1000, input_size=224,
             batch_size=None, adptpool=False, dtype=None, device="cuda:0", norm_layer=None, **kwargs):
    """torchlib/models.py:499-516 with the keyword arguments train.py passes (:259-268): the same network as a
    ResNet18Engine.  One thing the reference's module does not need must be given: `batch_size` (the engine's buffers are
    sized for it; other sizes run on `engine.sibling(n)`).  `pretrained=True` loads the ImageNet weights the way train.py
    does (a local file; refused when absent).  `adptpool` (an adaptive average pool in front of fc) is the identity at the
    resolutions the engine serves (multiples of 32 whose last feature map the 7 x 7 / final pooling covers)."""
    import torch

    from .engine import ResNet18Engine

    if batch_size is None:
        raise TypeError("resnet18(): batch_size=... is required (the HIP engine allocates its activations up front)")
    if kwargs:
        raise TypeError("resnet18(): unsupported arguments {}".format(sorted(kwargs)))
    norm = "batch"
    if norm_layer is not None:      # train.py:308 passes a GroupNorm factory for differentially_private = yes
        norm = "group"
    eng = ResNet18Engine(batch_size, num_classes, in_channels, input_size, pooling,
                         dtype=dtype or torch.bfloat16, device=device, norm=norm)
    eng.init_weights()
    if pretrained:
        import train as _train        # the loader of the CLI (reads PRIMIA_PRETRAINED_RESNET18 / torch's hub cache)

        _train.load_pretrained(eng, num_classes)
    return eng


def vgg16(*args, **kwargs):
    raise NotImplementedError("vgg16 is outside the path this library accelerates (ResNet-18: BASELINE.json north_star)")


def conv_at_resolution(*args, **kwargs):
    raise NotImplementedError("the small ConvNets are outside the path this library accelerates (ResNet-18)")
