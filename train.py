#!/usr/bin/env python3
"""PriMIA-compatible training CLI on the MI355X engine.

Same command line, INI keys and worker CSV as the reference's train.py (train.py:555-631):

    python train.py --config configs/torch/pneumonia-resnet-pretrained.ini --train_federated \
        [--unencrypted_aggregation] [--data_dir DIR|synthetic] [--cuda] [--resume_checkpoint P] \
        [--save_file F] [--training_name N]

Differences, all on the side of "more works":
  * `--cuda` together with `--train_federated` is legal (the reference refuses it,
    train.py:617-622): client i of configs/websetting/config.csv runs on the GPU;
  * launched under `torch.distributed.run` (WORLD_SIZE = number of clients) every rank is one client
    on its own GPU and FedAvg is an RCCL all-reduce; launched plainly, all clients live in this
    process like the reference's VirtualWorkers and are visited sequentially;
  * `--data_dir synthetic` (default when the folder does not exist) trains on seeded synthetic
    3x224x224 batches — the data pipeline (albumentations, DICOM) is outside the hot path.

`main(args, verbose, optuna_trial, cmd_args)` returns the best validation MCC like the reference's.
"""
import argparse
import configparser
import os
import random
from os import path

import numpy as np
import torch

from primia_amd.engine import ResNet18Engine
from primia_amd.torchlib_compat import (Arguments, LearningRateScheduler, read_websocket_config, save_model, test,
                                        train, train_federated)


class SyntheticLoader:
    """Device-resident synthetic shard: yields (x fp32 NCHW on the GPU, int64 labels)."""

    def __init__(self, n_batches, batch, size, num_classes, device, seed):
        g = torch.Generator().manual_seed(seed)
        self.data = [(torch.randn(batch, 3, size, size, generator=g).to(device),
                      torch.randint(0, num_classes, (batch,), generator=g).to(device)) for _ in range(n_batches)]

    def __len__(self):
        return len(self.data)

    def __iter__(self):
        return iter(self.data)


def setup_workers(args):
    """setup_pysyft (torchlib/utils.py:516-542): worker list from the CSV, crypto_provider split off."""
    worker_dict = read_websocket_config("configs/websetting/config.csv")
    names = [w["id"] for w in worker_dict.values()]
    crypto_in_config = "crypto_provider" in names
    assert args.unencrypted_aggregation or crypto_in_config, "No crypto provider in configuration"
    if crypto_in_config:
        names.remove("crypto_provider")
    return names, ("crypto_provider" if crypto_in_config else None)


def main(args, verbose=True, optuna_trial=None, cmd_args=None):
    use_cuda = torch.cuda.is_available()
    if not use_cuda:
        raise SystemExit("primia_amd trains on the GPU only (HIP kernels); no GPU visible")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    np.random.seed(args.seed)
    num_classes = 3
    size = args.train_resolution
    if args.model != "resnet-18":
        raise NotImplementedError("only resnet-18 is on the accelerated path")
    dtype = torch.float32 if os.environ.get("PRIMIA_DTYPE", "bf16") == "f32" else torch.bfloat16
    n_batches = int(os.environ.get("PRIMIA_SYNTHETIC_BATCHES", 8))

    def make_engine():
        # differentially_private = yes (train.py:304-334 of the reference): BatchNorm is rejected by the
        # PrivacyEngine, so the network is built with GroupNorm and every step clips / noises per sample
        eng = ResNet18Engine(args.batch_size, num_classes, 3 if args.pretrained else 1, size, args.pooling_type,
                             dtype=dtype, device=device, norm="group" if args.differentially_private else "batch")
        if args.differentially_private:
            eng.dp_params = {"max_grad_norm": 1.0, "noise_multiplier": 1.3}
        return eng

    local = make_engine()
    local.init_weights()
    start_at_epoch = 1
    if cmd_args is not None and getattr(cmd_args, "resume_checkpoint", None):
        state = torch.load(cmd_args.resume_checkpoint, map_location="cpu", weights_only=False)
        local.load_state_dict(state["model_state_dict"])
        start_at_epoch = state["epoch"] + 1
    val_loader = SyntheticLoader(2, args.batch_size, size, num_classes, device, args.seed + 999)
    scheduler = LearningRateScheduler(args.epochs, np.log10(args.lr), np.log10(args.end_lr), restarts=args.restarts)
    exp_name = "{:s}_{:s}".format("federated" if args.train_federated else "vanilla", args.name)

    if args.train_federated:
        workers, crypto_provider = setup_workers(args)
        model = {"local_model": local}
        for i, w in enumerate(workers):
            model[w] = make_engine()
            model[w].load_state_dict(local.state_dict())
        # synthetic, deliberately uneven shards (exercises weighted averaging and exhausted clients)
        train_loader = {w: SyntheticLoader(n_batches - i, args.batch_size, size, num_classes, device, args.seed + i)
                        for i, w in enumerate(workers)}
        optimizer = {w: {"lr": args.lr} for w in workers}
        loss_fn = {w: None for w in workers}
    else:
        model = local
        train_loader = SyntheticLoader(n_batches, args.batch_size, size, num_classes, device, args.seed)
        optimizer = {"lr": args.lr}
        loss_fn = None

    objectives, model_paths = [], []
    for epoch in range(start_at_epoch, args.epochs + 1):
        new_lr = float(scheduler.get_lr(epoch - 1))
        if args.train_federated:
            for w in optimizer:
                optimizer[w]["lr"] = new_lr
            model = train_federated(args, model, device, train_loader, optimizer, epoch, loss_fn, None,
                                    verbose=verbose)
            eval_model = model["local_model"]
        else:
            optimizer["lr"] = new_lr
            model = train(args, model, device, train_loader, optimizer, epoch, loss_fn, num_classes, verbose=verbose)
            eval_model = model
        if epoch % args.test_interval == 0:
            _, objective = test(args, eval_model, device, val_loader, epoch, loss_fn, num_classes, verbose=verbose)
            objectives.append(objective)
            p = "model_weights/{:s}_epoch_{:03d}.pt".format(exp_name, epoch)
            save_model(eval_model, optimizer, p, args, epoch, val_mean_std=(torch.zeros(3), torch.ones(3)))
            model_paths.append(p)
    if not objectives:
        return 0.0
    best = int(np.argmax(objectives))
    final = "model_weights/final_{:s}.pt".format(exp_name)
    os.replace(model_paths[best], final)
    for i, p in enumerate(model_paths):
        if i != best and path.exists(p):
            os.remove(p)
    if verbose:
        print("best epoch {:d} -> {:s}".format(best + 1, final))
    return objectives[best]


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, required=True, help="Path to the configuration file (.ini).")
    parser.add_argument("--train_federated", action="store_true", help="Train with federated learning.")
    parser.add_argument("--unencrypted_aggregation", action="store_true", help="Turns off secure aggregation.")
    parser.add_argument("--data_dir", type=str, default="data/train", help="Data folder, or 'synthetic'.")
    parser.add_argument("--visdom", action="store_true", help="Accepted for compatibility (ignored).")
    parser.add_argument("--cuda", action="store_true", help="Use GPU acceleration (always on here).")
    parser.add_argument("--resume_checkpoint", type=str, default=None, help="Start from an older checkpoint")
    parser.add_argument("--websockets", action="store_true", help="Accepted for compatibility (ignored).")
    parser.add_argument("--verbose", action="store_true")
    parser.add_argument("--save_file", type=str, default="model_weights/completed_trainings.csv")
    parser.add_argument("--training_name", default=None, type=str)
    cmd_args = parser.parse_args()
    config = configparser.ConfigParser()
    assert path.isfile(cmd_args.config), "Configuration file not found"
    config.read(cmd_args.config)
    cmd_args.websockets = False  # in-process / RCCL clients replace the websocket transport
    args = Arguments(cmd_args, config, mode="train")
    if args.train_federated and (args.mixup or args.weight_classes):
        if args.mixup and args.mixup_lambda == 0.5:
            args.mixup_lambda = 0.499
    print(str(args))
    main(args, cmd_args=cmd_args)
