#!/usr/bin/env python3
"""PriMIA-compatible training CLI on the MI355X engine.

Same command line, INI keys and worker CSV as the reference's train.py (train.py:555-631):

    python train.py --config configs/torch/pneumonia-resnet-pretrained.ini --train_federated \
        [--unencrypted_aggregation] [--data_dir DIR|synthetic] [--cuda] [--resume_checkpoint P] \
        [--save_file F] [--training_name N]

Two deployments of the same federated epoch (torchlib/utils.py:936-1233):

  * plain launch: every client of configs/websetting/config.csv lives in this process on GPU 0 and is visited
    in turn, like the reference's VirtualWorkers (`torchlib_compat.train_federated`);
  * `python -m torch.distributed.run --nproc-per-node K train.py ... --train_federated`: rank k IS client k of the CSV
    (crypto_provider row excluded, utils.py:531-542) on GPU k; the process group is created before anything touches
    the GPU's allocator, FedAvg is an RCCL all-reduce of the flat arena (`fed.federated_epoch`), secure aggregation
    hides every client's encoded update under pairwise one-time masks, rank 0 validates and writes the checkpoints.
    K must equal the number of clients in the CSV (PRIMIA_WEBSOCKETS_CONFIG selects another CSV, e.g.
    configs/websetting/config_8gpu.csv).

`--cuda` together with `--train_federated` is legal here (the reference refuses it, train.py:617-622).
`--data_dir synthetic` trains on seeded synthetic batches; a folder is read as the reference lays it out
(`<data_dir>/worker<i>/<class>/<image>` per client, `<data_dir>/validation/<class>/<image>`; vanilla training reads
`<data_dir>/<class>/<image>`), resized / cropped / normalised on the GPU (primia_amd.imagefolder).

`main(args, verbose, optuna_trial, cmd_args)` returns the best validation MCC like the reference's.
"""
import argparse
import configparser
import os
import random
import shutil
from os import path
from warnings import warn

from types import SimpleNamespace

import numpy as np
import torch

from primia_amd.engine import ResNet18Engine
from primia_amd.optim import EngineOptimizer
from primia_amd.torchlib_compat import (Arguments, LearningRateScheduler, read_websocket_config, save_model, test,
                                        train, train_federated)

WEBSOCKETS_CONFIG = os.environ.get("PRIMIA_WEBSOCKETS_CONFIG", "configs/websetting/config.csv")


class SyntheticLoader:
    """Device-resident synthetic shard: yields (x fp32 NCHW on the GPU, int64 labels) — or, with `args` of a federated
    run that has mixup / weight_classes on, the registered form of the same samples (one-hot rows, each sample blended
    with its predecessor: torchlib/utils.py:680-734 through primia_amd.imagefolder.register)."""

    def __init__(self, n_batches, batch, size, num_classes, device, seed, channels=3, args=None):
        g = torch.Generator().manual_seed(seed)
        self.data = [(torch.randn(batch, channels, size, size, generator=g).to(device),
                      torch.randint(0, num_classes, (batch,), generator=g).to(device)) for _ in range(n_batches)]
        if args is not None and getattr(args, "train_federated", False) and (args.mixup or args.weight_classes):
            from types import SimpleNamespace

            from primia_amd.imagefolder import register

            once = SimpleNamespace(**{**vars(args), "repetitions_dataset": 1})   # synthetic shards are not repeated
            x, y = register([torch.cat([d for d, _ in self.data])], torch.cat([t for _, t in self.data]), once,
                            num_classes, seed)
            self.data = [(x[i * batch:(i + 1) * batch], y[i * batch:(i + 1) * batch]) for i in range(n_batches)]

    def __len__(self):
        return len(self.data)

    def __iter__(self):
        return iter(self.data)

    @property
    def targets(self):
        return torch.cat([t for _, t in self.data])


PRETRAINED_FILES = ("resnet18-5c106cde.pth", "resnet18-f37072fd.pth")      # torchvision's ImageNet ResNet-18 files


def load_pretrained(engine, num_classes, rank=0, world=1):
    """`pretrained = yes` (torchlib/models.py:488-496): the ImageNet state dict goes into the freshly constructed
    network, then `fc` is REPLACED by a new Linear(512, num_classes) with torch's default initialisation.  The
    reference downloads the file; here it is read from PRIMIA_PRETRAINED_RESNET18 or torch's hub cache
    (~/.cache/torch/hub/checkpoints/) — there is no network on the training box.  A missing file ENDS the run (the
    reference's main preset depends on these weights; training from scratch instead would be a different experiment)
    unless PRIMIA_ALLOW_RANDOM_INIT=1 says that is what is wanted.  Under torch.distributed.run rank 0 reads the file
    and every rank takes rank 0's arena, so all clients start from the same model whatever their own caches hold."""
    if engine.norm != "batch":
        raise SystemExit("pretrained ImageNet weights carry BatchNorm statistics; differentially_private = yes builds "
                         "a GroupNorm network (train.py:304-334)")
    # rank 0's outcome travels to every rank BEFORE anybody raises: 1 loaded, 0 no file, 2 the file does not fit / does
    # not load (a SystemExit on rank 0 alone left the others blocked in the broadcast until the RCCL timeout)
    status, problem = 1, ""
    if rank == 0:
        cands = [os.environ.get("PRIMIA_PRETRAINED_RESNET18")] + [
            path.join(path.expanduser("~"), ".cache", "torch", "hub", "checkpoints", f) for f in PRETRAINED_FILES]
        src = next((c for c in cands if c and path.isfile(c)), None)
        if src is None:
            status = 0
        else:
            try:
                sd = torch.load(src, map_location="cpu", weights_only=True)
                own = engine.state_dict()
                for k, v in sd.items():
                    if k.startswith("fc."):
                        continue
                    if k not in own or tuple(own[k].shape) != tuple(v.shape):
                        raise ValueError("{:s} does not fit this network ({} vs {})".format(
                            k, tuple(v.shape), tuple(own[k].shape) if k in own else None))
                    own[k] = v.to(torch.float32)
                fc = torch.nn.Linear(512, num_classes)        # model.fc = nn.Linear(512 * block.expansion, num_classes)
                own["fc.weight"], own["fc.bias"] = fc.weight.detach().clone(), fc.bias.detach().clone()
                engine.load_state_dict(own)
            except Exception as e:  # noqa: BLE001 - reported on every rank below
                status, problem = 2, "{:s}: {:s}".format(src, str(e))
    if world > 1:
        import torch.distributed as dist

        flag = torch.tensor([status], device=engine.flat.device)
        dist.broadcast(flag, 0)
        status = int(flag.item())
    if status == 2:
        raise SystemExit("pretrained = yes, but the ImageNet state dict could not be used" + (": " + problem if problem else
                         " (see rank 0's message)"))
    found = status == 1
    if not found:
        msg = ("pretrained = yes, but no ImageNet ResNet-18 state dict was found (set PRIMIA_PRETRAINED_RESNET18 to a "
               "torchvision resnet18 .pth)")
        if os.environ.get("PRIMIA_ALLOW_RANDOM_INIT") != "1":
            raise SystemExit(msg + "; PRIMIA_ALLOW_RANDOM_INIT=1 trains from the random initialisation instead")
        warn(msg + ": training starts from the random initialisation (PRIMIA_ALLOW_RANDOM_INIT=1)")
    if world > 1:
        dist.broadcast(engine.flat, 0)          # found or not: every client starts from rank 0's model
        engine.refresh_weights()
    return found


def setup_workers(args):
    """setup_pysyft (torchlib/utils.py:516-542): worker list from the CSV, crypto_provider split off."""
    worker_dict = read_websocket_config(WEBSOCKETS_CONFIG)
    names = [w["id"] for w in worker_dict.values()]
    crypto_in_config = "crypto_provider" in names
    assert args.unencrypted_aggregation or crypto_in_config, "No crypto provider in configuration"
    if crypto_in_config:
        names.remove("crypto_provider")
    return names, ("crypto_provider" if crypto_in_config else None)


def init_distributed():
    """(rank, world, device).  Under torch.distributed.run the process group is created FIRST, bound to this rank's
    GPU (backend "nccl" = RCCL; PRIMIA_BACKEND=gloo lets several ranks share one GPU in the tests)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("primia_amd trains on the GPU only (HIP kernels); no GPU visible")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world == 1:
        return 0, 1, device
    import torch.distributed as dist

    backend = os.environ.get("PRIMIA_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return dist.get_rank(), world, device


def resume(cmd_args, args, model, optimizer, worker_names):
    """The four checkpoint / configuration combinations of train.py:344-389.  Returns the epoch to start at —
    `state["epoch"]` itself, as the reference does (the saved epoch is run again)."""
    state = torch.load(cmd_args.resume_checkpoint, map_location="cpu", weights_only=False)
    ckpt_args = state["args"]
    was_federated = bool(getattr(ckpt_args, "train_federated", False))
    opt_sd = state["optim_state_dict"]
    if args.train_federated and was_federated:
        for w in worker_names:
            if w not in opt_sd:
                warn("The worker names of the checkpoint and the current configuration cannot be matched.")
                raise SystemExit(1)
            if w in optimizer:          # per-rank launch: each rank restores its own client's optimizer
                optimizer[w].load_state_dict(opt_sd[w])
        for m in model.values():
            m.load_state_dict(state["model_state_dict"])
    elif args.train_federated and not was_federated:
        assert len(opt_sd) == 2 and "param_groups" in opt_sd and "state" in opt_sd  # no federated checkpoint
        for w in worker_names:
            if w in optimizer:
                optimizer[w].load_state_dict(opt_sd)
        for m in model.values():
            m.load_state_dict(state["model_state_dict"])
    elif not args.train_federated and was_federated:
        # no optimizer is loaded.  (The reference indexes model_state_dict["local_model"] here although save_model
        # stores the local model's state dict itself, utils.py:1484-1486 — both layouts are accepted.)
        sd = state["model_state_dict"]
        model.load_state_dict(sd["local_model"] if "local_model" in sd else sd)
    else:
        optimizer.load_state_dict(opt_sd)
        model.load_state_dict(state["model_state_dict"])
    return state["epoch"]


def main(args, verbose=True, optuna_trial=None, cmd_args=None):
    rank, world, device = init_distributed()
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    np.random.seed(args.seed)
    num_classes = 3
    size = args.train_resolution
    channels = 3 if args.pretrained else 1
    if args.model != "resnet-18":
        raise NotImplementedError("only resnet-18 is on the accelerated path")
    if world > 1 and not args.train_federated:
        raise SystemExit("several ranks are several federated clients: launch with --train_federated")
    if args.differentially_private and (args.weight_classes or args.mixup):
        # the per-sample (DP-SGD) loss gradient takes hard, unweighted labels: say so before any data is registered
        raise SystemExit("differentially_private = yes cannot be combined with weight_classes / mixup: the per-sample "
                         "clipped gradients are formed from hard, unweighted targets")
    dtype = torch.float32 if os.environ.get("PRIMIA_DTYPE", "bf16") == "f32" else torch.bfloat16
    n_batches = int(os.environ.get("PRIMIA_SYNTHETIC_BATCHES", 8))
    synthetic = args.data_dir in (None, "synthetic")
    if not synthetic:
        if not path.isdir(str(args.data_dir)):
            # (round 2 silently trained on synthetic batches here)
            raise SystemExit("data_dir {!r} does not exist; pass --data_dir synthetic for seeded synthetic batches".format(
                args.data_dir))
        from primia_amd import imagefolder

    def make_engine():
        # differentially_private = yes (train.py:304-334 of the reference): BatchNorm is rejected by the
        # PrivacyEngine, so the network is built with GroupNorm and every step clips / noises per sample
        eng = ResNet18Engine(args.batch_size, num_classes, channels, size, args.pooling_type,
                             dtype=dtype, device=device, norm="group" if args.differentially_private else "batch")
        if args.differentially_private:
            eng.dp_params = {"max_grad_norm": 1.0, "noise_multiplier": 1.3}
        return eng

    local = make_engine()
    local.init_weights()
    if args.pretrained:
        load_pretrained(local, num_classes, rank, world)
    val_mean_std = (torch.zeros(channels), torch.ones(channels))
    exp_name = "{:s}_{:s}".format("federated" if args.train_federated else "vanilla", args.name)
    group = None
    masks = None

    if args.train_federated:
        workers, crypto_provider = setup_workers(args)
        if world > 1:
            import torch.distributed as dist

            from primia_amd import fed

            if len(workers) != world:
                raise SystemExit("{:d} ranks but {:d} clients in {:s}: one rank per client".format(
                    world, len(workers), WEBSOCKETS_CONFIG))
            mine = [workers[rank]]
        else:
            mine = list(workers)
        model = {"local_model": local}
        for w in mine:
            model[w] = make_engine()
            model[w].load_state_dict(local.state_dict())
        train_loader, stats = {}, {}
        for w in mine:
            i = workers.index(w)
            if synthetic:
                # deliberately uneven shards (exercises weighted averaging and exhausted clients)
                train_loader[w] = SyntheticLoader(max(1, n_batches - i), args.batch_size, size, num_classes, device,
                                                  args.seed + i, channels, args)
                stats[w] = (torch.zeros(channels, device=device), torch.ones(channels, device=device))
            else:
                train_loader[w], stats[w] = imagefolder.client_loader(
                    path.join(args.data_dir, "worker{:d}".format(i + 1)), args, device, channels, args.seed + i)
        # setup_pysyft's secure average of the clients' (mean, std) (utils.py:764-794) -> val_mean_std
        if world > 1 and not args.unencrypted_aggregation:
            masks = fed.PairwiseMasks.setup(local.flat.numel(), device, group)
        if world > 1:
            m, s = fed.exchange_mean_std(*stats[mine[0]], masks=masks)
        else:
            from primia_amd import fed

            m, s = fed.secure_mean_of([stats[w] for w in mine])
        val_mean_std = (m.cpu(), s.cpu())
        optimizer = {w: EngineOptimizer.from_args(model[w], args) for w in mine}
        # train.py:335-342: Cross_entropy_one_hot (soft targets) with mixup or federated weight_classes
        soft = bool(args.mixup or args.weight_classes)
        loss_fn = {w: SimpleNamespace(soft=soft) for w in mine}
    else:
        workers, mine = [], []
        model = local
        if synthetic:
            train_loader = SyntheticLoader(n_batches, args.batch_size, size, num_classes, device, args.seed, channels)
        else:
            train_loader, (m, s) = imagefolder.client_loader(args.data_dir, args, device, channels, args.seed)
            val_mean_std = (m.cpu(), s.cpu())
        optimizer = EngineOptimizer.from_args(model, args)
        loss_fn = None
    if synthetic or not path.isdir(path.join(str(args.data_dir), "validation")):
        val_loader = SyntheticLoader(2, args.batch_size, size, num_classes, device, args.seed + 999, channels)
    else:
        val_loader = imagefolder.validation_loader(path.join(args.data_dir, "validation"), args, device, channels,
                                                   val_mean_std)

    # class weights (train.py:192-195, utils.py:469-513): counted over the training targets of every client — across
    # ranks the per-class counts are summed — and handed to every engine's loss
    if args.weight_classes:
        from primia_amd.datapipe import class_counts, class_weights_from_counts

        # counted from the loaders' held targets (no batch is drawn: the augmentation and shuffle streams stay where
        # they are); across ranks the per-class COUNTS are summed (the weights are not additive), under the pairwise
        # masks when aggregation is secure
        occ = class_counts(train_loader, num_classes)
        if world > 1:
            import torch.distributed as dist

            if masks is not None:
                occ = fed.masked_sum(occ.to(device), masks).cpu()
            else:
                occ = occ.to(device)
                dist.all_reduce(occ)
                occ = occ.cpu()
        cw = class_weights_from_counts(occ)
        engines = [m for m in model.values()] if isinstance(model, dict) else [model]
        for m in engines:
            m.class_weight = cw.to(device).float().contiguous()

    start_at_epoch = 1
    if cmd_args is not None and getattr(cmd_args, "resume_checkpoint", None):
        print("Resume training from a given checkpoint.")
        start_at_epoch = resume(cmd_args, args, model, optimizer, workers)
    scheduler = LearningRateScheduler(args.epochs, np.log10(args.lr), np.log10(args.end_lr), restarts=args.restarts)
    reps = args.repetitions_dataset if "repetitions_dataset" in vars(args) else 1

    objectives, model_paths = [], []
    local_flat = None
    for epoch in range(start_at_epoch, args.epochs + 1):
        for o in (optimizer.values() if args.train_federated else [optimizer]):
            scheduler.adjust_learning_rate(o, epoch - 1)
        if args.train_federated and world > 1:
            w = mine[0]
            avg_loss, _, local_flat, optimizer[w] = fed.federated_epoch(
                model[w], train_loader[w], args, optimizer[w], group=group, local_flat=local_flat, masks=masks,
                soft_targets=bool(args.mixup or args.weight_classes))
            local.flat.copy_(local_flat)              # "local_model": the last global average
            for b in local.num_batches_tracked:
                local.num_batches_tracked[b] = 0
            local.refresh_weights()
            if verbose and rank == 0:
                print("Train Epoch: {} \tLoss: {:.6f}".format(epoch, avg_loss))
            eval_model = local
        elif args.train_federated:
            model = train_federated(args, model, device, train_loader, optimizer, epoch, loss_fn, crypto_provider,
                                    verbose=verbose)
            eval_model = model["local_model"]
        else:
            model = train(args, model, device, train_loader, optimizer, epoch, loss_fn, num_classes, verbose=verbose)
            eval_model = model
        if epoch % args.test_interval == 0:
            opt_for_ckpt = optimizer
            if world > 1:
                # the checkpoint holds every worker's optimizer state (utils.py:1471-1472): collect them on rank 0
                import torch.distributed as dist

                gathered = [None] * world
                dist.all_gather_object(gathered, (mine[0], optimizer[mine[0]].state_dict()))
                opt_for_ckpt = dict(gathered)
            # rank 0 validates; what it decides (pruned / failed) reaches every rank BEFORE anybody moves on, so a
            # pruned trial or a validation error ends all ranks together instead of leaving them in a barrier until
            # the RCCL timeout.  The pruning check precedes the checkpoint, as train.py:505-517 of the reference.
            verdict = [None]        # None = go on, "pruned", or the exception rank 0 hit
            if rank == 0:
                try:
                    _, objective = test(args, eval_model, device, val_loader, epoch, loss_fn, num_classes,
                                        verbose=verbose)
                    objectives.append(objective)
                    if optuna_trial:
                        optuna_trial.report(objective, epoch * reps)
                        if optuna_trial.should_prune():
                            verdict[0] = "pruned"
                    if verdict[0] is None:
                        p = "model_weights/{:s}_epoch_{:03d}.pt".format(exp_name, epoch * reps)
                        save_model(eval_model, opt_for_ckpt, p, args, epoch, val_mean_std=val_mean_std)
                        model_paths.append(p)
                except Exception as e:  # noqa: BLE001 - forwarded to every rank, re-raised below
                    verdict[0] = e
            if world > 1:
                dist.broadcast_object_list(verdict, src=0)
            if verdict[0] is not None:
                if world > 1:
                    dist.destroy_process_group()
                if isinstance(verdict[0], str):
                    try:
                        import optuna

                        raise optuna.TrialPruned()
                    except ImportError:
                        raise RuntimeError("trial pruned") from None
                raise verdict[0]
    best_score = 0.0
    if rank == 0 and objectives:
        # the LAST occurrence of the highest score wins (train.py:514-521)
        scores = np.array(objectives)[::-1]
        best = len(scores) - int(np.argmax(scores)) - 1
        final = "model_weights/final_{:s}.pt".format(exp_name)
        shutil.copyfile(model_paths[best], final)
        for p in model_paths:
            if path.exists(p):
                os.remove(p)
        best_score = float(objectives[best])
        if verbose:
            print("Highest matthews coefficient was {:.1f}% in epoch {:d}".format(
                best_score, (best + 1) * args.test_interval * (reps if args.train_federated else 1)))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return best_score


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
    args = Arguments(cmd_args, config, mode="train", verbose=int(os.environ.get("RANK", 0)) == 0)
    if args.train_federated and (args.mixup or args.weight_classes):
        if args.mixup and args.mixup_lambda == 0.5:
            args.mixup_lambda = 0.499
    if int(os.environ.get("RANK", 0)) == 0:
        print(str(args))
    main(args, cmd_args=cmd_args)
