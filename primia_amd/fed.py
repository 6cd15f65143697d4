"""Federated averaging across clients, one client per GPU / process.

Restates the exchange step of the reference's federated epoch —
`aggregation()` torchlib/utils.py:1000-1092, `send_new_models()` :1095-1105 and the sync schedule of
`secure_aggregation_epoch()` :1108-1233 — as collectives on the flat parameter arena:

  reference (hub and spoke, per state-dict key)         here (per client rank k, whole arena)
  ------------------------------------------------       ---------------------------------------------
  plaintext: sum_k (theta_k * w_k)  [/K if unweighted]   scale by w_k -> all_reduce(SUM) [-> /K]
  secure   : fix_prec(theta_k * w_k).share().get(),      scale by w_k -> fx_encode -> all_reduce(SUM) on
             ring sum of shares, reconstruct, float_prec int64 (wraps mod 2^64) -> fx_decode [-> /K]
             [/K if unweighted]                          (additive sharing commutes with ring addition, so
                                                         the reconstructed value is bit-identical)
  send_new_models to workers that still have batches     a rank adopts the average only if it still has
                                                         batches (mid-epoch) — always at the epoch end

`num_batches_tracked` never takes part (utils.py:1028,1040): it is not in the arena.
The arithmetic on the arena (scale / encode / decode) runs as HIP kernels (`HipArenaOps`); there is
no CPU implementation in this package.
"""
import torch
import torch.distributed as dist

from . import _lib


class HipArenaOps:
    """Element-wise arena arithmetic on the GPU through the C ABI."""

    def scale(self, x, a):
        _lib.call("primia_scale", x, x.numel(), float(a))

    def divide(self, x, d):
        _lib.call("primia_divide", x, x.numel(), float(d))

    def encode(self, x, q, scale):
        _lib.call("primia_fx_encode", x, q, x.numel(), float(scale))

    def decode(self, q, x, scale):
        _lib.call("primia_fx_decode", q, x, x.numel(), float(scale))


def fedavg_allreduce(flat, out, weight=None, secure=False, precision_fractional=16, base=10, group=None,
                     ops=None, scratch=None):
    """Average the clients' arenas.

    flat   : this client's fp32 arena (not modified)
    out    : fp32 tensor of the same size receiving the average (may alias nothing else)
    weight : this client's w_k, or None for the unweighted mean (sum, then / K)
    secure : reproduce the fixed-precision encode -> ring sum -> decode numerics
    """
    ops = ops or HipArenaOps()
    K = dist.get_world_size(group) if dist.is_initialized() else 1
    out.copy_(flat)
    if weight is not None:
        ops.scale(out, weight)
    if secure:
        scale = float(base ** precision_fractional)
        q = scratch if scratch is not None else torch.empty(out.numel(), dtype=torch.int64, device=out.device)
        ops.encode(out, q, scale)
        if K > 1:
            dist.all_reduce(q, op=dist.ReduceOp.SUM, group=group)
        ops.decode(q, out, scale)
    elif K > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    if weight is None:
        ops.divide(out, float(K))
    return out


def exchange_mean_std(mean, std, group=None, ops=None, precision_fractional=3, base=10):
    """setup_pysyft's secure average of the clients' data statistics (torchlib/utils.py:764-794): each rank
    holds its local (mean, std); they are fixed-point encoded (`fix_precision()` defaults: 10^3), summed in
    the 2^64 ring across the ranks (one int64 all-reduce — additive sharing commutes with the ring sum),
    decoded and divided by the number of clients.  Returns (mean, std) as the reference's `val_mean_std`."""
    ops = ops or HipArenaOps()
    K = dist.get_world_size(group) if dist.is_initialized() else 1
    both = torch.cat([mean.reshape(-1), std.reshape(-1)]).to(torch.float32).contiguous()
    q = torch.empty(both.numel(), dtype=torch.int64, device=both.device)
    scale = float(base ** precision_fractional)
    ops.encode(both, q, scale)
    if K > 1:
        dist.all_reduce(q, op=dist.ReduceOp.SUM, group=group)
    out = torch.empty_like(both)
    ops.decode(q, out, scale)
    ops.divide(out, float(K))
    n = mean.numel()
    return out[:n].reshape(mean.shape), out[n:].reshape(std.shape)


class SyncSchedule:
    """Which batches end with a FedAvg sync and who adopts the result — the control flow of
    secure_aggregation_epoch (utils.py:1159-1230), shared by every rank."""

    def __init__(self, num_batches_per_client, sync_every_n_batch):
        self.num_batches = list(num_batches_per_client)
        self.max_batches = max(self.num_batches)
        self.sync_every = int(sync_every_n_batch)

    def trains(self, client, batch_idx):
        return batch_idx < self.num_batches[client]

    def sync_after(self, batch_idx):
        return batch_idx > 0 and batch_idx % self.sync_every == 0

    def adopts(self, client, batch_idx):
        """send_new_models is restricted to workers with num_batches > batch_idx (utils.py:1189-1194)."""
        return self.num_batches[client] > batch_idx


def all_gather_int(value, group=None, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [int(value)]
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)
    return [int(o.item()) for o in outs]


def federated_epoch(engine, loader, lr, weight_decay, sync_every_n_batch, weighted_averaging=False,
                    secure=False, precision_fractional=16, optimizer="SGD", betas=(0.9, 0.999),
                    keep_optim_dict=False, soft_targets=False, group=None, ops=None, local_flat=None):
    """One federated epoch for THIS rank's client (the body of secure_aggregation_epoch).

    Returns (mean of this client's per-step losses, number of steps, local_flat) where local_flat
    holds the last global average ("local_model" in the reference)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = engine.flat.device
    counts = all_gather_int(len(loader), group, dev)
    sched = SyncSchedule(counts, sync_every_n_batch)
    weight = None
    if weighted_averaging:
        weight = counts[rank] / float(sum(counts))
    if local_flat is None:
        local_flat = torch.empty_like(engine.flat)
    scratch = torch.empty(engine.flat.numel(), dtype=torch.int64, device=dev) if secure else None
    if not keep_optim_dict:
        engine.reset_optimizer()
    losses = []
    it = iter(loader)

    def sync(final, batch_idx):
        fedavg_allreduce(engine.flat, local_flat, weight, secure, precision_fractional, 10, group, ops, scratch)
        if final or sched.adopts(rank, batch_idx):
            engine.flat.copy_(local_flat)
            engine.refresh_weights()
        if not keep_optim_dict:
            engine.reset_optimizer()

    for batch_idx in range(sched.max_batches):
        if sched.trains(rank, batch_idx):
            data, target = next(it)
            engine.forward(data)
            losses.append(engine.loss_backward(target, soft=soft_targets).clone())
            if optimizer == "SGD":
                engine.sgd_step(lr, weight_decay)
            else:
                engine.adam_step(lr, betas, 1e-8, weight_decay)
        if sched.sync_after(batch_idx):
            sync(False, batch_idx)
    sync(True, sched.max_batches)
    mean_loss = float(torch.stack(losses).mean().item()) if losses else float("nan")
    return mean_loss, len(losses), local_flat
