"""Device-side data path in front of the training step: MixUp, one-hot targets,
dataset mean/std and the client split — the reference's torchlib classes with the arithmetic on the GPU.

  MixUp(λ, p)            torchlib/utils.py:327-400   (same constructor, same use of random.random())
  To_one_hot(classes)    torchlib/utils.py:444-466
  calc_mean_std(data)    torchlib/dataloader.py:220-247
  iid_round_robin_split  data/server_simulation/distribute_data.py:62-74 (seed 0, i::num_workers)
  label_skew_split       the non-IID shards BASELINE.json configs[2] asks for (the reference only ships the
                         IID split): each client draws its class proportions from Dirichlet(alpha).

The HIP kernels are mandatory (primia_amd._lib raises when the library is missing); there is no CPU fallback.
"""
from random import random
from typing import Optional

import numpy as np
import torch

from ._lib import call, query

_DEV = "cuda"


def _dev(t):
    return t if t.is_cuda else t.to(_DEV)


class MixUp(torch.nn.Module):
    """MixUp over one batch (torchlib/utils.py:327-400): sample i of the first half is blended with sample i of the
    second half, `λ·a + (1-λ)·b`, for data and one-hot targets alike; an odd batch keeps its last sample unmixed.

    Behaviour kept from the reference because seeded runs depend on it: `p` must be a number in [0, 1] and, when
    non-zero, ONE `random()` draw decides whether the call mixes at all; a falsy `λ` (None or 0) means ONE further
    draw per mixing call; inputs may be a batch tensor or a tuple of equally shaped per-sample tensors (the
    collate-free form), in which case a skipped call returns the first sample.  The blend runs in `primia_mixup`."""

    def __init__(self, λ: Optional[float] = None, p: Optional[float] = None):
        super().__init__()
        if not 0.0 <= p <= 1.0:
            raise AssertionError("MixUp: p is a probability, got %r" % (p,))
        if λ and not 0.0 <= λ <= 1.0:
            raise AssertionError("MixUp: the mix factor lies in [0, 1], got %r" % (λ,))
        self.p, self.λ = p, λ

    @staticmethod
    def _count(items, what):
        """Number of samples of a batch tensor or of a tuple of same-shaped tensors."""
        if torch.is_tensor(items):
            return items.shape[0]
        shapes = {tuple(t.shape) for t in items} if isinstance(items, (tuple, list)) else None
        if shapes is None or len(shapes) != 1:
            raise ValueError(f"MixUp: {what} must be one batch tensor or a tuple of equally shaped tensors")
        return len(items)

    def forward(self, pair):
        if len(pair) != 2:
            raise AssertionError("MixUp takes (data, target)")
        data, target = pair
        if self.p and random() > self.p:                      # this call passes its input through
            return (data, target) if torch.is_tensor(data) else (data[0], target[0])
        if not torch.is_tensor(data) and type(data) != tuple:   # the reference accepts tuples only
            raise ValueError("MixUp: data must be one batch tensor or a tuple of equally shaped tensors")
        n = self._count(data, "data")
        if self._count(target, "targets") != n:
            raise ValueError("MixUp: one (one-hot) target per sample is needed")
        if n == 1:
            return data, target
        lam = self.λ or random()
        batch = [t if torch.is_tensor(t) else torch.stack(tuple(t)).squeeze(1) for t in (data, target)]
        return self._mix(batch[0], lam), self._mix(batch[1], lam)

    @staticmethod
    def _mix(t, lam):
        t = _dev(t).to(torch.float32).contiguous()
        L = t.shape[0]
        out = torch.empty(((L + 1) // 2, *t.shape[1:]), dtype=torch.float32, device=t.device)
        per = t[0].numel()
        # the scalars as torch sees them: float32(λ) and float32(1.0 - λ), the latter formed in double
        call("primia_mixup", t, out, L, per, float(np.float32(lam)), float(np.float32(1.0 - lam)))
        return out


class To_one_hot(torch.nn.Module):
    def __init__(self, num_classes, device=_DEV):
        super().__init__()
        self.num_classes = num_classes
        self.device = device

    def forward(self, x):
        if type(x) in (int, list):
            x = torch.tensor(x)
        scalar = x.dim() == 0
        lab = _dev(x.reshape(-1).to(torch.int64)).contiguous()
        out = torch.empty(lab.shape[0], self.num_classes, dtype=torch.float32, device=lab.device)
        call("primia_to_one_hot", lab, out, lab.shape[0], self.num_classes)
        return out[0] if scalar else out


def calc_mean_std(data):
    """`data`: the stacked dataset [N, C, ...] (or a list / dataset of equally shaped samples).  Per-channel
    statistics when C is 1 or 3, over everything otherwise — exactly the reference's rule.  Returns (mean, std)."""
    if not torch.is_tensor(data):
        data = torch.stack([d[0] if isinstance(d, (tuple, list)) else d for d in data])
    data = _dev(data).to(torch.float32).contiguous()
    N, C = data.shape[0], data.shape[1]
    if C in (1, 3):
        n, c, hw = N, C, data[0, 0].numel()
    else:
        n, c, hw = 1, 1, data.numel()
    mean = torch.empty(c, dtype=torch.float32, device=data.device)
    std = torch.empty(c, dtype=torch.float32, device=data.device)
    wsb = query("primia_channel_stats_workspace_bytes", c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=data.device)
    call("primia_channel_mean_std", data, n, c, hw, mean, std, ws, wsb)
    return (mean, std) if C in (1, 3) else (mean[0], std[0])


def iid_round_robin_split(n_items, num_workers, seed=0):
    """The reference's split (distribute_data.py:62-74): shuffle with random.seed(0), deal i::num_workers."""
    import random as _r

    idx = list(range(n_items))
    rng = _r.Random()
    rng.seed(seed)
    rng.shuffle(idx)
    return [idx[i::num_workers] for i in range(num_workers)]


def label_skew_split(labels, num_workers, alpha=0.5, seed=0):
    """Non-IID client shards by label skew: for every class the samples are divided between the clients in
    proportions drawn from Dirichlet(alpha) (small alpha = strong skew).  Every sample is assigned exactly
    once; deterministic in `seed`.  Returns a list of index lists."""
    labels = np.asarray(labels)
    rng = np.random.default_rng(seed)
    shards = [[] for _ in range(num_workers)]
    for c in np.unique(labels):
        idx = np.flatnonzero(labels == c)
        rng.shuffle(idx)
        cuts = (np.cumsum(rng.dirichlet([alpha] * num_workers)) * len(idx)).astype(int)[:-1]
        for w, part in enumerate(np.split(idx, cuts)):
            shards[w].extend(part.tolist())
    for s in shards:
        s.sort()
    return shards


def register_federated(samples, onehot, orders, mix: Optional[MixUp]):
    """One worker's dataset registration (setup_pysyft, torchlib/utils.py:694-734) on device-resident data.

    samples [n, C, H, W] fp32 and onehot [n, classes] fp32 live on the GPU; `orders` holds one index sequence per
    dataset repetition (with MixUp the reference walks a freshly shuffled DataLoader every repetition; without it, the
    dataset order).  With `mix`, sample k of the walk is blended with the UNMIXED sample k-1 of the walk through the
    reference's own call `mixup(((d, last_d), (t, last_t)))` — same `random()` draws (one for p, one more for an unset
    λ), the blend itself in `primia_mixup`.  Returns (data [n_total, C, H, W], targets [n_total, classes])."""
    data = torch.empty((sum(len(o) for o in orders), *samples.shape[1:]), dtype=torch.float32, device=samples.device)
    tgts = torch.empty((data.shape[0], onehot.shape[1]), dtype=torch.float32, device=samples.device)
    pos, last = 0, None
    for order in orders:
        for k in order:
            d, t = samples[k:k + 1], onehot[k:k + 1]
            if mix is not None:
                original = (d, t)
                if last is not None:
                    d, t = mix(((d, last[0]), (t, last[1])))
                last = original
            data[pos].copy_(d.reshape(samples.shape[1:]))
            tgts[pos].copy_(t.reshape(-1))
            pos += 1
    return data, tgts


def class_counts(train_loader, num_classes):
    """Occurrences of every class over the training targets (the counting loop of calc_class_weights,
    torchlib/utils.py:479-497).  A loader that holds its targets (`.targets`: hard labels or one-hot / mixed rows, which
    count for their arg-max class) is counted from them directly — the ragged last batch included, as the reference's
    loader yields it, and without drawing a single augmentation or shuffle; anything else is iterated."""
    loaders = list(train_loader.values()) if isinstance(train_loader, dict) else [train_loader]
    occ = torch.zeros(num_classes, dtype=torch.float64)

    def count(target):
        if target.dim() == 2:
            target = target.max(dim=1)[1]
        return torch.bincount(target.reshape(-1).to(torch.int64), minlength=num_classes)[:num_classes].double().cpu()

    for tl in loaders:
        held = getattr(tl, "targets", None)
        if held is not None:
            occ += count(held)
        else:
            for _, target in tl:
                occ += count(target)
    return occ.to(torch.float32)


def class_weights_from_counts(occ):
    """torchlib/utils.py:498-513: 1 / count, normalised to sum 1; no target at all -> ones (the reference warns)."""
    from warnings import warn

    if torch.sum(occ).item() == 0:
        warn("class weights could not be calculated - no weights are used")
        return torch.ones((occ.numel(),))
    cw = 1.0 / occ
    cw /= torch.sum(cw)
    return cw


def calc_class_weights(args, train_loader, num_classes):
    """torchlib/utils.py:469-513: every class's share of 1 / (its number of training targets), normalised to sum 1.
    `train_loader` is one loader, or {worker: loader} when federated; soft / one-hot targets (mixup or federated
    weight_classes) count for their arg-max class."""
    return class_weights_from_counts(class_counts(train_loader, num_classes))


def random_split(dataset, lengths, generator=None):
    """torchlib/dataloader.py:440-450 (torch 1.5's random_split, vendored there because torch 1.4 took no generator):
    a seeded permutation of the indices, cut into consecutive Subsets of the given lengths."""
    import torch
    from torch.utils.data import Subset

    if sum(lengths) != len(dataset):
        raise ValueError("Sum of input lengths does not equal the length of the input dataset!")
    indices = torch.randperm(sum(lengths), generator=generator).tolist()
    out, offset = [], 0
    for length in lengths:
        offset += length
        out.append(Subset(dataset, indices[offset - length:offset]))
    return out

