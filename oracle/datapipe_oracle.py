"""ORACLE (test infrastructure, not product code) — CPU restatement of the device-side pieces of
PriMIA's data path that sit right in front of the training step and of the
secure mean/std exchange of setup_pysyft (§8f item 1).

Only tests/ and the golden-minting scripts may import this file.

What it restates (reference = /root/reference):
  * MixUp.forward                — torchlib/utils.py:327-400
  * To_one_hot.forward           — torchlib/utils.py:444-466
  * calc_mean_std                — torchlib/dataloader.py:220-247 (torch.std_mean over (N, H, W))
  * the IID round-robin split    — data/server_simulation/distribute_data.py:62-74
  * mean/std exchange            — torchlib/utils.py:764-794 (fix_precision() = 10^3, ring sum, / #workers)

Parity pin: tests/golden/make_datapipe_golden.py executes the reference's own MixUp / To_one_hot /
calc_mean_std definitions (extracted from the files under /root/reference at mint time) on seeded
inputs and checks this restatement bit for bit; the vectors are committed as tests/golden/datapipe.npz.
The split is pinned the same way (the dealing statements of distribute_data.py executed from the file, ImageFolder
replaced by a length).  The exchange is pinned by tests/golden/make_secure_ref_golden.py (mint_mean_std): the
reference's FixedPrecisionTensor / AdditiveSharingTensor classes run utils.py:764-794's chain and this restatement
matches bit for bit (tests/golden/mean_std_ref.npz).
"""
import random as _random

import torch


def mixup(x, y, lam=None, p=None, rng=_random):
    """MixUp.forward (utils.py:337-400).  `x`, `y`: tensors [L, ...] or tuples of equally shaped tensors.
    `rng.random()` is consumed exactly as the reference consumes `random()`: once for p, once for λ."""
    if p:
        if rng.random() > p:
            if torch.is_tensor(x):
                return x, y
            return x[0], y[0]
    L = x.shape[0] if torch.is_tensor(x) else len(x)
    if L == 1:
        return x, y
    lam = lam if lam else rng.random()
    if not torch.is_tensor(x):
        x = torch.stack(x).squeeze(1)
    if not torch.is_tensor(y):
        y = torch.stack(y).squeeze(1)
    if L % 2 == 0:
        h = L // 2
        return lam * x[:h] + (1.0 - lam) * x[h:], lam * y[:h] + (1.0 - lam) * y[h:]
    h = (L - 1) // 2
    out_x = torch.zeros((h + 1, *x.shape[1:]))
    out_y = torch.zeros((h + 1, *y.shape[1:]))
    out_x[-1], out_y[-1] = x[-1], y[-1]
    out_x[:-1] = lam * x[:h] + (1.0 - lam) * x[h:-1]
    out_y[:-1] = lam * y[:h] + (1.0 - lam) * y[h:-1]
    return out_x, out_y


def to_one_hot(x, num_classes):
    """To_one_hot.forward (utils.py:449-466): float32 one-hot rows; a scalar gives a vector."""
    if isinstance(x, (int, list)):
        x = torch.tensor(x)
    if x.dim() == 0:
        out = torch.zeros((num_classes,))
        out[int(x)] = 1.0
        return out
    x = x.reshape(-1)
    out = torch.zeros((x.shape[0], num_classes))
    out[torch.arange(x.shape[0]), x] = 1.0
    return out


def calc_mean_std(data):
    """calc_mean_std (dataloader.py:220-247) on the stacked data [N, C, ...]: per-channel statistics over
    every other axis when C is 1 or 3 ("ugly hack"), else over everything; unbiased std.  Returns (mean, std)."""
    if data.shape[1] in (1, 3):
        dims = (0, *range(2, data.dim()))
    else:
        dims = tuple(range(data.dim()))
    std, mean = torch.std_mean(data, dim=dims)
    return mean, std


def iid_round_robin_split(n_items, num_workers, seed=0):
    """distribute_data.py:62-74: indices shuffled with random.seed(0) and dealt i::num_workers."""
    idx = list(range(n_items))
    r = _random.Random()
    r.seed(seed)
    r.shuffle(idx)
    return [idx[i::num_workers] for i in range(num_workers)]


def exchange_mean_std(means, stds, precision_fractional=3, base=10):
    """setup_pysyft's secure average of the workers' data statistics (utils.py:764-794): every worker's mean
    and std are fixed-point encoded (fix_precision() defaults: base 10, 3 fractional digits), secret-shared,
    summed in the 2^64 ring, reconstructed, decoded and divided by the number of workers."""
    scale = base ** precision_fractional
    enc = lambda t: (t * scale).long()
    m = enc(means[0].clone())
    s = enc(stds[0].clone())
    for a, b in zip(means[1:], stds[1:]):
        m = m + enc(a)
        s = s + enc(b)
    return m.float() / scale / len(stds), s.float() / scale / len(stds)


def register_federated(samples, targets, orders, mix, lam=None, p=None, rng=_random):
    """The registration loop of setup_pysyft (torchlib/utils.py:694-734) for one worker.

    samples : per-sample tensors [C, H, W]; targets : per-sample one-hot rows [3] (mix or weight_classes) or ints.
    orders  : one index sequence per dataset repetition — with `mixup` the reference wraps the dataset in a
              DataLoader(batch_size=1, shuffle=True), so every repetition walks its own permutation; without it the
              dataset order (the caller passes range(n) `repetitions_dataset` times).
    mix     : args.mixup.  Sample k of the walk is blended with the UNMIXED sample k-1 of the walk (`last_set`
              carries across repetitions): MixUp(((d, last_d), (t, last_t))) on [1, ...] tensors, i.e. one
              `rng.random()` for p (a skipped call keeps d) and, when λ is unset, one more for λ.
    Returns (data [n_total, C, H, W], targets [n_total, 3] or [n_total])."""
    data, tgts = [], []
    last = None
    for order in orders:
        for k in order:
            d, t = samples[k], targets[k]
            if mix:
                d, t = d.unsqueeze(0), t.unsqueeze(0)           # the DataLoader's batch dimension
                original = (d, t)
                if last:
                    d, t = mixup((d, last[0]), (t, last[1]), lam, p, rng=rng)
                last = original
            data.append(d)
            tgts.append(t)
    sel = torch.stack(data)
    tt = torch.stack(tgts) if torch.is_tensor(tgts[0]) else torch.tensor(tgts)
    if mix:
        sel, tt = sel.squeeze(1), tt.squeeze(1)
    return sel, tt


def calc_class_weights(loaders, batch_size, num_classes, soft_targets):
    """calc_class_weights (torchlib/utils.py:469-513): occurrences of every class over all loaders' batches (soft or
    one-hot targets are reduced with max(dim=1) first — the first maximum wins), weights 1 / count normalised to
    sum 1; all-zero counts give ones."""
    occ = torch.zeros(num_classes)
    for tl in loaders:
        for _, target in tl:
            if soft_targets:
                target = target.max(dim=1)[1]
            for i in range(num_classes):
                occ[i] += float((target == i).sum().item())
    if torch.sum(occ).item() == 0:
        return torch.ones((num_classes,))
    cw = 1.0 / occ
    cw /= torch.sum(cw)
    return cw
