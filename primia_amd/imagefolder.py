"""Real image folders in front of the training step.

The reference registers every client's COMPLETE dataset on its worker before training starts
(torchlib/utils.py:612-739): an ImageFolder (`<root>/<class>/<image>`, classes sorted by name) is read once, a
statistics pass (Resize -> RandomCrop -> ToFloat) yields the client's per-channel mean / std, every image then goes
through create_albu_transform (torchlib/dataloader.py:138-217) with that mean / std, the stacked tensors are tagged
#traindata / #traintargets and a FederatedDataLoader (batch_size, shuffle=True) iterates them.

Here the decode (PIL, host) produces uint8 HWC arrays; everything after it runs on the GPU: `primia_image_prepare`
(resize, crop, to-float, normalise — one launch per image into the client's device-resident dataset tensor) and
`primia_mean_std`.  The transform chain is primia_amd.augment.TrainTransform: every member of the reference's
create_albu_transform on the GPU (both shipped presets run as written).

Loaders yield the ragged final batch as the reference's do (DataLoader / FederatedDataLoader, drop_last = False):
`len(loader)` = ceil(n / batch_size); such a batch — and the batches MixUp halves in the local training loop — run on a
sibling engine of their size (ResNet18Engine.sibling: same parameters, its own activations).
"""
import os
import random
from warnings import warn

import numpy as np
import torch

from ._lib import call

EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")

def scan(root):
    """torchvision.datasets.ImageFolder's listing: classes = sorted sub-directory names, samples sorted per class."""
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    samples = []
    for ci, c in enumerate(classes):
        for dirpath, _, files in sorted(os.walk(os.path.join(root, c))):
            for f in sorted(files):
                if f.lower().endswith(EXTENSIONS) and not f.startswith("._"):
                    samples.append((os.path.join(dirpath, f), ci))
    return classes, samples


def decode(filename, channels):
    """default_loader (RGB) / single_channel_loader ('L', torchlib/dataloader.py:250-255) -> uint8 [H, W, C]."""
    from PIL import Image

    with open(filename, "rb") as f:
        img = Image.open(f).convert("RGB" if channels == 3 else "L")
        a = np.asarray(img, dtype=np.uint8)
    return a if a.ndim == 3 else a[:, :, None]


def crop_offsets(R, S, rng):
    """albumentations.RandomCrop.get_params / functional.get_random_crop_coords: two uniform draws, h then w."""
    h_start, w_start = rng.random(), rng.random()
    return int((R - S) * h_start), int((R - S) * w_start)


def prepare(samples, args, device, channels, rng, mean=None, std=None):
    """All samples -> fp32 [n, C, S, S] on the device (one primia_image_prepare launch per image)."""
    R, S = args.inference_resolution, args.train_resolution
    out = torch.empty(len(samples), channels, S, S, dtype=torch.float32, device=device)
    for i, (fn, _) in enumerate(samples):
        img = torch.from_numpy(np.ascontiguousarray(decode(fn, channels))).to(device)
        oy, ox = crop_offsets(R, S, rng)
        call("primia_image_prepare", img, img.shape[0], img.shape[1], channels, R, oy, ox, S, 0, mean, std, out[i])
    return out


class DeviceLoader:
    """FederatedDataLoader(batch_size, shuffle=True) over a device-resident (data, targets) pair (drop_last as given:
    synthetic / fixed-size callers keep whole batches only)."""

    def __init__(self, data, targets, batch_size, shuffle, seed, drop_last=True):
        self.data, self.targets, self.batch_size, self.shuffle = data, targets, batch_size, shuffle
        self.gen = torch.Generator().manual_seed(seed)
        self.drop_last = drop_last       # False: the ragged final batch is yielded too (validation: every sample counts)

    def __len__(self):
        n, b = self.data.shape[0], self.batch_size
        return n // b if self.drop_last else (n + b - 1) // b

    def __iter__(self):
        n = self.data.shape[0]
        order = torch.randperm(n, generator=self.gen) if self.shuffle else torch.arange(n)
        order = order.to(self.data.device)
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield self.data.index_select(0, idx), self.targets.index_select(0, idx)


class AugmentingLoader:
    """DataLoader(dataset, batch_size, shuffle=True) over a transforming dataset: every epoch shuffles and sends each
    decoded (device-resident, uint8) image through the transform chain again."""

    def __init__(self, images, targets, transform, rng, batch_size, seed):
        self.images, self.targets, self.tf, self.rng, self.batch_size = images, targets, transform, rng, batch_size
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.images) + self.batch_size - 1) // self.batch_size      # DataLoader's default: drop_last = False

    def __iter__(self):
        order = torch.randperm(len(self.images), generator=self.gen).tolist()
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield torch.stack([self.tf(self.images[i], self.rng) for i in idx]), self.targets[torch.tensor(idx, device=self.targets.device)]


def register(data_per_rep, targets, args, num_classes, seed):
    """The registration of one worker's dataset (torchlib/utils.py:680-734).  `data_per_rep[r]` = the dataset as its
    r-th walk sees it (every walk passes each image through the stochastic transform chain again); with `mixup` or
    `weight_classes` the targets become one-hot rows (To_one_hot), and with `mixup` every walk is a fresh shuffle in
    which each sample is blended with the previous unmixed one.  Hard int64 targets otherwise."""
    from .datapipe import MixUp, To_one_hot, register_federated

    if torch.is_tensor(data_per_rep):
        data_per_rep = [data_per_rep]
    reps, n = len(data_per_rep), data_per_rep[0].shape[0]
    fed = bool(getattr(args, "train_federated", False))   # vanilla training mixes per batch instead (utils.py:1249-1267)
    mix = fed and bool(getattr(args, "mixup", False))
    if not fed or not (mix or getattr(args, "weight_classes", False)):
        return torch.cat(data_per_rep), targets.repeat(reps)
    onehot = To_one_hot(num_classes, device=targets.device)(targets)
    gen = torch.Generator().manual_seed(seed + 7919)          # the shuffled DataLoader of utils.py:697-703
    orders = [[r * n + k for k in (torch.randperm(n, generator=gen).tolist() if mix else range(n))] for r in range(reps)]
    mixer = MixUp(λ=getattr(args, "mixup_lambda", None), p=args.mixup_prob) if mix else None
    return register_federated(torch.cat(data_per_rep), onehot.repeat(reps, 1), orders, mixer)


def client_loader(root, args, device, channels, seed):
    """One client's registration (torchlib/utils.py:643-739): returns (loader, (mean, std)) with mean / std on the
    device (they take part in the secure mean/std exchange)."""
    from .datapipe import calc_mean_std

    classes, samples = scan(root)
    assert len(classes) == 3, "We can only handle data that has 3 classes: normal, bacterial and viral"
    if len(samples) < args.batch_size:
        raise ValueError("{:s}: {:d} images, fewer than one batch of {:d}".format(root, len(samples), args.batch_size))
    from .augment import TrainTransform

    rng = random.Random(seed)
    raw = prepare(samples, args, device, channels, rng)                       # Resize, RandomCrop, ToFloat
    mean, std = calc_mean_std(raw)
    del raw
    tf = TrainTransform(args, mean, std, device, channels, seed)              # create_albu_transform(args, mean, std)
    images = [torch.from_numpy(np.ascontiguousarray(decode(fn, channels))).to(device) for fn, _ in samples]
    targets = torch.tensor([t for _, t in samples], dtype=torch.int64, device=device)
    if not getattr(args, "train_federated", False):
        # vanilla training draws fresh augmentations every epoch (a DataLoader over the transforming dataset)
        return AugmentingLoader(images, targets, tf, rng, args.batch_size, seed), (mean, std)
    reps = int(getattr(args, "repetitions_dataset", 1) or 1)
    walks = [torch.stack([tf(img, rng) for img in images]) for _ in range(reps)]   # utils.py:704-717: one pass per walk
    data, targets = register(walks, targets, args, len(classes), seed)
    return DeviceLoader(data, targets, args.batch_size, True, seed, drop_last=False), (mean, std)


def validation_loader(root, args, device, channels, val_mean_std):
    """The validation folder, normalised with the exchanged mean / std (torchlib/utils.py:815-860): Resize to the
    inference resolution, centre crop to the train resolution when they differ, no shuffling."""
    classes, samples = scan(root)
    assert len(classes) == 3, "We can only handle data that has 3 classes: normal, bacterial and viral"
    R, S = args.inference_resolution, args.train_resolution
    mean = val_mean_std[0].to(device).float().contiguous()
    std = val_mean_std[1].to(device).float().contiguous()
    out = torch.empty(len(samples), channels, S, S, dtype=torch.float32, device=device)
    off = (R - S) // 2
    for i, (fn, _) in enumerate(samples):
        img = torch.from_numpy(np.ascontiguousarray(decode(fn, channels))).to(device)
        call("primia_image_prepare", img, img.shape[0], img.shape[1], channels, R, off, off, S, 0, mean, std, out[i])
    targets = torch.tensor([t for _, t in samples], dtype=torch.int64, device=device)
    return DeviceLoader(out, targets, args.batch_size, False, 0, drop_last=False)
