"""GPU: the image-folder path (primia_amd.imagefolder, primia_image_prepare) against a plain torch-CPU restatement of
a.Resize -> a.RandomCrop -> a.ToFloat -> a.Normalize (torchlib/dataloader.py:138-217), and the CLI on a real folder.
Parity is UNPINNED for the resize: albumentations / cv2 are not in this image, so the comparison is against bilinear
interpolation with half-pixel centres (cv2.INTER_LINEAR's definition) evaluated in float; cv2's 8-bit path uses 11-bit
fixed-point weights and may land one grey level away on ties — tolerance 1 / 255 (before normalisation)."""
import os
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from primia_amd import imagefolder  # noqa: E402
from primia_amd._lib import call  # noqa: E402


def make_folder(root, per_class=5, seed=0, channels=3):
    from PIL import Image

    rng = np.random.RandomState(seed)
    for c in ("bacterial", "normal", "viral"):
        os.makedirs(os.path.join(root, c), exist_ok=True)
        for i in range(per_class):
            h, w = rng.randint(40, 90), rng.randint(40, 90)
            a = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
            Image.fromarray(a).save(os.path.join(root, c, f"img{i}.png"))


def reference_prepare(img_u8, R, oy, ox, S, mean=None, std=None):
    x = torch.from_numpy(img_u8).permute(2, 0, 1).float().unsqueeze(0)
    r = F.interpolate(x, size=(R, R), mode="bilinear", align_corners=False)[0]
    r = torch.clamp(torch.floor(r + 0.5), 0, 255)[:, oy:oy + S, ox:ox + S] / 255.0
    if mean is not None:
        r = (r - mean.view(-1, 1, 1)) / std.view(-1, 1, 1)
    return r


@pytest.mark.parametrize("channels", [3, 1])
def test_image_prepare_matches_bilinear_reference(cuda, tmp_path, channels):
    make_folder(str(tmp_path))
    classes, samples = imagefolder.scan(str(tmp_path))
    assert classes == ["bacterial", "normal", "viral"] and len(samples) == 15
    assert [t for _, t in samples] == [0] * 5 + [1] * 5 + [2] * 5
    R, S = 72, 64
    mean = torch.tensor([0.4, 0.5, 0.6][:channels])
    std = torch.tensor([0.2, 0.25, 0.3][:channels])
    rng = random.Random(3)
    for fn, _ in samples[:6]:
        img = imagefolder.decode(fn, channels)
        oy, ox = imagefolder.crop_offsets(R, S, rng)
        for m, s in ((None, None), (mean, std)):
            out = torch.empty(channels, S, S, device=cuda)
            call("primia_image_prepare", torch.from_numpy(np.ascontiguousarray(img)).to(cuda), img.shape[0],
                 img.shape[1], channels, R, oy, ox, S, 0, None if m is None else m.to(cuda),
                 None if s is None else s.to(cuda), out)
            want = reference_prepare(img, R, oy, ox, S, m, s)
            diff = (out.cpu() - want).abs()
            tol = 1.0 / 255 / (1.0 if s is None else s.min().item()) + 1e-6
            assert diff.max() <= tol
            assert (diff > 1e-6).float().mean() < 2e-3          # apart from rounding ties the levels agree exactly
    # vertical flip = the same image upside down
    img = imagefolder.decode(samples[0][0], channels)
    a, b = torch.empty(channels, S, S, device=cuda), torch.empty(channels, S, S, device=cuda)
    d = torch.from_numpy(np.ascontiguousarray(img)).to(cuda)
    call("primia_image_prepare", d, img.shape[0], img.shape[1], channels, S, 0, 0, S, 0, None, None, a)
    call("primia_image_prepare", d, img.shape[0], img.shape[1], channels, S, 0, 0, S, 1, None, None, b)
    assert torch.equal(a.flip(1), b)


def test_client_loader_statistics_and_batches(cuda, tmp_path):
    make_folder(str(tmp_path), per_class=6)
    args = SimpleNamespace(inference_resolution=64, train_resolution=64, batch_size=4, repetitions_dataset=2,
                           train_federated=True)       # a federated client registers its dataset repetitions_dataset times
    loader, (mean, std) = imagefolder.client_loader(str(tmp_path), args, cuda, 3, seed=1)
    assert len(loader) == (18 * 2) // 4
    args5 = SimpleNamespace(inference_resolution=64, train_resolution=64, batch_size=5, repetitions_dataset=1,
                            train_federated=True)
    l5, _ = imagefolder.client_loader(str(tmp_path), args5, cuda, 3, seed=1)
    assert len(l5) == 4 and [x.shape[0] for x, _ in l5] == [5, 5, 5, 3]      # FederatedDataLoader: drop_last = False
    # the dataset is normalised with its own statistics: per-channel mean 0 / std 1 (torch.std_mean, unbiased)
    s, m = torch.std_mean(loader.data, dim=(0, 2, 3))
    assert m.abs().max() < 1e-4 and (s - 1).abs().max() < 1e-3
    seen = []
    for x, y in loader:
        assert x.shape == (4, 3, 64, 64) and y.shape == (4,) and x.is_cuda
        seen += y.tolist()
    assert len(seen) == 36 and set(seen) == {0, 1, 2}
    with pytest.raises(ValueError):
        imagefolder.client_loader(str(tmp_path), SimpleNamespace(inference_resolution=64, train_resolution=64,
                                                                 batch_size=64, repetitions_dataset=1), cuda, 3, 1)
    # vanilla training: a loader that re-augments every epoch; with CLAHE on, the data differ from the plain chain
    van = SimpleNamespace(inference_resolution=64, train_resolution=64, batch_size=4, train_federated=False, clahe=True,
                          rotation=10, scale=0.1, albu_prob=0.5, individual_albu_probs=0.5, noise_std=0.05, noise_prob=0.5,
                          randomgamma=True, blur=True)
    vl, _ = imagefolder.client_loader(str(tmp_path), van, cuda, 3, seed=1)
    assert len(vl) == 5                       # DataLoader's default drop_last = False: 4 whole batches + the last 2 images
    e1 = [x.clone() for x, _ in vl]
    e2 = [x.clone() for x, _ in vl]
    assert [x.shape[0] for x in e1] == [4, 4, 4, 4, 2] and all(torch.isfinite(x).all() for x in e1)
    assert not all(torch.equal(a, b) for a, b in zip(e1, e2))


def test_validation_counts_every_sample(cuda):
    """torchlib/utils.py:1354-1467: the validation pass sees the WHOLE folder — 11 samples through an engine built for 8
    (the ragged tail of 3 runs on a sibling engine; eval-mode BatchNorm makes the logits independent of the batching) —
    and averages the loss per batch of test_batch_size, then over batches."""
    from primia_amd import resnet_spec as rs
    from primia_amd.engine import ResNet18Engine
    from primia_amd.torchlib_compat import test as validate

    torch.manual_seed(5)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, 64, "max"))
    g = torch.Generator().manual_seed(6)
    x = torch.randn(11, 3, 64, 64, generator=g).to(cuda)
    y = torch.randint(0, 3, (11,), generator=g).to(cuda)
    eng = ResNet18Engine(8, 3, 3, 64, "max", dtype=torch.float32, device=cuda)
    eng.load_state_dict(sd)
    loader = imagefolder.DeviceLoader(x, y, 8, False, 0, drop_last=False)
    assert len(loader) == 2 and [d.shape[0] for d, _ in loader] == [8, 3]
    assert len(imagefolder.DeviceLoader(x, y, 8, False, 0)) == 1
    whole = ResNet18Engine(11, 3, 3, 64, "max", dtype=torch.float32, device=cuda)
    whole.load_state_dict(sd)
    whole.eval()
    nll = -torch.log_softmax(whole.forward(x), dim=1).gather(1, y.view(-1, 1)).reshape(-1).double().cpu()
    for tbs, want in ((1, float(nll.mean())), (4, float(np.mean([float(nll[i:i + 4].mean()) for i in (0, 4, 8)])))):
        loss, _ = validate(SimpleNamespace(test_batch_size=tbs), eng, cuda, loader, 1, None, 3, verbose=False)
        assert abs(loss - want) < 1e-5 * abs(want), (tbs, loss, want)
    assert eng.training                      # test() leaves the model in training mode, like the reference
