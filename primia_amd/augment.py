"""The training-time transform chain of the reference, create_albu_transform (torchlib/dataloader.py:138-217), with the
image arithmetic on the GPU (csrc/augment.hip) and the random PARAMETERS drawn on the host:

    transforms.RandomAffine(rotation, (translate, translate), (1 - scale, 1 + scale), shear)      [PIL image]
    a.Resize(R, R) -> a.RandomCrop(S, S) [-> a.FromFloat -> a.CLAHE(always_apply, clip_limit (1, 1))]
    a.Compose([a.VerticalFlip, a.RandomGamma, a.RandomBrightness, a.Blur, ..., a.GaussNoise], p = albu_prob)
    a.ToFloat(255) -> a.Normalize(mean, std, 1.0)

Built: RandomAffine, Resize, RandomCrop, CLAHE, VerticalFlip, RandomGamma, RandomBrightness, Blur, GaussNoise, ToFloat,
Normalize.  NOT built (warping / synthetic-weather transforms of albumentations): elastic, optical_distortion,
grid_distortion, grid_shuffle, hsv, invert, cutout, shadow, fog, sun_flare, solarize, equalize, grid_dropout — a
configuration that enables one of them is REFUSED (`unsupported(args)`), unless PRIMIA_SKIP_UNSUPPORTED_AUG=1 asks to
train without them (a warning names what was dropped).

Draw order (Python's `random`, as torchvision's RandomAffine.get_params and albumentations' BasicTransform.__call__ use
it): affine angle, [translate x, y], scale, shear; crop h, w; Compose coin; then per enabled transform its own coin and,
if it fires, its parameters.  GaussNoise's per-pixel normal values come from torch's generator on the device
(albumentations uses NumPy's RandomState, whose stream cannot be reproduced here).  cv2 / albumentations / Pillow are
not in this image: the chain follows their published behaviour and is unpinned against their binaries.
"""
import math
import os
from warnings import warn

import numpy as np
import torch

from ._lib import call, query

UNBUILT = ("elastic", "optical_distortion", "grid_distortion", "grid_shuffle", "hsv", "invert", "cutout", "shadow", "fog",
           "sun_flare", "solarize", "equalize", "grid_dropout")


def unsupported(args):
    return [k for k in UNBUILT if getattr(args, k, False)]


def check(args):
    """Refuse a configuration whose augmentations cannot be honoured (or, on request, say what is dropped)."""
    bad = unsupported(args)
    if not bad or not getattr(args, "albu_prob", 0) or not getattr(args, "individual_albu_probs", 0):
        return
    msg = ("the albumentations transforms {:s} are not part of the accelerated data path".format(", ".join(bad)))
    if os.environ.get("PRIMIA_SKIP_UNSUPPORTED_AUG") == "1":
        warn(msg + ": training WITHOUT them (PRIMIA_SKIP_UNSUPPORTED_AUG=1)")
    else:
        raise SystemExit(msg + "; switch them off in the [albumentations] section, or set PRIMIA_SKIP_UNSUPPORTED_AUG=1 "
                               "to train without them")


def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision 0.5 `_get_inverse_affine_matrix` (one shear angle, degrees): output pixel -> source pixel."""
    angle, shear = math.radians(angle), math.radians(shear)
    scale = 1.0 / scale
    d = math.cos(angle + shear) * math.cos(angle) + math.sin(angle + shear) * math.sin(angle)
    m = [math.cos(angle + shear), math.sin(angle + shear), 0, -math.sin(angle), math.cos(angle), 0]
    m = [scale / d * v for v in m]
    m[2] += m[0] * (-center[0] - translate[0]) + m[1] * (-center[1] - translate[1])
    m[5] += m[3] * (-center[0] - translate[0]) + m[4] * (-center[1] - translate[1])
    m[2] += center[0]
    m[5] += center[1]
    return m


def gamma_table(gamma):
    return (np.power(np.arange(0, 256.0 / 255, 1.0 / 255)[:256], gamma) * 255).astype(np.uint8)


def brightness_table(alpha, beta):
    lut = np.arange(0, 256, dtype=np.float32)
    if alpha != 1:
        lut *= np.float32(alpha)
    if beta != 0:
        lut += np.float32(beta * 255.0)
    return np.clip(lut, 0, 255).astype(np.uint8)


class TrainTransform:
    """create_albu_transform(args, mean, std) for device-resident uint8 HWC images: `tf(img, rng) -> fp32 [C, S, S]`."""

    def __init__(self, args, mean, std, device, channels, seed=0):
        check(args)
        from types import SimpleNamespace

        # (keys a hand-built `args` may lack count as switched off, like an INI with every probability at zero)
        keys = dict(rotation=0.0, translate=0.0, scale=0.0, shear=0.0, albu_prob=0.0, individual_albu_probs=0.0,
                    noise_std=0.0, noise_prob=0.0, clahe=False, randomgamma=False, randombrightness=False, blur=False)
        self.cfg = SimpleNamespace(inference_resolution=args.inference_resolution, train_resolution=args.train_resolution,
                                   **{k: getattr(args, k, v) for k, v in keys.items()})
        self.device, self.C = torch.device(device), channels
        self.mean = None if mean is None else mean.to(device).float().reshape(-1).contiguous()
        self.std = None if std is None else std.to(device).float().reshape(-1).contiguous()
        self.gen = torch.Generator(device=self.device).manual_seed(seed)          # GaussNoise values
        S = args.train_resolution
        self.ws_bytes = query("primia_clahe_workspace_bytes", S, S, channels)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)

    def __call__(self, img, rng, augment=True):
        dev, C = self.device, self.C
        a = self.cfg
        R, S = a.inference_resolution, a.train_resolution
        H, W = img.shape[0], img.shape[1]
        if augment and (a.rotation or a.translate or a.scale or a.shear):
            # RandomAffine.get_params (torchvision 0.5): angle, translations (rounded pixels), scale, shear
            angle = rng.uniform(-a.rotation, a.rotation)
            max_dx, max_dy = a.translate * W, a.translate * H
            tr = (np.round(rng.uniform(-max_dx, max_dx)), np.round(rng.uniform(-max_dy, max_dy)))
            sc = rng.uniform(1.0 - a.scale, 1.0 + a.scale)
            sh = rng.uniform(-a.shear, a.shear)
            m = inverse_affine_matrix((W * 0.5 + 0.5, H * 0.5 + 0.5), angle, tr, sc, sh)
            warped = torch.empty_like(img)
            call("primia_image_affine_u8", img, H, W, C, *[float(v) for v in m], warped)
            img = warped
        oy, ox = int((R - S) * rng.random()), int((R - S) * rng.random())             # a.RandomCrop
        cur = torch.empty(S, S, C, dtype=torch.uint8, device=dev)
        call("primia_image_resize_crop_u8", img, H, W, C, R, oy, ox, S, 0, cur)
        if a.clahe:
            call("primia_clahe_u8", cur, S, S, C, 1.0, self.ws, self.ws_bytes, cur)        # clip_limit = (1, 1)
        if augment and rng.random() < a.albu_prob:                                     # a.Compose(train_tf_albu, p)
            p = a.individual_albu_probs
            if rng.random() < p:                                                       # a.VerticalFlip
                cur = torch.flip(cur, dims=[0]).contiguous()
            if a.randomgamma and rng.random() < p:                  # gamma_limit (80, 120)
                t = torch.from_numpy(gamma_table(rng.randint(80, 120) / 100.0)).to(dev)
                call("primia_image_lut_u8", cur, cur.numel(), t, cur)
            if a.randombrightness and rng.random() < p:             # limit 0.2, contrast fixed at 1
                alpha = 1.0 + rng.uniform(0.0, 0.0)
                beta = 0.0 + rng.uniform(-0.2, 0.2)
                t = torch.from_numpy(brightness_table(alpha, beta)).to(dev)
                call("primia_image_lut_u8", cur, cur.numel(), t, cur)
            if a.blur and rng.random() < p:                         # blur_limit 7
                k = rng.choice(list(range(3, 8, 2)))
                out = torch.empty_like(cur)
                call("primia_image_box_blur_u8", cur, S, S, C, k, out)
                cur = out
            if rng.random() < a.noise_prob:                                            # a.GaussNoise(var_limit = noise_std^2)
                var = rng.uniform(0.0, a.noise_std ** 2)
                noise = torch.randn(cur.numel(), generator=self.gen, device=dev) * (var ** 0.5)
                call("primia_image_add_noise_u8", cur, noise, cur.numel(), cur)
        out = torch.empty(C, S, S, dtype=torch.float32, device=dev)
        call("primia_image_finish", cur, S, C, self.mean, self.std, out)
        return out
