"""The training-time transform chain of the reference, create_albu_transform (torchlib/dataloader.py:138-217), with the
image arithmetic on the GPU (csrc/augment.hip) and the random PARAMETERS drawn on the host:

    transforms.RandomAffine(rotation, (translate, translate), (1 - scale, 1 + scale), shear)      [PIL image]
    a.Resize(R, R) -> a.RandomCrop(S, S) [-> a.FromFloat -> a.CLAHE(always_apply, clip_limit (1, 1))]
    a.Compose([a.VerticalFlip, a.RandomGamma, a.RandomBrightness, a.Blur, ..., a.GaussNoise], p = albu_prob)
    a.ToFloat(255) -> a.Normalize(mean, std, 1.0)

Built: every member of create_albu_transform — RandomAffine, Resize, RandomCrop, CLAHE, VerticalFlip, RandomGamma,
RandomBrightness, Blur, ElasticTransform, OpticalDistortion, GridDistortion, RandomGridShuffle, HueSaturationValue,
InvertImg, Cutout, RandomShadow, RandomFog, RandomSunFlare, Solarize, Equalize, GridDropout, GaussNoise, ToFloat,
Normalize — so both shipped presets (configs/torch/pneumonia-resnet-pretrained.ini and ...-fast.ini, which switches every
one of them on) run as written.  `UNBUILT` is empty; `check(args)` stays as the place a future gap would be refused.

The three warping transforms are `cv2.remap` with a generated coordinate field (albumentations 0.4.6, the release the
reference pins: functional.py elastic_transform / optical_distortion / grid_distortion): the host draws their few
parameters (ElasticTransform: a seed for numpy's RandomState, whose two uniform fields are the only bulk data that
crosses to the device), the maps are formed and sampled on the GPU (primia_warp_map_*, primia_image_remap_u8).

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

UNBUILT = ()


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


def affine_from_points(pts1, pts2):
    """cv2.getAffineTransform: the 2 x 3 float64 matrix mapping the three points pts1 onto pts2."""
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        x, y = float(pts1[i][0]), float(pts1[i][1])
        a[2 * i] = [x, y, 1, 0, 0, 0]
        a[2 * i + 1] = [0, 0, 0, x, y, 1]
        b[2 * i], b[2 * i + 1] = float(pts2[i][0]), float(pts2[i][1])
    return np.linalg.solve(a, b).reshape(2, 3)


def invert_affine(m):
    """cv2.invertAffineTransform (warpAffine samples the source through the inverse)."""
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22, a12, a21 = m[1, 1] * d, m[0, 0] * d, -m[0, 1] * d, -m[1, 0] * d
    return np.array([[a11, a12, -a11 * m[0, 2] - a12 * m[1, 2]], [a21, a22, -a21 * m[0, 2] - a22 * m[1, 2]]], np.float64)


def grid_axis(n, num_steps, steps):
    """One axis of albumentations' F.grid_distortion: float32 source positions of n pixels."""
    step = n // num_steps
    xx = np.zeros(n, np.float32)
    prev = 0
    for idx, x in enumerate(range(0, n, step)):
        start, end = x, x + step
        if end > n:
            end, cur = n, n
        else:
            cur = prev + step * steps[idx]
        xx[start:end] = np.linspace(prev, cur, end - start)
        prev = cur
    return xx


def fog_params(H, W, rng, fog_coef_lower=0.3, fog_coef_upper=1.0):
    """RandomFog.get_params_dependent_on_targets: fog_coef and the haze points, from Python's `random` stream."""
    fog_coef = rng.uniform(fog_coef_lower, fog_coef_upper)
    hw = max(1, int(W // 3 * fog_coef))
    haze = []
    midx, midy = W // 2 - 2 * hw, H // 2 - hw
    index = 1
    while midx > -hw or midy > -hw:
        for _ in range(hw // 10 * index):
            haze.append((rng.randint(midx, W - midx - hw), rng.randint(midy, H - midy - hw)))
        midx -= 3 * hw * W // (H + W)
        midy -= 3 * hw * H // (H + W)
        index += 1
    return fog_coef, haze


def cutout_holes(H, W, rng, num_holes=5, max_h_size=80, max_w_size=80):
    """a.Cutout.get_params_dependent_on_targets (dataloader.py:178-182: 5 holes of at most 80 x 80)."""
    holes = []
    for _ in range(num_holes):
        y, x = rng.randint(0, H), rng.randint(0, W)
        y1 = min(max(y - max_h_size // 2, 0), H)
        y2 = min(max(y1 + max_h_size, 0), H)
        x1 = min(max(x - max_w_size // 2, 0), W)
        x2 = min(max(x1 + max_w_size, 0), W)
        holes.append((x1, y1, x2, y2))
    return holes


def grid_dropout_holes(H, W, ratio=0.5):
    """a.GridDropout with its defaults: unit = max(2, W // 10), holes of ratio x unit at the unit grid's corners."""
    unit_w = max(2, W // 10)
    unit_h = max(min(unit_w, H), 2)
    hole_w = min(max(int(unit_w * ratio), 1), unit_w - 1)
    hole_h = min(max(int(unit_h * ratio), 1), unit_h - 1)
    holes = []
    for i in range(W // unit_w + 1):
        for j in range(H // unit_h + 1):
            x1, y1 = min(unit_w * i, W), min(unit_h * j, H)
            holes.append((x1, y1, min(x1 + hole_w, W), min(y1 + hole_h, H)))
    return holes


def grid_shuffle_tiles(H, W, seed, grid=(3, 3)):
    """a.RandomGridShuffle.get_params_dependent_on_targets: tiles of equal shape are permuted among themselves by
    np.random.RandomState(seed); rows (y, x, old_y, old_x, height, width)."""
    n, m = grid
    rs = np.random.RandomState(seed)
    hs = np.linspace(0, H, n + 1, dtype=np.int64)
    ws = np.linspace(0, W, m + 1, dtype=np.int64)
    hm, wm = np.meshgrid(hs, ws, indexing="ij")
    ih, iw = hm[:-1, :-1], wm[:-1, :-1]
    sizes = np.stack((hm[1:, 1:] - ih, wm[1:, 1:] - iw), axis=2)
    new_index = np.stack(np.indices((n, m)), axis=2)
    for size in np.unique(sizes.reshape(-1, 2), axis=0):
        eq = np.all(sizes == size, axis=2)
        new_index[eq] = rs.permutation(new_index[eq])
    a, b = new_index[..., 0], new_index[..., 1]
    return np.stack([ih.reshape(-1), iw.reshape(-1), ih[a, b].reshape(-1), iw[a, b].reshape(-1),
                     sizes[..., 0].reshape(-1), sizes[..., 1].reshape(-1)], axis=1).astype(np.int32)


def hsv_tables(hue_shift, sat_shift, val_shift):
    """F._shift_hsv_uint8's three cv2.LUT tables (hue mod 180; saturation and value clipped)."""
    i = np.arange(256, dtype=np.int16)
    return np.stack([np.mod(i + hue_shift, 180).astype(np.uint8), np.clip(i + sat_shift, 0, 255).astype(np.uint8),
                     np.clip(i + val_shift, 0, 255).astype(np.uint8)])


def solarize_table(threshold):
    i = np.arange(256)
    return np.where(i < threshold, i, 255 - i).astype(np.uint8)


def shadow_vertices(H, W, rng, shadow_roi=(0, 0.5, 1, 1), lower=1, upper=2, dimension=5):
    """a.RandomShadow.get_params_dependent_on_targets: [num_shadows][5][(x, y)] in the lower half of the image."""
    n = rng.randint(lower, upper)
    x_min, y_min, x_max, y_max = shadow_roi
    x_min, x_max, y_min, y_max = int(x_min * W), int(x_max * W), int(y_min * H), int(y_max * H)
    return np.array([[(rng.randint(x_min, x_max), rng.randint(y_min, y_max)) for _ in range(dimension)] for _ in range(n)],
                    np.int32)


def sun_flare_steps(H, W, rng, flare_roi=(0, 0, 1, 0.5), angle_lower=0.0, angle_upper=1.0, circles_lower=6,
                    circles_upper=10, src_radius=400, src_color=(255, 255, 255)):
    """a.RandomSunFlare.get_params_dependent_on_targets + the drawing schedule of F.add_sun_flare: rows
    (x, y, radius, r, g, b), their blend weights, and the step at which the overlay restarts from the output."""
    angle = 2 * math.pi * rng.uniform(angle_lower, angle_upper)
    lx, ly, ux, uy = flare_roi
    cx, cy = rng.uniform(lx, ux), rng.uniform(ly, uy)
    cx, cy = int(W * cx), int(H * cy)
    num = rng.randint(circles_lower, circles_upper)
    xs, ys = [], []
    for rx in range(0, W, 10):
        xs.append(rx)
        ys.append(2 * cy - (math.tan(angle) * (rx - cx) + cy))
    geo, alpha = [], []
    for _ in range(num):
        a = rng.uniform(0.05, 0.2)
        r = rng.randint(0, len(xs) - 1)
        rad = rng.randint(1, max(H // 100 - 2, 2))
        col = tuple(rng.randint(max(c - 50, 0), c) for c in src_color)
        geo.append((int(xs[r]), int(ys[r]), rad ** 3, *col))
        alpha.append(a)
    n_first = len(geo)
    num_times = src_radius // 10
    al = np.linspace(0.0, 1, num=num_times)
    rad = np.linspace(1, src_radius, num=num_times)
    for i in range(num_times):
        geo.append((cx, cy, int(rad[i]), *src_color))
        alpha.append(al[num_times - i - 1] ** 3)
    geo = np.clip(np.array(geo, np.int64), -2 ** 30, 2 ** 30).astype(np.int32).reshape(-1, 6)
    return geo, np.array(alpha, np.float64), n_first


class TrainTransform:
    """create_albu_transform(args, mean, std) for device-resident uint8 HWC images: `tf(img, rng) -> fp32 [C, S, S]`."""

    def __init__(self, args, mean, std, device, channels, seed=0):
        check(args)
        from types import SimpleNamespace

        # (keys a hand-built `args` may lack count as switched off, like an INI with every probability at zero)
        keys = dict(rotation=0.0, translate=0.0, scale=0.0, shear=0.0, albu_prob=0.0, individual_albu_probs=0.0,
                    noise_std=0.0, noise_prob=0.0, clahe=False, randomgamma=False, randombrightness=False, blur=False,
                    elastic=False, optical_distortion=False, grid_distortion=False, fog=False, grid_shuffle=False, hsv=False,
                    invert=False, cutout=False, shadow=False, sun_flare=False, solarize=False, equalize=False,
                    grid_dropout=False)
        self.cfg = SimpleNamespace(inference_resolution=args.inference_resolution, train_resolution=args.train_resolution,
                                   **{k: getattr(args, k, v) for k, v in keys.items()})
        self.device, self.C = torch.device(device), channels
        self.mean = None if mean is None else mean.to(device).float().reshape(-1).contiguous()
        self.std = None if std is None else std.to(device).float().reshape(-1).contiguous()
        self.gen = torch.Generator(device=self.device).manual_seed(seed)          # GaussNoise values
        S = args.train_resolution
        self.ws_bytes = query("primia_clahe_workspace_bytes", S, S, channels)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)
        # coordinate maps and the elastic workspace (float64 plane + two float32 planes), built once
        self.map_x = torch.empty(S, S, dtype=torch.float32, device=self.device)
        self.map_y = torch.empty(S, S, dtype=torch.float32, device=self.device)
        self.warp_ws = torch.empty(S * S * 16, dtype=torch.uint8, device=self.device)
        for key, name in (("shadow", "RandomShadows"), ("fog", "RandomFog"), ("sun_flare", "RandomSunFlare")):
            if getattr(self.cfg, key) and channels != 3:
                raise AssertionError(name + " needs 3 channels")          # torchlib/dataloader.py:184-191
        if self.cfg.hsv and channels != 3:
            raise AssertionError("HueSaturationValue needs 3 channels")   # (albumentations raises for grayscale input)
        self.eq_ws = torch.empty(3 * 256 * 5, dtype=torch.uint8, device=self.device)

    def _remap(self, cur):
        out = torch.empty_like(cur)
        S = cur.shape[0]
        call("primia_image_remap_u8", cur, S, S, self.C, self.map_x, self.map_y, out)
        return out

    def elastic(self, cur, seed, alpha=1.0, sigma=50.0, alpha_affine=50.0):
        """a.ElasticTransform().apply(img, random_state=seed) (F.elastic_transform, approximate=False)."""
        S, dev = cur.shape[0], self.device
        rs = np.random.RandomState(seed)
        center_square = np.float32((S, S)) // 2
        square_size = min((S, S)) // 3
        pts1 = np.float32([center_square + square_size, [center_square[0] + square_size, center_square[1] - square_size],
                           center_square - square_size])
        pts2 = pts1 + rs.uniform(-alpha_affine, alpha_affine, size=pts1.shape).astype(np.float32)
        inv = invert_affine(affine_from_points(pts1, pts2))
        call("primia_warp_map_affine", S, S, *[float(v) for v in inv.reshape(-1)], self.map_x, self.map_y)
        cur = self._remap(cur)
        fx = torch.from_numpy(rs.rand(S, S)).to(dev)
        fy = torch.from_numpy(rs.rand(S, S)).to(dev)
        call("primia_warp_map_elastic", S, S, fx, fy, float(sigma), float(alpha), self.warp_ws, self.warp_ws.numel(),
             self.map_x, self.map_y)
        return self._remap(cur)

    def optical(self, cur, k, dx, dy):
        """F.optical_distortion (albumentations 0.4.6: fx = fy = width)."""
        S = cur.shape[0]
        call("primia_warp_map_optical", S, S, float(np.float32(k)), float(S), float(S), S * 0.5 + dx, S * 0.5 + dy,
             (S - 1) * 0.5, (S - 1) * 0.5, self.map_x, self.map_y)
        return self._remap(cur)

    def grid(self, cur, xsteps, ysteps, num_steps=5):
        """F.grid_distortion: piecewise-linear axes (host, a few hundred values), meshgrid + remap on the device."""
        S, dev = cur.shape[0], self.device
        xx = torch.from_numpy(grid_axis(S, num_steps, xsteps)).to(dev)
        yy = torch.from_numpy(grid_axis(S, num_steps, ysteps)).to(dev)
        call("primia_warp_map_grid", S, S, xx, yy, self.map_x, self.map_y)
        return self._remap(cur)

    def fog(self, cur, fog_coef, haze_list, alpha_coef=0.08):
        """F.add_fog: haze discs blended in list order, then cv2.blur(hw // 10)."""
        S, dev = cur.shape[0], self.device
        hw = max(int(S // 3 * fog_coef), 10)
        hz = torch.tensor(haze_list, dtype=torch.int32).reshape(-1, 2).contiguous().to(dev)
        out = torch.empty_like(cur)
        call("primia_image_fog_u8", cur, S, S, self.C, hz if len(haze_list) else None, len(haze_list), hw,
             float(np.float32(alpha_coef * fog_coef)), out)
        k = hw // 10
        if k <= 1:
            return out
        blurred = torch.empty_like(out)
        call("primia_image_box_blur_u8", out, S, S, self.C, k, blurred)
        return blurred

    def _i32(self, rows):
        return torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(self.device)

    def grid_shuffle(self, cur, seed):
        S = cur.shape[0]
        tiles = grid_shuffle_tiles(S, S, seed)
        out = torch.empty_like(cur)
        call("primia_image_swap_tiles_u8", cur, S, S, self.C, self._i32(tiles), len(tiles), out)
        return out

    def hsv_shift(self, cur, hue_shift, sat_shift, val_shift):
        S = cur.shape[0]
        luts = torch.from_numpy(hsv_tables(hue_shift, sat_shift, val_shift)).to(self.device)
        out = torch.empty_like(cur)
        call("primia_image_hsv_shift_u8", cur, S, S, luts, out)
        return out

    def lut(self, cur, table):
        call("primia_image_lut_u8", cur, cur.numel(), torch.from_numpy(table).to(self.device), cur)
        return cur

    def fill_rects(self, cur, holes, fill=0):
        S = cur.shape[0]
        call("primia_image_fill_rects_u8", cur, S, S, self.C, self._i32(holes), len(holes), int(fill))
        return cur

    def shadow(self, cur, vertices):
        S = cur.shape[0]
        out = torch.empty_like(cur)
        call("primia_image_shadow_u8", cur, S, S, self._i32(vertices), vertices.shape[0], vertices.shape[1], out)
        return out

    def sun_flare(self, cur, geo, alpha, n_first):
        S = cur.shape[0]
        a32 = torch.from_numpy(alpha.astype(np.float32)).to(self.device)
        b32 = torch.from_numpy((1.0 - alpha).astype(np.float32)).to(self.device)
        out = torch.empty_like(cur)
        call("primia_image_sun_flare_u8", cur, S, S, self._i32(geo), a32, b32, len(geo), n_first, out)
        return out

    def equalize(self, cur):
        S = cur.shape[0]
        out = torch.empty_like(cur)
        call("primia_image_equalize_u8", cur, S, S, self.C, self.eq_ws, self.eq_ws.numel(), out)
        return out

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
            if a.elastic and rng.random() < p:                      # get_params: random.randint(0, 10000)
                cur = self.elastic(cur, rng.randint(0, 10000))
            if a.optical_distortion and rng.random() < p:           # distort_limit 0.05, shift_limit 0.05
                k = rng.uniform(-0.05, 0.05)
                dx, dy = round(rng.uniform(-0.05, 0.05)), round(rng.uniform(-0.05, 0.05))
                cur = self.optical(cur, k, dx, dy)
            if a.grid_distortion and rng.random() < p:              # num_steps 5, distort_limit 0.3
                xsteps = [1 + rng.uniform(-0.3, 0.3) for _ in range(6)]
                ysteps = [1 + rng.uniform(-0.3, 0.3) for _ in range(6)]
                cur = self.grid(cur, xsteps, ysteps)
            if a.grid_shuffle and rng.random() < p:                 # grid (3, 3); get_params: random.randint(0, 10000)
                cur = self.grid_shuffle(cur, rng.randint(0, 10000))
            if a.hsv and rng.random() < p:                          # hue 20, saturation 30, value 20
                hs, ss, vs = rng.uniform(-20, 20), rng.uniform(-30, 30), rng.uniform(-20, 20)
                cur = self.hsv_shift(cur, hs, ss, vs)
            if a.invert and rng.random() < p:
                cur = self.lut(cur, (255 - np.arange(256)).astype(np.uint8))
            if a.cutout and rng.random() < p:                       # num_holes 5, 80 x 80 (dataloader.py:178-182)
                cur = self.fill_rects(cur, cutout_holes(S, S, rng))
            if a.shadow and rng.random() < p:
                cur = self.shadow(cur, shadow_vertices(S, S, rng))
            if a.fog and rng.random() < p:                          # fog_coef (0.3, 1), alpha_coef 0.08
                fog_coef, haze = fog_params(S, S, rng)
                cur = self.fog(cur, fog_coef, haze)
            if a.sun_flare and rng.random() < p:
                cur = self.sun_flare(cur, *sun_flare_steps(S, S, rng))
            if a.solarize and rng.random() < p:                     # threshold (128, 128): the draw is still made
                cur = self.lut(cur, solarize_table(rng.uniform(128, 128)))
            if a.equalize and rng.random() < p:
                cur = self.equalize(cur)
            if a.grid_dropout and rng.random() < p:
                cur = self.fill_rects(cur, grid_dropout_holes(S, S))
            if rng.random() < a.noise_prob:                                            # a.GaussNoise(var_limit = noise_std^2)
                var = rng.uniform(0.0, a.noise_std ** 2)
                noise = torch.randn(cur.numel(), generator=self.gen, device=dev) * (var ** 0.5)
                call("primia_image_add_noise_u8", cur, noise, cur.numel(), cur)
        out = torch.empty(C, S, S, dtype=torch.float32, device=dev)
        call("primia_image_finish", cur, S, C, self.mean, self.std, out)
        return out


def create_albu_transform(args, mean, std, device="cuda:0", channels=None, seed=0):
    """torchlib/dataloader.py:138-217 by its own name: the training transform chain for `args`, as a TrainTransform
    (`tf(uint8 HWC device image, random.Random) -> fp32 [C, S, S]`).  `channels` defaults to 3 for `pretrained` presets and
    1 otherwise, as the reference's end_transformations do."""
    if channels is None:
        channels = 3 if getattr(args, "pretrained", True) else 1
    return TrainTransform(args, mean, std, device, channels, seed)

