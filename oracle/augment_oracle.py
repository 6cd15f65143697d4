"""ORACLE (test infrastructure, not product code) — NumPy restatement of the image arithmetic behind the augmentation
chain of create_albu_transform (/root/reference/torchlib/dataloader.py:138-217).

Only tests/ may import this file.

PARITY UNPINNED: the arithmetic lives in third-party binaries that are not in /root/reference and not in this image —
OpenCV (CLAHE, cv2.blur, cv2.LUT, cv2.cvtColor), albumentations 0.4.x (the LUT tables, parameter draws), Pillow
(Image.transform for torchvision's RandomAffine), none of them pinned to a version by the reference's environment file.
The functions below follow the published algorithms (OpenCV modules/imgproc/src/clahe.cpp; box filter with
BORDER_REFLECT_101 and round-to-nearest; PIL's nearest-neighbour affine sampling at pixel centres; the CIE L*a*b*
definition with D65 white and the sRGB transfer function, where cv2's 8-bit path uses fixed-point tables) and serve as
the bit-exact reference for the HIP kernels of csrc/augment.hip; agreement with cv2's own output is NOT established.
"""
import numpy as np


def reflect101(p, n):
    p = np.asarray(p)
    if n == 1:
        return np.zeros_like(p)
    period = 2 * (n - 1)
    p = np.mod(p, period)
    return np.where(p >= n, period - p, p)


def affine_nearest(img, m):
    """PIL Image.transform(size, AFFINE, m, NEAREST): out[y, x] = img[floor(d(x+.5)+e(y+.5)+f), floor(a(x+.5)+b(y+.5)+c)]."""
    H, W = img.shape[:2]
    a, b, c, d, e, f = [np.float32(v).astype(np.float64) for v in m]
    ys, xs = np.mgrid[0:H, 0:W]
    xi = np.floor(a * (xs + 0.5) + b * (ys + 0.5) + c).astype(np.int64)
    yi = np.floor(d * (xs + 0.5) + e * (ys + 0.5) + f).astype(np.int64)
    ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
    out = np.zeros_like(img)
    out[ok] = img[yi[ok], xi[ok]]
    return out


def resize_crop(img, R, oy, ox, S, flip_v=False):
    """a.Resize(R, R) (bilinear, half-pixel centres, clamped, rounded to uint8) -> crop (oy, ox, S) [-> vertical flip]."""
    Hin, Win = img.shape[:2]
    f32 = np.float32
    y = np.arange(S)
    ry = (S - 1 - y if flip_v else y) + oy
    rx = np.arange(S) + ox
    sy = (ry.astype(f32) + f32(0.5)) * (f32(Hin) / f32(R)) - f32(0.5)
    sx = (rx.astype(f32) + f32(0.5)) * (f32(Win) / f32(R)) - f32(0.5)

    def split(s, n):
        i0 = np.floor(s).astype(np.int64)
        fr = (s - i0.astype(f32)).astype(f32)
        fr = np.where(i0 < 0, f32(0), fr)
        i0 = np.maximum(i0, 0)
        i1 = i0 + 1
        over = i1 >= n
        fr = np.where(over & (i0 >= n - 1), f32(0), fr)
        i0 = np.where(over & (i0 >= n - 1), n - 1, i0)
        i1 = np.minimum(i1, n - 1)
        return i0, i1, fr.astype(f32)

    y0, y1, fy = split(sy, Hin)
    x0, x1, fx = split(sx, Win)
    p = img.astype(f32)
    p00, p01 = p[y0][:, x0], p[y0][:, x1]
    p10, p11 = p[y1][:, x0], p[y1][:, x1]
    fxb = fx[None, :, None] if img.ndim == 3 else fx[None, :]
    fyb = fy[:, None, None] if img.ndim == 3 else fy[:, None]
    top = p00 + (p01 - p00) * fxb
    bot = p10 + (p11 - p10) * fxb
    v = np.floor(top + (bot - top) * fyb + f32(0.5))
    return np.clip(v, 0, 255).astype(np.uint8)


def clahe_plane(pl, clip_limit, tiles=8):
    """OpenCV CLAHE on one uint8 plane (clahe.cpp): tile histograms on the reflect-101 padded image, clip +
    redistribute, cumulative LUT, bilinear blend of the four neighbouring tiles' LUTs."""
    H, W = pl.shape
    tw, th = -(-W // tiles), -(-H // tiles)
    clip = 0
    if clip_limit > 0:
        clip = max(int(np.float32(clip_limit) * np.float32(tw * th) / np.float32(256.0)), 1)
    ys = reflect101(np.arange(th * tiles), H)
    xs = reflect101(np.arange(tw * tiles), W)
    ext = pl[ys][:, xs]
    lut = np.zeros((tiles, tiles, 256), np.uint8)
    scale = np.float32(255.0) / np.float32(tw * th)
    for ty in range(tiles):
        for tx in range(tiles):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].reshape(-1), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch = clipped // 256
                residual = clipped - batch * 256
                hist = hist + batch
                if residual:
                    step = max(256 // residual, 1)
                    i = 0
                    while i < 256 and residual > 0:
                        hist[i] += 1
                        i += step
                        residual -= 1
            s = np.cumsum(hist).astype(np.float32) * scale
            lut[ty, tx] = np.clip(np.rint(s), 0, 255).astype(np.uint8)
    f32 = np.float32
    txf = np.arange(W).astype(f32) * (f32(1.0) / f32(tw)) - f32(0.5)
    tyf = np.arange(H).astype(f32) * (f32(1.0) / f32(th)) - f32(0.5)
    tx1, ty1 = np.floor(txf).astype(np.int64), np.floor(tyf).astype(np.int64)
    xa, ya = (txf - tx1.astype(f32)).astype(f32), (tyf - ty1.astype(f32)).astype(f32)
    tx2, ty2 = np.minimum(tx1 + 1, tiles - 1), np.minimum(ty1 + 1, tiles - 1)
    tx1, ty1 = np.maximum(tx1, 0), np.maximum(ty1, 0)
    v = pl.astype(np.int64)
    l11 = lut[ty1[:, None], tx1[None, :], v].astype(f32)
    l12 = lut[ty1[:, None], tx2[None, :], v].astype(f32)
    l21 = lut[ty2[:, None], tx1[None, :], v].astype(f32)
    l22 = lut[ty2[:, None], tx2[None, :], v].astype(f32)
    xa, ya = xa[None, :], ya[:, None]
    res = (l11 * (f32(1) - xa) + l12 * xa) * (f32(1) - ya) + (l21 * (f32(1) - xa) + l22 * xa) * ya
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def gamma_table(gamma):
    """albumentations.gamma_transform for uint8: cv2.LUT with (arange(256) / 255) ** gamma * 255, cast to uint8."""
    return (np.power(np.arange(0, 256.0 / 255, 1.0 / 255)[:256], gamma) * 255).astype(np.uint8)


def brightness_table(alpha, beta):
    """albumentations._brightness_contrast_adjust_uint (beta_by_max): clip(arange(256) * alpha + beta * 255)."""
    lut = np.arange(0, 256, dtype=np.float32)
    if alpha != 1:
        lut *= np.float32(alpha)
    if beta != 0:
        lut += np.float32(beta * 255.0)
    return np.clip(lut, 0, 255).astype(np.uint8)


def box_blur(img, k):
    """cv2.blur(img, (k, k)): mean of the k x k window, BORDER_REFLECT_101, round half to even like cvRound."""
    H, W = img.shape[:2]
    r = k // 2
    ys, xs = reflect101(np.arange(-r, H + r), H), reflect101(np.arange(-r, W + r), W)
    ext = img[ys][:, xs].astype(np.int64)
    s = np.zeros(img.shape, np.int64)
    for dy in range(k):
        for dx in range(k):
            s += ext[dy:dy + H, dx:dx + W]
    return np.clip(np.rint(s.astype(np.float32) / np.float32(k * k)), 0, 255).astype(np.uint8)


def add_noise(img, noise):
    """albumentations.gauss_noise on uint8 (@clipped): (img.astype(float32) + gauss) clipped to [0, 255], astype(uint8)."""
    return np.clip(img.astype(np.float32) + noise.astype(np.float32), 0, 255).astype(np.uint8)


def finish(img, mean=None, std=None):
    """a.ToFloat(255) -> a.Normalize(mean, std, max_pixel_value=1.0); HWC uint8 -> CHW float32."""
    v = img.astype(np.float32) / np.float32(255.0)
    if v.ndim == 2:
        v = v[:, :, None]
    if mean is not None:
        v = (v - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(v.transpose(2, 0, 1)).astype(np.float32)


def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision 0.5 transforms.functional._get_inverse_affine_matrix (single shear angle, degrees)."""
    import math

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
