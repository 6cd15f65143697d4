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


# ======================================================================================================================
# The warping transforms and RandomFog of albumentations 0.4.6 (the version the reference pins:
# /root/reference/environment_torch.yml:135), switched on by the reference's shipped preset
# (/root/reference/configs/torch/pneumonia-resnet-pretrained.ini:44-46,52; torchlib/dataloader.py:167-172,188-189).
# albumentations and cv2 are absent from the reference tree and from this image: the functions restate
# albumentations/augmentations/functional.py (elastic_transform, optical_distortion, grid_distortion, add_fog) and
# transforms.py (the get_params draws) of that release, with cv2.remap / cv2.warpAffine as float bilinear interpolation
# under BORDER_REFLECT_101 (cv2's 8-bit path interpolates with 5-bit fixed-point coordinates: at most one grey level off),
# cv2.initUndistortRectifyMap from its documented pinhole model, cv2.circle as the disc (dx^2 + dy^2 <= r^2),
# cv2.addWeighted / cv2.blur with round-half-to-even.  PARITY UNPINNED against the binaries, like the rest of this file.
# The random streams ARE the published ones: Python's `random` for the parameter draws, numpy.random.RandomState(seed)
# for ElasticTransform's fields (legacy MT19937 stream, stable across NumPy versions), scipy.ndimage.gaussian_filter.
# ======================================================================================================================
def remap_bilinear(img, map_x, map_y):
    """cv2.remap(img, map_x, map_y, INTER_LINEAR, BORDER_REFLECT_101) in float32 arithmetic, rounded to nearest even."""
    H, W = img.shape[:2]
    f32 = np.float32
    mx, my = np.asarray(map_x, f32), np.asarray(map_y, f32)
    x0f, y0f = np.floor(mx), np.floor(my)
    fx, fy = (mx - x0f).astype(f32), (my - y0f).astype(f32)
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    xa, xb = reflect101(x0, W), reflect101(x0 + 1, W)
    ya, yb = reflect101(y0, H), reflect101(y0 + 1, H)
    p = img.astype(f32)
    if img.ndim == 3:
        fx, fy = fx[..., None], fy[..., None]
    top = p[ya, xa] * (f32(1) - fx) + p[ya, xb] * fx
    bot = p[yb, xa] * (f32(1) - fx) + p[yb, xb] * fx
    v = top * (f32(1) - fy) + bot * fy
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def affine_from_points(pts1, pts2):
    """cv2.getAffineTransform: the 2 x 3 matrix M (float64) with M [x, y, 1]^T = pts2 for the three point pairs."""
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        x, y = float(pts1[i][0]), float(pts1[i][1])
        a[2 * i] = [x, y, 1, 0, 0, 0]
        a[2 * i + 1] = [0, 0, 0, x, y, 1]
        b[2 * i], b[2 * i + 1] = float(pts2[i][0]), float(pts2[i][1])
    return np.linalg.solve(a, b).reshape(2, 3)


def invert_affine(m):
    """cv2.invertAffineTransform (what warpAffine applies without WARP_INVERSE_MAP), float64."""
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    a12, a21 = -m[0, 1] * d, -m[1, 0] * d
    b1 = -a11 * m[0, 2] - a12 * m[1, 2]
    b2 = -a21 * m[0, 2] - a22 * m[1, 2]
    return np.array([[a11, a12, b1], [a21, a22, b2]], np.float64)


def affine_maps(H, W, inv):
    """dst pixel (x, y) -> source coordinates under the inverse matrix (float64 arithmetic, float32 maps)."""
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    return ((inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]).astype(np.float32),
            (inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]).astype(np.float32))


def elastic_params(H, W, seed, alpha=1.0, sigma=50.0, alpha_affine=50.0):
    """F.elastic_transform's random part: (inverse affine matrix, dx, dy) from np.random.RandomState(seed)."""
    from scipy.ndimage import gaussian_filter

    rs = np.random.RandomState(seed)
    center_square = np.float32((H, W)) // 2
    square_size = min((H, W)) // 3
    pts1 = np.float32([center_square + square_size, [center_square[0] + square_size, center_square[1] - square_size],
                       center_square - square_size])
    pts2 = pts1 + rs.uniform(-alpha_affine, alpha_affine, size=pts1.shape).astype(np.float32)
    inv = invert_affine(affine_from_points(pts1, pts2))
    r1, r2 = rs.rand(H, W), rs.rand(H, W)
    dx = np.float32(gaussian_filter(r1 * 2 - 1, sigma) * alpha)
    dy = np.float32(gaussian_filter(r2 * 2 - 1, sigma) * alpha)
    return inv, dx, dy, (r1, r2)


def elastic_transform(img, seed, alpha=1.0, sigma=50.0, alpha_affine=50.0):
    """a.ElasticTransform().apply: warpAffine(random affine) then remap(x + dx, y + dy), both bilinear / REFLECT_101."""
    H, W = img.shape[:2]
    inv, dx, dy, _ = elastic_params(H, W, seed, alpha, sigma, alpha_affine)
    img = remap_bilinear(img, *affine_maps(H, W, inv))
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    return remap_bilinear(img, np.float32(xs + dx), np.float32(ys + dy))


def optical_maps(H, W, k, dx, dy):
    """cv2.initUndistortRectifyMap(camera, (k, k, 0, 0, 0), None, None, (W, H), CV_32FC1) for albumentations 0.4.6's
    camera matrix fx = fy = width, principal point (W / 2 + dx, H / 2 + dy); the new camera matrix defaults to the same
    focal lengths with the principal point at the image centre ((W - 1) / 2, (H - 1) / 2)."""
    fx = fy = float(W)
    cx, cy = W * 0.5 + dx, H * 0.5 + dy
    ncx, ncy = (W - 1) * 0.5, (H - 1) * 0.5
    k = float(np.float32(k))
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    x, y = (xs - ncx) / fx, (ys - ncy) / fy
    r2 = x * x + y * y
    kr = 1 + k * r2 + k * r2 * r2
    return (fx * (x * kr) + cx).astype(np.float32), (fy * (y * kr) + cy).astype(np.float32)


def optical_distortion(img, k, dx, dy):
    H, W = img.shape[:2]
    return remap_bilinear(img, *optical_maps(H, W, k, dx, dy))


def grid_axis(n, num_steps, steps):
    """One axis of F.grid_distortion: piecewise-linear positions for n pixels."""
    step = n // num_steps
    xx = np.zeros(n, np.float32)
    prev = 0
    for idx, x in enumerate(range(0, n, step)):
        start, end = x, x + step
        if end > n:
            end = n
            cur = n
        else:
            cur = prev + step * steps[idx]
        xx[start:end] = np.linspace(prev, cur, end - start)
        prev = cur
    return xx


def grid_distortion(img, xsteps, ysteps, num_steps=5):
    H, W = img.shape[:2]
    map_x, map_y = np.meshgrid(grid_axis(W, num_steps, xsteps), grid_axis(H, num_steps, ysteps))
    return remap_bilinear(img, map_x.astype(np.float32), map_y.astype(np.float32))


def fog_params(H, W, rng, fog_coef_lower=0.3, fog_coef_upper=1.0):
    """RandomFog.get_params_dependent_on_targets: (fog_coef, haze_list) from Python's `random` stream."""
    fog_coef = rng.uniform(fog_coef_lower, fog_coef_upper)
    hw = max(1, int(W // 3 * fog_coef))
    haze = []
    midx, midy = W // 2 - 2 * hw, H // 2 - hw
    index = 1
    while midx > -hw or midy > -hw:
        for _ in range(hw // 10 * index):
            x = rng.randint(midx, W - midx - hw)
            y = rng.randint(midy, H - midy - hw)
            haze.append((x, y))
        midx -= 3 * hw * W // (H + W)
        midy -= 3 * hw * H // (H + W)
        index += 1
    return fog_coef, haze


def add_fog(img, fog_coef, haze_list, alpha_coef=0.08):
    """F.add_fog: per haze point a white disc blended in with weight alpha_coef * fog_coef, then cv2.blur(hw // 10)."""
    H, W = img.shape[:2]
    hw = max(int(W // 3 * fog_coef), 10)
    f32 = np.float32
    alpha = f32(alpha_coef * fog_coef)
    beta = f32(1 - alpha_coef * fog_coef)
    rad = hw // 2
    ys, xs = np.mgrid[0:H, 0:W]
    out = img.copy()
    for x, y in haze_list:
        inside = (xs - (x + hw // 2)) ** 2 + (ys - (y + hw // 2)) ** 2 <= rad * rad
        v = np.rint(f32(255) * alpha + out.astype(f32) * beta)
        out = np.where(inside[..., None] if img.ndim == 3 else inside, np.clip(v, 0, 255).astype(np.uint8), out)
    return box_blur_anchor(out, hw // 10)


def box_blur_anchor(img, k):
    """cv2.blur with any kernel size: window [p - k // 2, p - k // 2 + k - 1] (anchor = k // 2), REFLECT_101."""
    if k <= 1:
        return img.copy()
    H, W = img.shape[:2]
    a = k // 2
    ys, xs = reflect101(np.arange(-a, H - a + k - 1), H), reflect101(np.arange(-a, W - a + k - 1), W)
    ext = img[ys][:, xs].astype(np.int64)
    s = np.zeros(img.shape, np.int64)
    for dy in range(k):
        for dx in range(k):
            s += ext[dy:dy + H, dx:dx + W]
    return np.clip(np.rint(s.astype(np.float32) / np.float32(k * k)), 0, 255).astype(np.uint8)


# ---- round 4, second batch: the nine remaining members of create_albu_transform (dataloader.py:173-201) --------------
# RandomGridShuffle, HueSaturationValue, InvertImg, Cutout, RandomShadow, RandomSunFlare, Solarize, Equalize, GridDropout
# (albumentations 0.4.6 augmentations/transforms.py + functional.py; OpenCV color_hsv.cpp, histogram.cpp).  Restated from
# the published sources; where a raster rule of cv2's drawing functions is approximated it is said at the function.

def invert_table():
    """F.invert: 255 - img."""
    return (255 - np.arange(256)).astype(np.uint8)


def solarize_table(threshold=128):
    """F.solarize (uint8): lut[i] = i if i < threshold else 255 - i."""
    i = np.arange(256)
    return np.where(i < threshold, i, 255 - i).astype(np.uint8)


def equalize_table(hist):
    """cv2.equalizeHist's table from a 256-bin histogram (imgproc/src/histogram.cpp, EqualizeHistLut_Invoker)."""
    hist = np.asarray(hist, np.int64)
    total = int(hist.sum())
    i0 = int(np.nonzero(hist)[0][0])
    lut = np.zeros(256, np.uint8)
    if hist[i0] == total:
        lut[:] = i0
        return lut
    scale = np.float32(255.0) / np.float32(total - hist[i0])
    s = np.cumsum(hist[i0 + 1:]).astype(np.float32) * scale
    lut[i0 + 1:] = np.clip(np.rint(s), 0, 255).astype(np.uint8)
    return lut


def equalize(img):
    """F.equalize(mode='cv', by_channels=True): cv2.equalizeHist per channel."""
    out = np.empty_like(img)
    planes = img.reshape(img.shape[0], img.shape[1], -1)
    o = out.reshape(planes.shape)
    for c in range(planes.shape[2]):
        o[..., c] = equalize_table(np.bincount(planes[..., c].reshape(-1), minlength=256))[planes[..., c]]
    return out


def cutout_holes(H, W, rng, num_holes=5, max_h_size=80, max_w_size=80):
    """a.Cutout.get_params_dependent_on_targets: (x1, y1, x2, y2) per hole, Python's `random` stream."""
    holes = []
    for _ in range(num_holes):
        y, x = rng.randint(0, H), rng.randint(0, W)
        y1 = int(np.clip(y - max_h_size // 2, 0, H))
        y2 = int(np.clip(y1 + max_h_size, 0, H))
        x1 = int(np.clip(x - max_w_size // 2, 0, W))
        x2 = int(np.clip(x1 + max_w_size, 0, W))
        holes.append((x1, y1, x2, y2))
    return holes


def grid_dropout_holes(H, W, ratio=0.5):
    """a.GridDropout defaults (no unit size limits, no hole counts, shift 0, no random offset): the hole rectangles."""
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


def fill_rects(img, holes, fill=0):
    """F.cutout: img[y1:y2, x1:x2] = fill_value."""
    out = img.copy()
    for x1, y1, x2, y2 in holes:
        out[y1:y2, x1:x2] = fill
    return out


def grid_shuffle_tiles(H, W, seed, grid=(3, 3)):
    """a.RandomGridShuffle.get_params_dependent_on_targets: rows of (y, x, old_y, old_x, height, width)."""
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


def swap_tiles(img, tiles):
    """F.swap_tiles_on_image."""
    out = img.copy()
    for y, x, oy, ox, h, w in tiles:
        out[y:y + h, x:x + w] = img[oy:oy + h, ox:ox + w]
    return out


_HSV_SHIFT = 12
_SDIV = np.zeros(256, np.int64)
_HDIV180 = np.zeros(256, np.int64)
_SDIV[1:] = np.rint((255 << _HSV_SHIFT) / (1.0 * np.arange(1, 256))).astype(np.int64)
_HDIV180[1:] = np.rint((180 << _HSV_SHIFT) / (6.0 * np.arange(1, 256))).astype(np.int64)


def rgb2hsv_u8(img):
    """cv2.cvtColor(COLOR_RGB2HSV) on uint8 (color_hsv.cpp RGB2HSV_b, hrange 180): 12-bit fixed-point division tables."""
    r, g, b = [img[..., i].astype(np.int64) for i in range(3)]
    v = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = v - vmin
    s = (diff * _SDIV[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV180[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def _sector_rgb(tab, sector):
    """(b, g, r) = tab[sector_data[sector][0..2]] of OpenCV's HSV2RGB_f / HLS2RGB_f; returns (r, g, b)."""
    pick = lambda k: np.take_along_axis(tab, _SECTOR[sector][..., k][..., None], axis=-1)[..., 0]
    return pick(2), pick(1), pick(0)


def hsv2rgb_u8(hsv):
    """cv2.cvtColor(COLOR_HSV2RGB) on uint8: HSV2RGB_f in float32 on (h, s / 255, v / 255), x 255, round, saturate."""
    f = np.float32
    h = hsv[..., 0].astype(f) * f(6.0 / 180.0)
    s = hsv[..., 1].astype(f) * f(1.0 / 255.0)
    v = hsv[..., 2].astype(f) * f(1.0 / 255.0)
    h = np.where(h >= 6, h - f(6), h)
    sector = np.floor(h).astype(np.int64)
    hf = h - sector.astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector, hf = np.where(bad, 0, sector), np.where(bad, f(0), hf)
    tab = np.stack([v, v * (f(1) - s), v * (f(1) - s * hf), v * (f(1) - s * (f(1) - hf))], axis=-1).astype(f)
    r, g, b = _sector_rgb(tab, sector)
    grey = s == 0
    out = [np.where(grey, v, c) for c in (r, g, b)]
    return np.stack([np.clip(np.rint(c * f(255)), 0, 255) for c in out], axis=-1).astype(np.uint8)


def shift_hsv(img, hue_shift, sat_shift, val_shift):
    """F._shift_hsv_uint8: RGB -> HSV, three cv2.LUT tables, HSV -> RGB."""
    hsv = rgb2hsv_u8(img)
    i = np.arange(256, dtype=np.int16)
    lut_h = np.mod(i + hue_shift, 180).astype(np.uint8)
    lut_s = np.clip(i + sat_shift, 0, 255).astype(np.uint8)
    lut_v = np.clip(i + val_shift, 0, 255).astype(np.uint8)
    return hsv2rgb_u8(np.stack([lut_h[hsv[..., 0]], lut_s[hsv[..., 1]], lut_v[hsv[..., 2]]], axis=-1)), (lut_h, lut_s, lut_v)


def rgb2hls_u8(img):
    """cv2.cvtColor(COLOR_RGB2HLS) on uint8: RGB2HLS_f in float32 on x / 255, then (h / 2, l * 255, s * 255) rounded."""
    f = np.float32
    r, g, b = [img[..., i].astype(f) * f(1.0 / 255.0) for i in range(3)]
    vmax = np.maximum(np.maximum(r, g), b)
    vmin = np.minimum(np.minimum(r, g), b)
    diff = (vmax - vmin).astype(f)
    l = ((vmax + vmin) * f(0.5)).astype(f)
    has = diff > np.finfo(f).eps
    safe = np.where(has, diff, f(1))
    s = np.where(l < f(0.5), diff / np.where(has, vmax + vmin, f(1)), diff / np.where(has, f(2) - vmax - vmin, f(1))).astype(f)
    d60 = (f(60.0) / safe).astype(f)
    h = np.where(vmax == r, (g - b) * d60, np.where(vmax == g, (b - r) * d60 + f(120), (r - g) * d60 + f(240))).astype(f)
    h = np.where(h < 0, h + f(360), h)
    h, s = np.where(has, h, f(0)), np.where(has, s, f(0))
    cv = lambda x: np.clip(np.rint(x), 0, 255)
    return np.stack([cv(h * f(0.5)), cv(l * f(255)), cv(s * f(255))], axis=-1).astype(np.uint8)


def hls2rgb_u8(hls):
    """cv2.cvtColor(COLOR_HLS2RGB) on uint8: HLS2RGB_f in float32 on (h, l / 255, s / 255)."""
    f = np.float32
    h = hls[..., 0].astype(f) * f(6.0 / 180.0)
    l = hls[..., 1].astype(f) * f(1.0 / 255.0)
    s = hls[..., 2].astype(f) * f(1.0 / 255.0)
    p2 = np.where(l <= f(0.5), l * (f(1) + s), l + s - l * s).astype(f)
    p1 = (f(2) * l - p2).astype(f)
    h = np.where(h >= 6, h - f(6), h)
    sector = np.floor(h).astype(np.int64)
    hf = h - sector.astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector, hf = np.where(bad, 0, sector), np.where(bad, f(0), hf)
    tab = np.stack([p2, p1, p1 + (p2 - p1) * (f(1) - hf), p1 + (p2 - p1) * hf], axis=-1).astype(f)
    r, g, b = _sector_rgb(tab, sector)
    grey = s == 0
    out = [np.where(grey, l, c) for c in (r, g, b)]
    return np.stack([np.clip(np.rint(c * f(255)), 0, 255) for c in out], axis=-1).astype(np.uint8)


def shadow_vertices(H, W, rng, shadow_roi=(0, 0.5, 1, 1), lower=1, upper=2, dimension=5):
    """a.RandomShadow.get_params_dependent_on_targets: [num_shadows][dimension][2] (x, y) vertices."""
    n = rng.randint(lower, upper)
    x_min, y_min, x_max, y_max = shadow_roi
    x_min, x_max, y_min, y_max = int(x_min * W), int(x_max * W), int(y_min * H), int(y_max * H)
    return np.array([[(rng.randint(x_min, x_max), rng.randint(y_min, y_max)) for _ in range(dimension)] for _ in range(n)],
                    np.int32)


def polygon_mask(H, W, verts):
    """The pixels cv2.fillPoly(mask, [verts], 255) sets, restated as: even-odd interior test at integer pixel coordinates
    (an edge counts for rows y0 <= y < y1) OR on one of the polygon's edges drawn as a DDA line (minor coordinate rounded
    half up).  cv2's fixed-point scanline may differ from this rule on individual boundary pixels."""
    ys, xs = np.mgrid[0:H, 0:W]
    inside = np.zeros((H, W), bool)
    edge = np.zeros((H, W), bool)
    n = len(verts)
    for k in range(n):
        x0, y0 = [int(v) for v in verts[k]]
        x1, y1 = [int(v) for v in verts[(k + 1) % n]]
        if y0 != y1:
            dy = y1 - y0
            strad = (y0 <= ys) != (y1 <= ys)
            lhs, rhs = xs * dy, x0 * dy + (ys - y0) * (x1 - x0)
            inside ^= strad & (lhs < rhs if dy > 0 else lhs > rhs)
        adx, ady = abs(x1 - x0), abs(y1 - y0)
        if adx >= ady:
            (xa, ya), (xb, yb) = ((x0, y0), (x1, y1)) if x0 <= x1 else ((x1, y1), (x0, y0))
            d = xb - xa
            if d == 0:
                edge |= (xs == xa) & (ys == ya)
            else:
                yy = ya + np.floor_divide(2 * (xs - xa) * (yb - ya) + d, 2 * d)
                edge |= (xs >= xa) & (xs <= xb) & (ys == yy)
        else:
            (xa, ya), (xb, yb) = ((x0, y0), (x1, y1)) if y0 <= y1 else ((x1, y1), (x0, y0))
            d = yb - ya
            xx = xa + np.floor_divide(2 * (ys - ya) * (xb - xa) + d, 2 * d)
            edge |= (ys >= ya) & (ys <= yb) & (xs == xx)
    return inside | edge


def add_shadow(img, vertices_list):
    """F.add_shadow: RGB -> HLS, L *= 0.5 (stored back into uint8: truncation) under the union of the polygons, HLS -> RGB."""
    H, W = img.shape[:2]
    hls = rgb2hls_u8(img)
    mask = np.zeros((H, W), bool)
    for verts in vertices_list:
        mask |= polygon_mask(H, W, verts)
    hls[..., 1] = np.where(mask, hls[..., 1] >> 1, hls[..., 1])
    return hls2rgb_u8(hls)


def sun_flare_params(H, W, rng, flare_roi=(0, 0, 1, 0.5), angle_lower=0.0, angle_upper=1.0, circles_lower=6,
                     circles_upper=10, src_radius=400, src_color=(255, 255, 255)):
    """a.RandomSunFlare.get_params_dependent_on_targets: the small circles, the flare centre."""
    import math

    angle = 2 * math.pi * rng.uniform(angle_lower, angle_upper)
    lx, ly, ux, uy = flare_roi
    cx, cy = rng.uniform(lx, ux), rng.uniform(ly, uy)
    cx, cy = int(W * cx), int(H * cy)
    num = rng.randint(circles_lower, circles_upper)
    xs, ys = [], []
    for rx in range(0, W, 10):
        ry = math.tan(angle) * (rx - cx) + cy
        xs.append(rx)
        ys.append(2 * cy - ry)
    circles = []
    for _ in range(num):
        alpha = rng.uniform(0.05, 0.2)
        r = rng.randint(0, len(xs) - 1)
        rad = rng.randint(1, max(H // 100 - 2, 2))
        col = tuple(rng.randint(max(c - 50, 0), c) for c in src_color)
        circles.append((alpha, (int(xs[r]), int(ys[r])), rad ** 3, col))
    return cx, cy, circles


def sun_flare_steps(cx, cy, circles, src_radius=400, src_color=(255, 255, 255)):
    """F.add_sun_flare as a flat list of (x, y, radius, r, g, b), weights, and the index where `overlay = output.copy()`."""
    geo = [(x, y, rad, *col) for _, (x, y), rad, col in circles]
    alpha = [a for a, *_ in circles]
    n_first = len(geo)
    num_times = src_radius // 10
    al = np.linspace(0.0, 1, num=num_times)
    rad = np.linspace(1, src_radius, num=num_times)
    for i in range(num_times):
        geo.append((cx, cy, int(rad[i]), *src_color))
        alpha.append(al[num_times - i - 1] ** 3)
    return np.array(geo, np.int32).reshape(-1, 6), np.array(alpha, np.float64), n_first


def add_sun_flare(img, cx, cy, circles, src_radius=400, src_color=(255, 255, 255)):
    """F.add_sun_flare: cv2.circle(filled) onto `overlay`, cv2.addWeighted(overlay, a, output, 1 - a, 0, output) after
    every circle.  A filled circle is restated as the disc dx^2 + dy^2 <= r^2 (cv2's midpoint raster differs on
    individual rim pixels); addWeighted in float32, rounded half to even, saturated."""
    geo, alpha, n_first = sun_flare_steps(cx, cy, circles, src_radius, src_color)
    f = np.float32
    H, W = img.shape[:2]
    ys, xs = np.mgrid[0:H, 0:W]
    overlay, output = img.astype(f), img.astype(f)
    for k in range(len(geo)):
        if k == n_first:
            overlay = output.copy()
        x, y, r = [int(v) for v in geo[k, :3]]
        inside = (xs - x).astype(np.int64) ** 2 + (ys - y).astype(np.int64) ** 2 <= r * r
        overlay = np.where(inside[..., None], geo[k, 3:6].astype(f), overlay)
        a, b = f(alpha[k]), f(1.0 - alpha[k])
        output = np.clip(np.rint(overlay * a + output * b), 0, 255).astype(f)
    return output.astype(np.uint8)
