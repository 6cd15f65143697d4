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
