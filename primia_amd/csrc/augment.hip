// The augmentation chain of create_albu_transform (torchlib/dataloader.py:138-217) on the GPU, for the members the
// reference's shipped presets switch on that are plain image arithmetic: RandomAffine (torchvision, nearest
// neighbour), Resize + RandomCrop (to uint8), CLAHE (always applied when `clahe = yes`), VerticalFlip, RandomGamma,
// RandomBrightness, Blur, GaussNoise, then ToFloat + Normalize.  Each is one launch on a decoded uint8 HWC image that
// already lives in device memory; the random PARAMETERS are drawn by the host (primia_amd/augment.py) the way
// torchvision / albumentations draw them, the kernels are deterministic functions of (image, parameters).
//
// cv2 / albumentations are not in this image, so these kernels follow the published algorithms (OpenCV's clahe.cpp,
// box filter with BORDER_REFLECT_101, cv2.LUT tables as albumentations builds them) and are held bit-exact to
// oracle/augment_oracle.py, which restates them in NumPy; parity with cv2's own binaries is unpinned.
// Registration-time work, a few hundred kilobytes per image: nowhere near a roofline, written for clarity.
#include "common.h"

// every product below is rounded on its own, as the NumPy / OpenCV float arithmetic these kernels follow does: no
// fused multiply-add contraction in this file
#pragma clang fp contract(off)

namespace primia {

// ---- RandomAffine: PIL Image.transform(AFFINE, NEAREST): source index = floor(a (x + .5) + b (y + .5) + c) ----------
__global__ __launch_bounds__(256) void affine_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int C, float a,
                                                        float b, float c, float d, float e, float f,
                                                        uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const double xs = (double)a * (x + 0.5) + (double)b * (y + 0.5) + (double)c;
    const double ys = (double)d * (x + 0.5) + (double)e * (y + 0.5) + (double)f;
    const int xi = (int)floor(xs), yi = (int)floor(ys);
    const bool in = xi >= 0 && xi < W && yi >= 0 && yi < H;
    for (int ch = 0; ch < C; ++ch) out[(long)idx * C + ch] = in ? src[((long)yi * W + xi) * C + ch] : (uint8_t)0;
}

// ---- Resize(R, R) + crop (oy, ox, S) [+ vertical flip] to uint8 HWC: image_prepare_kernel's sampling ---------------
__global__ __launch_bounds__(256) void resize_crop_u8_kernel(const uint8_t* __restrict__ src, int Hin, int Win, int C,
                                                             int R, int oy, int ox, int S, int flip_v,
                                                             uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * S) return;
    const int y = idx / S, x = idx - y * S;
    const int ry = (flip_v ? S - 1 - y : y) + oy, rx = x + ox;
    const float sy = ((float)ry + 0.5f) * ((float)Hin / (float)R) - 0.5f;
    const float sx = ((float)rx + 0.5f) * ((float)Win / (float)R) - 0.5f;
    int y0 = (int)floorf(sy), x0 = (int)floorf(sx);
    float fy = sy - (float)y0, fx = sx - (float)x0;
    if (y0 < 0) { y0 = 0; fy = 0.f; }
    if (x0 < 0) { x0 = 0; fx = 0.f; }
    int y1 = y0 + 1, x1 = x0 + 1;
    if (y1 >= Hin) { y1 = Hin - 1; if (y0 >= Hin - 1) { y0 = Hin - 1; fy = 0.f; } }
    if (x1 >= Win) { x1 = Win - 1; if (x0 >= Win - 1) { x0 = Win - 1; fx = 0.f; } }
    for (int c = 0; c < C; ++c) {
        const float p00 = src[((long)y0 * Win + x0) * C + c], p01 = src[((long)y0 * Win + x1) * C + c];
        const float p10 = src[((long)y1 * Win + x0) * C + c], p11 = src[((long)y1 * Win + x1) * C + c];
        const float top = p00 + (p01 - p00) * fx, bot = p10 + (p11 - p10) * fx;
        const float v = fminf(fmaxf(floorf(top + (bot - top) * fy + 0.5f), 0.f), 255.f);
        out[(long)idx * C + c] = (uint8_t)v;
    }
}

// ---- CLAHE (OpenCV clahe.cpp), 8 x 8 tiles, on one uint8 plane with pixel stride `ps` ---------------------------
__device__ __forceinline__ int reflect101(int p, int n) {   // BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

// one block per tile: histogram of the (reflect-padded) tile, clip, redistribute, cumulative LUT
__global__ __launch_bounds__(256) void clahe_lut_kernel(const uint8_t* __restrict__ img, int H, int W, int ps, int tw,
                                                        int th, int clip, uint8_t* __restrict__ lut) {
    __shared__ int hist[256];
    __shared__ int scan[256];
    const int tx = blockIdx.x, ty = blockIdx.y, t = threadIdx.x;
    hist[t] = 0;
    __syncthreads();
    for (int i = t; i < tw * th; i += 256) {
        const int y = reflect101(ty * th + i / tw, H), x = reflect101(tx * tw + i % tw, W);
        atomicAdd(&hist[img[((long)y * W + x) * ps]], 1);
    }
    __syncthreads();
    if (clip > 0) {
        // clipped = sum of the excesses; every bin gets clipped / 256, the residual goes to bins 0, step, 2 step, ...
        int v = hist[t];
        const int ex = v > clip ? v - clip : 0;
        scan[t] = ex;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) scan[t] += scan[t + o];
            __syncthreads();
        }
        const int clipped = scan[0];
        __syncthreads();
        const int batch = clipped / 256;
        int residual = clipped - batch * 256;
        v = (v > clip ? clip : v) + batch;
        if (residual != 0) {
            int step = 256 / residual;
            if (step < 1) step = 1;
            if (t % step == 0 && t / step < residual) ++v;
        }
        hist[t] = v;
        __syncthreads();
    }
    // inclusive prefix sum (Hillis-Steele), then lut = saturate(round_half_even(sum * 255 / tile_area))
    scan[t] = hist[t];
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int add = t >= o ? scan[t - o] : 0;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    const float scale = 255.0f / (float)(tw * th);
    float r = rintf((float)scan[t] * scale);
    r = fminf(fmaxf(r, 0.f), 255.f);
    lut[((long)(ty * gridDim.x + tx)) * 256 + t] = (uint8_t)r;
}

__global__ __launch_bounds__(256) void clahe_apply_kernel(const uint8_t* __restrict__ img, int H, int W, int ps, int tw,
                                                          int th, int tiles, const uint8_t* __restrict__ lut,
                                                          uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const float txf = (float)x * (1.0f / (float)tw) - 0.5f, tyf = (float)y * (1.0f / (float)th) - 0.5f;
    int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
    const float xa = txf - (float)tx1, ya = tyf - (float)ty1;
    int tx2 = tx1 + 1, ty2 = ty1 + 1;
    tx1 = tx1 < 0 ? 0 : tx1;
    ty1 = ty1 < 0 ? 0 : ty1;
    tx2 = tx2 > tiles - 1 ? tiles - 1 : tx2;
    ty2 = ty2 > tiles - 1 ? tiles - 1 : ty2;
    const int v = img[(long)idx * ps];
    const float l11 = lut[(ty1 * tiles + tx1) * 256 + v], l12 = lut[(ty1 * tiles + tx2) * 256 + v];
    const float l21 = lut[(ty2 * tiles + tx1) * 256 + v], l22 = lut[(ty2 * tiles + tx2) * 256 + v];
    const float res = (l11 * (1.0f - xa) + l12 * xa) * (1.0f - ya) + (l21 * (1.0f - xa) + l22 * xa) * ya;
    out[(long)idx * ps] = (uint8_t)fminf(fmaxf(rintf(res), 0.f), 255.f);
}

// ---- RGB <-> CIE L*a*b* (D65, sRGB transfer), 8 bit: L * 255 / 100, a + 128, b + 128 ---------------------------
__device__ __forceinline__ float srgb_to_linear(float c) { return c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float linear_to_srgb(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f; }
__device__ __forceinline__ float lab_f(float t) { return t > 0.008856f ? cbrtf(t) : 7.787f * t + 16.0f / 116.0f; }
__device__ __forceinline__ uint8_t sat_u8(float v) { return (uint8_t)fminf(fmaxf(rintf(v), 0.f), 255.f); }

__global__ __launch_bounds__(256) void rgb_lab_kernel(const uint8_t* __restrict__ in, long n, int inverse,
                                                      uint8_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p0 = in[3 * i], p1 = in[3 * i + 1], p2 = in[3 * i + 2];
    if (!inverse) {
        const float r = srgb_to_linear(p0 / 255.f), g = srgb_to_linear(p1 / 255.f), b = srgb_to_linear(p2 / 255.f);
        const float X = (0.412453f * r + 0.357580f * g + 0.180423f * b) / 0.950456f;
        const float Y = 0.212671f * r + 0.715160f * g + 0.072169f * b;
        const float Z = (0.019334f * r + 0.119193f * g + 0.950227f * b) / 1.088754f;
        const float fx = lab_f(X), fy = lab_f(Y), fz = lab_f(Z);
        const float L = Y > 0.008856f ? 116.f * fy - 16.f : 903.3f * Y;
        out[3 * i] = sat_u8(L * 255.f / 100.f);
        out[3 * i + 1] = sat_u8(500.f * (fx - fy) + 128.f);
        out[3 * i + 2] = sat_u8(200.f * (fy - fz) + 128.f);
    } else {
        const float L = p0 * 100.f / 255.f, a = p1 - 128.f, b = p2 - 128.f;
        const float fy = (L + 16.f) / 116.f, fx = fy + a / 500.f, fz = fy - b / 200.f;
        auto inv = [](float t) { return t > 0.206893f ? t * t * t : (t - 16.f / 116.f) / 7.787f; };
        const float X = inv(fx) * 0.950456f, Y = L > 7.9996f ? inv(fy) : L / 903.3f, Z = inv(fz) * 1.088754f;
        const float r = 3.240479f * X - 1.537150f * Y - 0.498535f * Z;
        const float g = -0.969256f * X + 1.875991f * Y + 0.041556f * Z;
        const float bl = 0.055648f * X - 0.204043f * Y + 1.057311f * Z;
        out[3 * i] = sat_u8(linear_to_srgb(fminf(fmaxf(r, 0.f), 1.f)) * 255.f);
        out[3 * i + 1] = sat_u8(linear_to_srgb(fminf(fmaxf(g, 0.f), 1.f)) * 255.f);
        out[3 * i + 2] = sat_u8(linear_to_srgb(fminf(fmaxf(bl, 0.f), 1.f)) * 255.f);
    }
}

// ---- cv2.LUT with a 256-entry table (RandomGamma, RandomBrightness) -------------------------------------------------
__global__ __launch_bounds__(256) void lut_u8_kernel(const uint8_t* __restrict__ in, long n, const uint8_t* __restrict__ table,
                                                     uint8_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[in[i]];
}

// ---- cv2.blur(img, (k, k)): normalised box filter, BORDER_REFLECT_101, rounded to nearest -------------------------
__global__ __launch_bounds__(256) void box_blur_u8_kernel(const uint8_t* __restrict__ in, int H, int W, int C, int k,
                                                          uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W, a = k / 2;      // window [p - k / 2, p - k / 2 + k - 1]: cv2's default anchor
    for (int c = 0; c < C; ++c) {
        int s = 0;
        for (int dy = -a; dy < k - a; ++dy)
            for (int dx = -a; dx < k - a; ++dx)
                s += in[((long)reflect101(y + dy, H) * W + reflect101(x + dx, W)) * C + c];
        out[(long)idx * C + c] = sat_u8((float)s / (float)(k * k));
    }
}

// ---- the warping transforms of albumentations 0.4.6 (ElasticTransform, OpticalDistortion, GridDistortion) ----------
// All three are cv2.remap(img, map_x, map_y, INTER_LINEAR, BORDER_REFLECT_101) with a generated coordinate field
// (torchlib/dataloader.py:167-172): the maps are formed on the device from the few parameters the host draws, one remap
// kernel samples them.  Bilinear weights in fp32, result rounded to nearest even (cv2's 8-bit path uses 5-bit fixed-point
// coordinates: unpinned, see oracle/augment_oracle.py).
__global__ __launch_bounds__(256) void remap_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int C,
                                                       const float* __restrict__ mx, const float* __restrict__ my,
                                                       uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const float x = mx[idx], y = my[idx];
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    // (coordinates far outside the image — a degenerate affine draw — are clamped before the integer conversion; the
    // reflection below is periodic, so the clamp only has to keep the value representable)
    const int x0 = (int)fminf(fmaxf(x0f, -1.0e6f), 1.0e6f), y0 = (int)fminf(fmaxf(y0f, -1.0e6f), 1.0e6f);
    const int xa = reflect101(x0, W), xb = reflect101(x0 + 1, W), ya = reflect101(y0, H), yb = reflect101(y0 + 1, H);
    for (int c = 0; c < C; ++c) {
        const float p00 = src[((long)ya * W + xa) * C + c], p01 = src[((long)ya * W + xb) * C + c];
        const float p10 = src[((long)yb * W + xa) * C + c], p11 = src[((long)yb * W + xb) * C + c];
        const float top = p00 * (1.f - fx) + p01 * fx, bot = p10 * (1.f - fx) + p11 * fx;
        out[(long)idx * C + c] = sat_u8(top * (1.f - fy) + bot * fy);
    }
}

// kind 0: affine  (p = inverse matrix a b c d e f: source = (a x + b y + c, d x + e y + f)), cv2.warpAffine
// kind 1: optical (p = k, fx, fy, cx, cy, ncx, ncy): cv2.initUndistortRectifyMap with distortion (k, k, 0, 0, 0)
__global__ __launch_bounds__(256) void warp_map_kernel(int H, int W, int kind, double p0, double p1, double p2, double p3,
                                                       double p4, double p5, double p6, float* __restrict__ mx,
                                                       float* __restrict__ my) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const double y = idx / W, x = idx - (idx / W) * W;
    if (kind == 0) {
        mx[idx] = (float)(p0 * x + p1 * y + p2);
        my[idx] = (float)(p3 * x + p4 * y + p5);
    } else {
        const double u = (x - p5) / p1, v = (y - p6) / p2;
        const double r2 = u * u + v * v;
        const double kr = 1.0 + p0 * r2 + p0 * r2 * r2;
        mx[idx] = (float)(p1 * (u * kr) + p3);
        my[idx] = (float)(p2 * (v * kr) + p4);
    }
}

// GridDistortion: map_x, map_y = meshgrid(xx, yy)  |  ElasticTransform: map = float32(index + displacement)
__global__ __launch_bounds__(256) void grid_map_kernel(int H, int W, const float* __restrict__ xx, const float* __restrict__ yy,
                                                       const float* __restrict__ dx, const float* __restrict__ dy,
                                                       float* __restrict__ mx, float* __restrict__ my) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    mx[idx] = xx ? xx[x] : (float)x + dx[idx];
    my[idx] = yy ? yy[y] : (float)y + dy[idx];
}

// scipy.ndimage.gaussian_filter's 1-D pass (correlate1d, mode "reflect": d c b a | a b c d | d c b a), float64, on
// u = 2 r - 1 of a uniform field r (first pass) or on the first pass's output; `scale`: factor applied to the result
// (alpha after the second pass), which is stored as float32 when `out32` is given.
__global__ __launch_bounds__(256) void gauss1d_kernel(const double* __restrict__ in, int H, int W, int axis, double sigma,
                                                      int radius, int affine_in, double scale, double* __restrict__ out,
                                                      float* __restrict__ out32) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    const int n = axis == 0 ? H : W, p = axis == 0 ? y : x;
    double wsum = 0.0;
    for (int t = -radius; t <= radius; ++t) wsum += exp(-0.5 / (sigma * sigma) * (double)t * (double)t);
    double acc = 0.0;
    for (int t = -radius; t <= radius; ++t) {
        int q = p + t;
        const int period = 2 * n;                      // half-sample symmetric reflection
        q %= period;
        if (q < 0) q += period;
        if (q >= n) q = period - 1 - q;
        double v = in[axis == 0 ? (long)q * W + x : (long)y * W + q];
        if (affine_in) v = v * 2.0 - 1.0;
        acc += v * (exp(-0.5 / (sigma * sigma) * (double)t * (double)t) / wsum);
    }
    acc *= scale;
    if (out32) out32[idx] = (float)acc;
    else out[idx] = acc;
}

// ---- RandomFog (F.add_fog): per haze point a white disc of radius hw / 2 blended in with cv2.addWeighted(alpha) ---------
// sequentially (a pixel covered by m discs is blended m times, in list order); the cv2.blur(hw / 10) that follows is
// primia_image_box_blur_u8.
__global__ __launch_bounds__(256) void fog_u8_kernel(const uint8_t* __restrict__ in, int H, int W, int C,
                                                     const int* __restrict__ haze, int n, int hw, float alpha, float beta,
                                                     uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W, rad = hw / 2;
    float v[3];
    for (int c = 0; c < C; ++c) v[c] = in[(long)idx * C + c];
    for (int i = 0; i < n; ++i) {
        const int dx = x - (haze[2 * i] + hw / 2), dy = y - (haze[2 * i + 1] + hw / 2);
        if (dx * dx + dy * dy <= rad * rad)
            for (int c = 0; c < C; ++c) v[c] = fminf(fmaxf(rintf(255.f * alpha + v[c] * beta), 0.f), 255.f);
    }
    for (int c = 0; c < C; ++c) out[(long)idx * C + c] = (uint8_t)v[c];
}

// ---- GaussNoise: image + noise (fp32, given), clipped to [0, 255], cast to uint8 (truncation, as ndarray.astype) -----
__global__ __launch_bounds__(256) void add_noise_u8_kernel(const uint8_t* __restrict__ in, const float* __restrict__ noise,
                                                           long n, uint8_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint8_t)fminf(fmaxf((float)in[i] + noise[i], 0.f), 255.f);
}

// ---- ToFloat(255) + Normalize(mean, std, max_pixel_value = 1): uint8 HWC -> fp32 CHW ------------------------------
__global__ __launch_bounds__(256) void finish_u8_kernel(const uint8_t* __restrict__ in, int S, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ stdv,
                                                        float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * S) return;
    for (int c = 0; c < C; ++c) {
        float v = (float)in[(long)idx * C + c] / 255.0f;
        if (mean) v = (v - mean[c]) / stdv[c];
        out[(long)c * S * S + idx] = v;
    }
}

// ---- round 4, second batch: RandomGridShuffle, HueSaturationValue, Cutout / GridDropout, RandomShadow, RandomSunFlare,
// Equalize (InvertImg and Solarize are tables for lut_u8_kernel).  albumentations 0.4.6 functional.py + OpenCV's
// color_hsv.cpp / histogram.cpp, restated in oracle/augment_oracle.py (where the raster approximations are named).

// cv2.equalizeHist: per-channel histogram (int32 [C][256], zeroed by the caller), table, apply
__global__ __launch_bounds__(256) void hist_u8_kernel(const uint8_t* __restrict__ in, long npix, int C, int* __restrict__ hist) {
    __shared__ int h[3 * 256];
    for (int i = threadIdx.x; i < C * 256; i += 256) h[i] = 0;
    __syncthreads();
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix * C; i += stride) atomicAdd(&h[(int)(i % C) * 256 + in[i]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < C * 256; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

__global__ __launch_bounds__(64) void equalize_lut_kernel(const int* __restrict__ hist, long total, uint8_t* __restrict__ lut) {
    if (threadIdx.x != 0) return;           // 256 bins: one thread per channel walks them (histogram.cpp's scalar loop)
    const int* h = hist + blockIdx.x * 256;
    uint8_t* t = lut + blockIdx.x * 256;
    int i = 0;
    while (i < 255 && !h[i]) ++i;
    if (h[i] == total) {
        for (int k = 0; k < 256; ++k) t[k] = (uint8_t)i;
        return;
    }
    const float scale = 255.f / (float)(total - h[i]);
    for (int k = 0; k <= i; ++k) t[k] = 0;
    long sum = 0;
    for (int k = i + 1; k < 256; ++k) {
        sum += h[k];
        const float v = rintf((float)sum * scale);
        t[k] = (uint8_t)fminf(fmaxf(v, 0.f), 255.f);
    }
}

__global__ __launch_bounds__(256) void lut_c_u8_kernel(const uint8_t* __restrict__ in, long n, int C,
                                                       const uint8_t* __restrict__ tables, uint8_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tables[(int)(i % C) * 256 + in[i]];
}

// F.cutout: every rectangle (x1, y1, x2, y2) filled (Cutout's five holes, GridDropout's grid)
__global__ __launch_bounds__(256) void fill_rects_u8_kernel(uint8_t* __restrict__ img, int H, int W, int C,
                                                            const int* __restrict__ rects, int n, int fill) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    bool hit = false;
    for (int k = 0; k < n && !hit; ++k) hit = x >= rects[4 * k] && x < rects[4 * k + 2] && y >= rects[4 * k + 1] && y < rects[4 * k + 3];
    if (hit)
        for (int c = 0; c < C; ++c) img[(long)idx * C + c] = (uint8_t)fill;
}

// F.swap_tiles_on_image: tile k = (y, x, old_y, old_x, h, w): dst[y.., x..] = src[old_y.., old_x..]; the tiles partition the image
__global__ __launch_bounds__(256) void swap_tiles_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int C,
                                                            const int* __restrict__ tiles, int n, uint8_t* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    int sy = y, sx = x;
    for (int k = 0; k < n; ++k) {
        const int* t = tiles + 6 * k;
        if (y >= t[0] && y < t[0] + t[4] && x >= t[1] && x < t[1] + t[5]) {
            sy = t[2] + (y - t[0]);
            sx = t[3] + (x - t[1]);
        }
    }
    for (int c = 0; c < C; ++c) dst[(long)idx * C + c] = src[((long)sy * W + sx) * C + c];
}


// (b, g, r) = tab[sector_data[sector]] of OpenCV's HSV2RGB_f / HLS2RGB_f
__device__ __forceinline__ void sector_rgb(const float* tab, int sector, float& r, float& g, float& b) {
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    b = tab[sd[sector][0]];
    g = tab[sd[sector][1]];
    r = tab[sd[sector][2]];
}

// F._shift_hsv_uint8: RGB2HSV_b (12-bit fixed-point division tables, hrange 180) -> three tables -> HSV2RGB_f in float32
__global__ __launch_bounds__(256) void hsv_shift_u8_kernel(const uint8_t* __restrict__ in, long npix,
                                                           const uint8_t* __restrict__ luts, uint8_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int r = in[3 * i], g = in[3 * i + 1], b = in[3 * i + 2];
    const int v = max(max(r, g), b), vmin = min(min(r, g), b), diff = v - vmin;
    const int sdiv = v ? (int)rint((double)(255 << 12) / (1.0 * v)) : 0;
    const int hdiv = diff ? (int)rint((double)(180 << 12) / (6.0 * diff)) : 0;
    const int s = (diff * sdiv + (1 << 11)) >> 12;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * hdiv + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    const int h2 = luts[h & 255], s2 = luts[256 + (s & 255)], v2 = luts[512 + v];
    float hf = (float)h2 * (float)(6.0 / 180.0);
    const float sf = (float)s2 * (float)(1.0 / 255.0), vf = (float)v2 * (float)(1.0 / 255.0);
    float rr = vf, gg = vf, bb = vf;
    if (sf != 0.f) {
        if (hf >= 6.f) hf -= 6.f;
        int sector = (int)floorf(hf);
        hf -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; hf = 0.f; }
        const float tab[4] = {vf, vf * (1.f - sf), vf * (1.f - sf * hf), vf * (1.f - sf * (1.f - hf))};
        sector_rgb(tab, sector, rr, gg, bb);
    }
    out[3 * i] = sat_u8(rr * 255.f);
    out[3 * i + 1] = sat_u8(gg * 255.f);
    out[3 * i + 2] = sat_u8(bb * 255.f);
}

// F.add_shadow: RGB2HLS_f on x / 255 -> uint8 (h / 2, 255 l, 255 s); L >>= 1 under the union of the polygons (even-odd
// interior at integer pixel coordinates, or on an edge's DDA line); HLS2RGB_f
__global__ __launch_bounds__(256) void shadow_u8_kernel(const uint8_t* __restrict__ in, int H, int W,
                                                        const int* __restrict__ verts, int npoly, int nv,
                                                        uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    bool covered = false;
    for (int pnum = 0; pnum < npoly; ++pnum) {
        const int* pv = verts + (long)pnum * nv * 2;
        bool inside = false, edge = false;
        for (int k = 0; k < nv; ++k) {
            const int x0 = pv[2 * k], y0 = pv[2 * k + 1];
            const int k1 = k + 1 == nv ? 0 : k + 1;
            const int x1 = pv[2 * k1], y1 = pv[2 * k1 + 1];
            if (y0 != y1 && ((y0 <= y) != (y1 <= y))) {
                const long dy = y1 - y0, lhs = (long)x * dy, rhs = (long)x0 * dy + (long)(y - y0) * (x1 - x0);
                if (dy > 0 ? lhs < rhs : lhs > rhs) inside = !inside;
            }
            const int adx = abs(x1 - x0), ady = abs(y1 - y0);
            if (adx >= ady) {
                const bool fw = x0 <= x1;
                const int xa = fw ? x0 : x1, ya = fw ? y0 : y1, xb = fw ? x1 : x0, yb = fw ? y1 : y0;
                const int d = xb - xa;
                if (d == 0) {
                    edge |= x == xa && y == ya;
                } else if (x >= xa && x <= xb) {
                    const long num = 2L * (x - xa) * (yb - ya) + d, den = 2L * d;
                    long q = num / den;
                    if (num % den != 0 && num < 0) --q;          // floor division
                    edge |= y == ya + (int)q;
                }
            } else {
                const bool fw = y0 <= y1;
                const int xa = fw ? x0 : x1, ya = fw ? y0 : y1, xb = fw ? x1 : x0, yb = fw ? y1 : y0;
                const int d = yb - ya;
                if (y >= ya && y <= yb) {
                    const long num = 2L * (y - ya) * (xb - xa) + d, den = 2L * d;
                    long q = num / den;
                    if (num % den != 0 && num < 0) --q;
                    edge |= x == xa + (int)q;
                }
            }
        }
        covered |= inside || edge;
    }
    const float k255 = (float)(1.0 / 255.0);
    const float r = (float)in[3L * idx] * k255, g = (float)in[3L * idx + 1] * k255, b = (float)in[3L * idx + 2] * k255;
    const float vmax = fmaxf(fmaxf(r, g), b), vmin = fminf(fminf(r, g), b);
    float diff = vmax - vmin, h = 0.f, s = 0.f;
    const float l = (vmax + vmin) * 0.5f;
    if (diff > 1.1920929e-07f) {
        s = l < 0.5f ? diff / (vmax + vmin) : diff / (2.f - vmax - vmin);
        diff = 60.f / diff;
        if (vmax == r) h = (g - b) * diff;
        else if (vmax == g) h = (b - r) * diff + 120.f;
        else h = (r - g) * diff + 240.f;
        if (h < 0.f) h += 360.f;
    }
    const int hq = sat_u8(h * 0.5f), sq = sat_u8(s * 255.f);
    int lq = sat_u8(l * 255.f);
    if (covered) lq >>= 1;
    // back
    float hf = (float)hq * (float)(6.0 / 180.0);
    const float lf = (float)lq * k255, sf = (float)sq * k255;
    float rr = lf, gg = lf, bb = lf;
    if (sf != 0.f) {
        const float p2 = lf <= 0.5f ? lf * (1.f + sf) : lf + sf - lf * sf;
        const float p1 = 2.f * lf - p2;
        if (hf >= 6.f) hf -= 6.f;
        int sector = (int)floorf(hf);
        hf -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; hf = 0.f; }
        const float tab[4] = {p2, p1, p1 + (p2 - p1) * (1.f - hf), p1 + (p2 - p1) * hf};
        sector_rgb(tab, sector, rr, gg, bb);
    }
    out[3L * idx] = sat_u8(rr * 255.f);
    out[3L * idx + 1] = sat_u8(gg * 255.f);
    out[3L * idx + 2] = sat_u8(bb * 255.f);
}

// F.add_sun_flare: step k draws a filled circle (x, y, r, colour) onto `overlay` and blends overlay into output with weight
// alpha[k]; at step n_first overlay = output.copy().  Every pixel walks the steps on its own.
__global__ __launch_bounds__(256) void sun_flare_u8_kernel(const uint8_t* __restrict__ in, int H, int W,
                                                           const int* __restrict__ geo, const float* __restrict__ alpha,
                                                           const float* __restrict__ beta, int n, int n_first,
                                                           uint8_t* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W;
    float ov[3], o[3];
    for (int c = 0; c < 3; ++c) ov[c] = o[c] = (float)in[3L * idx + c];
    for (int k = 0; k < n; ++k) {
        if (k == n_first)
            for (int c = 0; c < 3; ++c) ov[c] = o[c];
        const int* g = geo + 6 * k;
        const long dx = x - g[0], dy = y - g[1], r = g[2];
        if (dx * dx + dy * dy <= r * r)
            for (int c = 0; c < 3; ++c) ov[c] = (float)g[3 + c];
        const float a = alpha[k], b = beta[k];
        for (int c = 0; c < 3; ++c) o[c] = fminf(fmaxf(rintf(ov[c] * a + o[c] * b), 0.f), 255.f);
    }
    for (int c = 0; c < 3; ++c) out[3L * idx + c] = (uint8_t)o[c];
}

}  // namespace primia

using namespace primia;

extern "C" {

int primia_image_affine_u8(const uint8_t* src, int H, int W, int C, float a, float b, float c, float d, float e, float f,
                           uint8_t* out, primia_stream_t st) {
    PRIMIA_REQUIRE(src && out && src != out && H > 0 && W > 0 && (C == 1 || C == 3));
    affine_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(src, H, W, C, a, b, c, d, e, f, out);
    return launch_status();
}

int primia_image_resize_crop_u8(const uint8_t* src, int Hin, int Win, int C, int R, int oy, int ox, int S, int flip_v,
                                uint8_t* out, primia_stream_t st) {
    PRIMIA_REQUIRE(src && out && Hin > 0 && Win > 0 && (C == 1 || C == 3) && R > 0 && S > 0 && S <= R);
    PRIMIA_REQUIRE(oy >= 0 && ox >= 0 && oy + S <= R && ox + S <= R);
    resize_crop_u8_kernel<<<ceil_div((long)S * S, 256), 256, 0, (hipStream_t)st>>>(src, Hin, Win, C, R, oy, ox, S, flip_v, out);
    return launch_status();
}

int64_t primia_clahe_workspace_bytes(int H, int W, int C) { return 64 * 256 + (C == 3 ? (int64_t)H * W * 3 : 0); }

int primia_clahe_u8(const uint8_t* img, int H, int W, int C, float clip_limit, void* workspace, int64_t workspace_bytes,
                    uint8_t* out, primia_stream_t stream) {
    PRIMIA_REQUIRE(img && out && workspace && H >= 8 && W >= 8 && (C == 1 || C == 3) && clip_limit >= 0.f);
    if (workspace_bytes < primia_clahe_workspace_bytes(H, W, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = 8;
    const int tw = (W + tiles - 1) / tiles, th = (H + tiles - 1) / tiles;      // tile size of the padded image
    int clip = 0;
    if (clip_limit > 0.f) {
        clip = (int)(clip_limit * (float)(tw * th) / 256.0f);
        if (clip < 1) clip = 1;
    }
    uint8_t* lut = (uint8_t*)workspace;
    const long n = (long)H * W;
    if (C == 1) {
        clahe_lut_kernel<<<dim3(tiles, tiles), 256, 0, st>>>(img, H, W, 1, tw, th, clip, lut);
        clahe_apply_kernel<<<ceil_div(n, 256), 256, 0, st>>>(img, H, W, 1, tw, th, tiles, lut, out);
    } else {
        uint8_t* lab = lut + 64 * 256;
        rgb_lab_kernel<<<ceil_div(n, 256), 256, 0, st>>>(img, n, 0, lab);
        clahe_lut_kernel<<<dim3(tiles, tiles), 256, 0, st>>>(lab, H, W, 3, tw, th, clip, lut);
        clahe_apply_kernel<<<ceil_div(n, 256), 256, 0, st>>>(lab, H, W, 3, tw, th, tiles, lut, lab);   // L in place (pointwise)
        rgb_lab_kernel<<<ceil_div(n, 256), 256, 0, st>>>(lab, n, 1, out);
    }
    return launch_status();
}

int primia_image_lut_u8(const uint8_t* in, int64_t n, const uint8_t* table256, uint8_t* out, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(in && out && table256 && n > 0);
    lut_u8_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)st>>>(in, n, table256, out);
    return launch_status();
}

int primia_image_box_blur_u8(const uint8_t* in, int H, int W, int C, int k, uint8_t* out, primia_stream_t st) {
    PRIMIA_REQUIRE(in && out && in != out && H > 0 && W > 0 && (C == 1 || C == 3) && k >= 1 && k <= 63);
    box_blur_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(in, H, W, C, k, out);
    return launch_status();
}

int primia_image_remap_u8(const uint8_t* src, int H, int W, int C, const float* map_x, const float* map_y, uint8_t* out,
                          primia_stream_t st) {
    PRIMIA_REQUIRE(src && out && src != out && map_x && map_y && H > 0 && W > 0 && (C == 1 || C == 3));
    remap_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(src, H, W, C, map_x, map_y, out);
    return launch_status();
}

int primia_warp_map_affine(int H, int W, double a, double b, double c, double d, double e, double f, float* map_x,
                           float* map_y, primia_stream_t st) {
    PRIMIA_REQUIRE(map_x && map_y && H > 0 && W > 0);
    warp_map_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(H, W, 0, a, b, c, d, e, f, 0.0, map_x, map_y);
    return launch_status();
}

int primia_warp_map_optical(int H, int W, double k, double fx, double fy, double cx, double cy, double new_cx,
                            double new_cy, float* map_x, float* map_y, primia_stream_t st) {
    PRIMIA_REQUIRE(map_x && map_y && H > 0 && W > 0 && fx != 0.0 && fy != 0.0);
    warp_map_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(H, W, 1, k, fx, fy, cx, cy, new_cx, new_cy, map_x,
                                                                              map_y);
    return launch_status();
}

int primia_warp_map_grid(int H, int W, const float* xx, const float* yy, float* map_x, float* map_y, primia_stream_t st) {
    PRIMIA_REQUIRE(xx && yy && map_x && map_y && H > 0 && W > 0);
    grid_map_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(H, W, xx, yy, nullptr, nullptr, map_x, map_y);
    return launch_status();
}

int primia_warp_map_elastic(int H, int W, const double* field_x, const double* field_y, double sigma, double alpha,
                            void* workspace, int64_t workspace_bytes, float* map_x, float* map_y, primia_stream_t st) {
    PRIMIA_REQUIRE(field_x && field_y && workspace && map_x && map_y && H > 0 && W > 0 && sigma > 0.0);
    // workspace: one float64 plane (first pass) + two float32 planes (the displacements)
    const int64_t need = (int64_t)H * W * (8 + 4 + 4);
    if (workspace_bytes < need) return PRIMIA_ERR_WORKSPACE;
    double* tmp = (double*)workspace;
    float* dx = (float*)(tmp + (long)H * W);
    float* dy = dx + (long)H * W;
    const int radius = (int)(4.0 * sigma + 0.5);          // scipy: truncate = 4.0
    const unsigned grid = ceil_div((long)H * W, 256);
    hipStream_t s = (hipStream_t)st;
    gauss1d_kernel<<<grid, 256, 0, s>>>(field_x, H, W, 0, sigma, radius, 1, 1.0, tmp, nullptr);
    gauss1d_kernel<<<grid, 256, 0, s>>>(tmp, H, W, 1, sigma, radius, 0, alpha, nullptr, dx);
    gauss1d_kernel<<<grid, 256, 0, s>>>(field_y, H, W, 0, sigma, radius, 1, 1.0, tmp, nullptr);
    gauss1d_kernel<<<grid, 256, 0, s>>>(tmp, H, W, 1, sigma, radius, 0, alpha, nullptr, dy);
    grid_map_kernel<<<grid, 256, 0, s>>>(H, W, nullptr, nullptr, dx, dy, map_x, map_y);
    return launch_status();
}

int primia_image_fog_u8(const uint8_t* in, int H, int W, int C, const int32_t* haze_xy, int n_haze, int hw, float alpha,
                        uint8_t* out, primia_stream_t st) {
    PRIMIA_REQUIRE(in && out && H > 0 && W > 0 && (C == 1 || C == 3) && n_haze >= 0 && (haze_xy || n_haze == 0) && hw >= 1);
    fog_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(in, H, W, C, (const int*)haze_xy, n_haze, hw, alpha,
                                                                          1.f - alpha, out);
    return launch_status();
}

int primia_image_add_noise_u8(const uint8_t* in, const float* noise, int64_t n, uint8_t* out, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(in && noise && out && n > 0);
    add_noise_u8_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)st>>>(in, noise, n, out);
    return launch_status();
}

int primia_image_finish(const uint8_t* in, int S, int C, const float* mean, const float* stdv, float* out,
                        primia_stream_t st) {
    PRIMIA_REQUIRE(in && out && S > 0 && (C == 1 || C == 3) && ((mean == nullptr) == (stdv == nullptr)));
    finish_u8_kernel<<<ceil_div((long)S * S, 256), 256, 0, (hipStream_t)st>>>(in, S, C, mean, stdv, out);
    return launch_status();
}

int primia_image_equalize_u8(const uint8_t* in, int H, int W, int C, void* workspace, int64_t workspace_bytes, uint8_t* out,
                             primia_stream_t st) {
    PRIMIA_REQUIRE(in && out && workspace && H > 0 && W > 0 && (C == 1 || C == 3) && workspace_bytes >= 3 * 256 * 5);
    hipStream_t s = (hipStream_t)st;
    int* hist = (int*)workspace;
    uint8_t* lut = (uint8_t*)workspace + 3 * 256 * 4;
    if (hipMemsetAsync(hist, 0, 3 * 256 * 4, s) != hipSuccess) return PRIMIA_ERR_LAUNCH;
    const long npix = (long)H * W;
    const int nb = (int)(ceil_div(npix * C, 256) < 512 ? ceil_div(npix * C, 256) : 512);
    hist_u8_kernel<<<nb, 256, 0, s>>>(in, npix, C, hist);
    equalize_lut_kernel<<<C, 64, 0, s>>>(hist, npix, lut);
    lut_c_u8_kernel<<<ceil_div(npix * C, 256), 256, 0, s>>>(in, npix * C, C, lut, out);
    return launch_status();
}

int primia_image_fill_rects_u8(uint8_t* img, int H, int W, int C, const int32_t* rects, int n, int fill, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(img && rects && H > 0 && W > 0 && (C == 1 || C == 3) && n > 0 && fill >= 0 && fill <= 255);
    fill_rects_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(img, H, W, C, (const int*)rects, n, fill);
    return launch_status();
}

int primia_image_swap_tiles_u8(const uint8_t* src, int H, int W, int C, const int32_t* tiles, int n, uint8_t* dst,
                               primia_stream_t st) {
    PRIMIA_REQUIRE(src && dst && src != dst && tiles && H > 0 && W > 0 && (C == 1 || C == 3) && n > 0);
    swap_tiles_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(src, H, W, C, (const int*)tiles, n, dst);
    return launch_status();
}

int primia_image_hsv_shift_u8(const uint8_t* in, int H, int W, const uint8_t* luts3x256, uint8_t* out, primia_stream_t st) {
    PRIMIA_REQUIRE(in && out && luts3x256 && H > 0 && W > 0);
    hsv_shift_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(in, (long)H * W, luts3x256, out);
    return launch_status();
}

int primia_image_shadow_u8(const uint8_t* in, int H, int W, const int32_t* vertices, int n_polygons, int n_vertices,
                           uint8_t* out, primia_stream_t st) {
    PRIMIA_REQUIRE(in && out && vertices && H > 0 && W > 0 && n_polygons > 0 && n_vertices >= 3);
    shadow_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(in, H, W, (const int*)vertices, n_polygons,
                                                                             n_vertices, out);
    return launch_status();
}

int primia_image_sun_flare_u8(const uint8_t* in, int H, int W, const int32_t* steps, const float* alpha, const float* beta,
                              int n_steps, int n_first, uint8_t* out, primia_stream_t st) {
    PRIMIA_REQUIRE(in && out && steps && alpha && beta && H > 0 && W > 0 && n_steps > 0 && n_first >= 0 && n_first <= n_steps);
    sun_flare_u8_kernel<<<ceil_div((long)H * W, 256), 256, 0, (hipStream_t)st>>>(in, H, W, (const int*)steps, alpha, beta,
                                                                                n_steps, n_first, out);
    return launch_status();
}

}  // extern "C"
