// Device-side pieces of PriMIA's data path in front of the training step: MixUp
// pair mixing, one-hot targets and the per-channel dataset statistics — HBM streaming kernels.
//
// Reference semantics:
//   MixUp       torchlib/utils.py:337-400   out = λ·x[:h] + (1-λ)·x[h:] (an odd trailing sample is passed through)
//   To_one_hot  torchlib/utils.py:449-466   float32 rows of a [N, classes] matrix
//   mean / std  torchlib/dataloader.py:220-247 (torch.std_mean over (N, H, W), unbiased)
#include "common.h"

namespace primia {

// out[i] = fl(fl(lam * a[i]) + fl(mu * b[i])): the three roundings of `λ * x[:h] + (1.0 - λ) * x[h:]` in torch
// (no fma contraction, or the result differs from the reference in the last bit)
__global__ __launch_bounds__(256) void mixup_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, long n4, long n, float lam, float mu) {
#pragma clang fp contract(off)  // hipcc's default contraction would fuse the multiply into the add (1 ulp off)
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 va = ((const f32x4*)a)[i], vb = ((const f32x4*)b)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = lam * va[k] + mu * vb[k];
        ((f32x4*)out)[i] = o;
    }
    // tail (n not a multiple of 4)
    for (long i = n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = lam * a[i] + mu * b[i];
}

__global__ __launch_bounds__(256) void one_hot_kernel(const int64_t* __restrict__ labels, float* __restrict__ out,
                                                      long n, int classes) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * classes) return;
    const long r = i / classes;
    const int c = (int)(i - r * classes);
    out[i] = labels[r] == (int64_t)c ? 1.f : 0.f;
}

// per-channel (sum, sum of squares) of an NCHW fp32 tensor: block (slab, c) reduces its part of every image
__global__ __launch_bounds__(256) void channel_sums_kernel(const float* __restrict__ x, long N, int C, long HW,
                                                           long per_slab, double* __restrict__ partials) {
    const int c = blockIdx.y;
    const long e0 = (long)blockIdx.x * per_slab;   // element range [e0, e1) of the N*HW elements of channel c
    long e1 = e0 + per_slab;
    if (e1 > N * HW) e1 = N * HW;
    double s1 = 0.0, s2 = 0.0;
    if ((HW & 3) == 0 && (e0 & 3) == 0) {
        // 16-byte loads; (n, hw) carried incrementally instead of a division per element
        long e = e0 + 4L * threadIdx.x;
        long n = e / HW, hw = e - n * HW;
        for (; e < e1; e += 1024) {
            const f32x4 v = *(const f32x4*)(x + (n * C + c) * HW + hw);
            const int lim = e1 - e < 4 ? (int)(e1 - e) : 4;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < lim) {
                    const double d = (double)v[k];
                    s1 += d;
                    s2 += d * d;
                }
            hw += 1024;
            while (hw >= HW) {
                hw -= HW;
                ++n;
            }
        }
    } else {
        for (long e = e0 + threadIdx.x; e < e1; e += 256) {
            const long n = e / HW, hw = e - n * HW;
            const double v = (double)x[(n * C + c) * HW + hw];
            s1 += v;
            s2 += v * v;
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    __shared__ double r1[4], r2[4];
    if ((threadIdx.x & 63) == 0) {
        r1[threadIdx.x >> 6] = s1;
        r2[threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[((long)blockIdx.x * C + c) * 2 + 0] = r1[0] + r1[1] + r1[2] + r1[3];
        partials[((long)blockIdx.x * C + c) * 2 + 1] = r2[0] + r2[1] + r2[2] + r2[3];
    }
}

__global__ void channel_stats_finalize_kernel(const double* __restrict__ partials, int nslab, int C, double count,
                                              float* __restrict__ mean, float* __restrict__ stdv) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nslab; ++k) {
        s1 += partials[((long)k * C + c) * 2 + 0];
        s2 += partials[((long)k * C + c) * 2 + 1];
    }
    const double m = s1 / count;
    double var = (s2 - s1 * m) / (count - 1.0);   // unbiased, as torch.std_mean
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    stdv[c] = (float)sqrt(var);
}

constexpr int kStatSlabs = 256;

}  // namespace primia

using namespace primia;

extern "C" {

int primia_mixup(const float* x, float* out_x, int64_t L, int64_t per_sample, float lam, float one_minus_lam,
                 primia_stream_t stream) {
    if (L == 0 || per_sample == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(x && out_x && L > 0 && per_sample > 0);
    hipStream_t st = (hipStream_t)stream;
    const int64_t h = L / 2;
    if (h > 0) {
        const long n = h * per_sample;
        const bool al = (((uintptr_t)x | (uintptr_t)out_x | (uintptr_t)(x + n)) & 15) == 0;
        const long n4 = al ? n / 4 : 0;
        long blocks = (n / 4 + 255) / 256;
        if (blocks < 1) blocks = 1;
        if (blocks > 4096) blocks = 4096;
        mixup_kernel<<<(int)blocks, 256, 0, st>>>(x, x + n, out_x, n4, n, lam, one_minus_lam);
    }
    if (L & 1) {  // the trailing sample is passed through (utils.py:389-398)
        if (hipMemcpyAsync(out_x + h * per_sample, x + (L - 1) * per_sample, per_sample * sizeof(float),
                           hipMemcpyDeviceToDevice, st) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
    }
    return launch_status();
}

int primia_to_one_hot(const int64_t* labels, float* out, int64_t n, int classes, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(labels && out && n > 0 && classes > 0);
    one_hot_kernel<<<(unsigned)((n * classes + 255) / 256), 256, 0, (hipStream_t)stream>>>(labels, out, n, classes);
    return launch_status();
}

int64_t primia_channel_stats_workspace_bytes(int C) { return (int64_t)kStatSlabs * C * 2 * sizeof(double); }

int primia_channel_mean_std(const float* x_nchw, int64_t N, int C, int64_t HW, float* mean, float* stdv,
                            void* workspace, int64_t workspace_bytes, primia_stream_t stream) {
    PRIMIA_REQUIRE(x_nchw && mean && stdv && workspace && N > 0 && C > 0 && HW > 0 && N * HW > 1);
    if (workspace_bytes < primia_channel_stats_workspace_bytes(C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long total = N * HW;
    int nslab = (int)((total + 4095) / 4096);
    if (nslab > kStatSlabs) nslab = kStatSlabs;
    long per = (total + nslab - 1) / nslab;
    per = (per + 3) & ~3L;  // slab starts stay 16-byte aligned
    nslab = (int)((total + per - 1) / per);
    channel_sums_kernel<<<dim3(nslab, C), 256, 0, st>>>(x_nchw, N, C, HW, per, (double*)workspace);
    channel_stats_finalize_kernel<<<(C + 63) / 64, 64, 0, st>>>((const double*)workspace, nslab, C, (double)total, mean,
                                                               stdv);
    return launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Image preparation in front of the stem: the deterministic core of create_albu_transform
// (torchlib/dataloader.py:138-217) — a.Resize(R, R) -> a.RandomCrop(S, S) [-> a.VerticalFlip] -> a.ToFloat(255) ->
// a.Normalize(mean, std, max_pixel_value=1.0) — on a decoded uint8 HWC image, one launch per image, writing one
// fp32 [C, S, S] plane set of the client's device-resident dataset.
//   resize: bilinear with half-pixel centres, source coordinates clamped to the image (cv2.INTER_LINEAR, what
//           albumentations.Resize calls); the result is rounded to the nearest uint8 level like cv2's 8-bit
//           path (cv2 evaluates the same weights in 11-bit fixed point: at most one level apart on ties);
//   crop:   window (oy, ox) of the resized image, offsets drawn by the host exactly as albumentations draws them;
//   normalise: (v / 255 - mean[c]) / std[c]; mean == nullptr stops after ToFloat (the statistics pass,
//           torchlib/utils.py:645-666).
// HBM-bound, trivially small: one thread per output pixel, coalesced fp32 stores per channel plane.
// ---------------------------------------------------------------------------------------------------------------
namespace primia {

__global__ __launch_bounds__(256) void image_prepare_kernel(const uint8_t* __restrict__ src, int Hin, int Win, int C,
                                                            int R, int oy, int ox, int S, int flip_v,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ stdv, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * S) return;
    const int y = idx / S, x = idx - y * S;
    const int ry = (flip_v ? S - 1 - y : y) + oy, rx = x + ox;       // pixel of the resized R x R image
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
        float v = floorf(top + (bot - top) * fy + 0.5f);
        v = fminf(fmaxf(v, 0.f), 255.f) / 255.0f;
        if (mean) v = (v - mean[c]) / stdv[c];
        out[(long)c * S * S + idx] = v;
    }
}

}  // namespace primia

extern "C" int primia_image_prepare(const uint8_t* src, int Hin, int Win, int C, int R, int oy, int ox, int S,
                                    int flip_v, const float* mean, const float* stdv, float* out,
                                    primia_stream_t st) {
    PRIMIA_REQUIRE(src && out && Hin > 0 && Win > 0 && (C == 1 || C == 3) && R > 0 && S > 0 && S <= R);
    PRIMIA_REQUIRE(oy >= 0 && ox >= 0 && oy + S <= R && ox + S <= R && ((mean == nullptr) == (stdv == nullptr)));
    primia::image_prepare_kernel<<<primia::ceil_div((long)S * S, 256), 256, 0, (hipStream_t)st>>>(
        src, Hin, Win, C, R, oy, ox, S, flip_v, mean, stdv, out);
    return primia::launch_status();
}
