// Pooling, classifier head and cross-entropy kernels (NHWC).  All are small next to the
// convolutions: the 3x3/2 pools stream the stem activation once with 16-byte accesses, the head
// works on [N, 512] fp32 features.
#include "common.h"

namespace primia {

// ---- 3x3 stride-2 pad-1 pooling ----------------------------------------------------------------
// One thread per (output pixel, 16-byte channel chunk).  Max: window scanned r-major with a strict
// '>' so the FIRST maximum wins (at::max_pool2d semantics); its position r*3+s is kept as one
// byte per element for the backward pass.
template <typename T, bool IS_MAX>
__global__ __launch_bounds__(256) void pool3x3s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            uint8_t* __restrict__ argmax, int N, int H,
                                                            int W, int C, int Ho, int Wo) {
    constexpr int CH = Chunk<T>::N;
    const int cpr = C / CH;
    const long total = (long)N * Ho * Wo * cpr;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int cc = (int)(q % cpr);
    long t = q / cpr;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best[CH];
    int pos[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        best[i] = IS_MAX ? -INFINITY : 0.f;
        pos[i] = -1;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int h = ho * 2 - 1 + r;
        if (h < 0 || h >= H) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int w = wo * 2 - 1 + s;
            if (w < 0 || w >= W) continue;
            float v[CH];
            Chunk<T>::unpack(*(const u32x4*)(x + (((long)n * H + h) * W + w) * C + cc * CH), v);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (IS_MAX) {
                    if (pos[i] < 0 || v[i] > best[i] || v[i] != v[i]) {
                        best[i] = v[i];
                        pos[i] = r * 3 + s;
                    }
                } else {
                    best[i] += v[i];
                }
            }
        }
    }
    if (!IS_MAX) {
#pragma unroll
        for (int i = 0; i < CH; ++i) best[i] *= (1.f / 9.f);  // count_include_pad=True
    }
    const long o = (((long)n * Ho + ho) * Wo + wo) * C + cc * CH;
    *(u32x4*)(y + o) = Chunk<T>::pack(best);
    if (IS_MAX) {
#pragma unroll
        for (int i = 0; i < CH; ++i) argmax[o + i] = (uint8_t)pos[i];
    }
}

// Gather form: each input pixel sums the gradients of the (<= 4) windows that selected it.
template <typename T, bool IS_MAX>
__global__ __launch_bounds__(256) void pool3x3s2_bwd_kernel(const T* __restrict__ dy,
                                                            const uint8_t* __restrict__ argmax,
                                                            T* __restrict__ dx, int N, int H, int W, int C,
                                                            int Ho, int Wo) {
    constexpr int CH = Chunk<T>::N;
    const int cpr = C / CH;
    const long total = (long)N * H * W * cpr;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int cc = (int)(q % cpr);
    long t = q / cpr;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float g[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) g[i] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int th = h + 1 - r;
        if (th < 0 || (th & 1)) continue;
        const int ho = th >> 1;
        if (ho >= Ho) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tw = w + 1 - s;
            if (tw < 0 || (tw & 1)) continue;
            const int wo = tw >> 1;
            if (wo >= Wo) continue;
            const long o = (((long)n * Ho + ho) * Wo + wo) * C + cc * CH;
            float v[CH];
            Chunk<T>::unpack(*(const u32x4*)(dy + o), v);
            if (IS_MAX) {
                // CH bytes of argmax codes
                uint8_t code[CH];
                if (CH == 8) {
                    const u32x2 a = *(const u32x2*)(argmax + o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        code[i] = (uint8_t)(a[0] >> (8 * i));
                        code[4 + i] = (uint8_t)(a[1] >> (8 * i));
                    }
                } else {
                    const uint32_t a = *(const uint32_t*)(argmax + o);
#pragma unroll
                    for (int i = 0; i < 4; ++i) code[i] = (uint8_t)(a >> (8 * i));
                }
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if (code[i] == r * 3 + s) g[i] += v[i];
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) g[i] += v[i] * (1.f / 9.f);
            }
        }
    }
    *(u32x4*)(dx + (((long)n * H + h) * W + w) * C + cc * CH) = Chunk<T>::pack(g);
}

// ---- global average pool: x[N, HW, C] -> feat[N, C] fp32 -----------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ feat,
                                                      int HW, int C) {
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += Elem<T>::load(x + ((long)n * HW + p) * C + c);
        feat[(long)n * C + c] = s / (float)HW;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ dx,
                                                      int HW, int C, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long n = i / ((long)HW * C);
    Elem<T>::store(dx + i, dfeat[n * C + c] / (float)HW);
}

// ---- linear --------------------------------------------------------------------------------------
// One wave per sample; lanes stride the input features.
__global__ __launch_bounds__(64) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y,
                                                        int in_f, int out_f) {
    const int n = blockIdx.x, lane = threadIdx.x;
    for (int j = 0; j < out_f; ++j) {
        float s = 0.f;
        for (int i = lane; i < in_f; i += 64) s += x[(long)n * in_f + i] * w[(long)j * in_f + i];
        s = wave_sum(s);
        if (lane == 0) y[(long)n * out_f + j] = s + (b ? b[j] : 0.f);
    }
}
// dx[n][i] = sum_j dy[n][j] w[j][i]
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ w,
                                                            const float* __restrict__ dy,
                                                            float* __restrict__ dx, int N, int in_f,
                                                            int out_f) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * in_f) return;
    const int f = (int)(i % in_f);
    const long n = i / in_f;
    float s = 0.f;
    for (int j = 0; j < out_f; ++j) s += dy[n * out_f + j] * w[(long)j * in_f + f];
    dx[i] = s;
}
// dw[j][i] = sum_n dy[n][j] x[n][i]; db[j] = sum_n dy[n][j].  Block = (output j, 64 inputs); 16 groups of
// 64 lanes each take every 16th sample and are combined through LDS; column in_f carries the bias.
__global__ __launch_bounds__(1024) void linear_bwd_dw_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ dy,
                                                             float* __restrict__ dw, float* __restrict__ db,
                                                             int N, int in_f, int out_f) {
    __shared__ float red[16][64];
    const int j = blockIdx.y, i = blockIdx.x * 64 + (threadIdx.x & 63), ng = threadIdx.x >> 6;
    float s = 0.f;
    if (i <= in_f)
        for (int n = ng; n < N; n += 16) s += dy[(long)n * out_f + j] * (i < in_f ? x[(long)n * in_f + i] : 1.f);
    red[ng][threadIdx.x & 63] = s;
    __syncthreads();
    if (ng == 0 && i <= in_f) {
        for (int g = 1; g < 16; ++g) s += red[g][threadIdx.x & 63];
        if (i < in_f)
            dw[(long)j * in_f + i] = s;
        else if (db)
            db[j] = s;
    }
}

// ---- head in two launches: AvgPool2d(7) -> flatten -> Linear (torchlib/models.py:400-404, 478-481, 495) ----
// forward : block per sample; 16-byte loads, the pixel rows spread over the block's row groups, per-channel sums
//           combined through LDS -> feat (kept for the weight gradient), then the out_f dot products by the waves.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ feat,
                                                       float* __restrict__ y, int HW, int C, int out_f) {
    constexpr int CH = Chunk<T>::N;
    extern __shared__ float sh[];              // [rpp][C] partial sums, then [C] feat in row 0
    const int tpr = C / CH, rpp = 256 / tpr;
    const int rg = threadIdx.x / tpr, cc = threadIdx.x % tpr;
    const int n = blockIdx.x;
    float s[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) s[i] = 0.f;
    if (rg < rpp)
        for (int p = rg; p < HW; p += rpp) {
            float v[CH];
            Chunk<T>::unpack(*(const u32x4*)(x + ((long)n * HW + p) * C + cc * CH), v);
#pragma unroll
            for (int i = 0; i < CH; ++i) s[i] += v[i];
        }
    if (rg < rpp) {
#pragma unroll
        for (int i = 0; i < CH; ++i) sh[rg * C + cc * CH + i] = s[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int g = 0; g < rpp; ++g) a += sh[g * C + c];
        a /= (float)HW;
        feat[(long)n * C + c] = a;
        sh[rpp * C + c] = a;
    }
    __syncthreads();
    const float* f = sh + rpp * C;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = wave; j < out_f; j += 4) {
        float a = 0.f;
        for (int i = lane; i < C; i += 64) a += f[i] * w[(long)j * C + i];
        a = wave_sum(a);
        if (lane == 0) y[(long)n * out_f + j] = a + (b ? b[j] : 0.f);
    }
}
// backward: dx[n][p][c] = (sum_j dy[n][j] w[j][c]) / HW for every pixel p (linear dx + avg-pool backward)
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                       T* __restrict__ dx, int HW, int C, int out_f) {
    constexpr int CH = Chunk<T>::N;
    extern __shared__ float sh[];              // [C]
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int j = 0; j < out_f; ++j) a += dy[(long)n * out_f + j] * w[(long)j * C + c];
        sh[c] = a / (float)HW;
    }
    __syncthreads();
    const int cpr = C / CH;
    for (int q = threadIdx.x; q < HW * cpr; q += 256) {
        const int c0 = (q % cpr) * CH;
        float v[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = sh[c0 + i];
        *(u32x4*)(dx + (long)n * HW * C + (long)q * CH) = Chunk<T>::pack(v);
    }
}

// ... that also forms the backward sums of the residual BatchNorm whose output gradient it writes (the last block's bn2:
// dx = dz of z = relu(bn(y) + identity)): sum g, sum g * xhat with g = dx AS STORED * mask bit — the gradient is one value per
// (image, channel), so the sums are that value times the count / the xhat sum of the unmasked pixels — as partials
// [N][2][C] (primia_bn_bwd_mask_from_sums consumes them; the 11-us reduction pass over (y, dz) is dropped).
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_bnsums_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                              T* __restrict__ dx, const T* __restrict__ bn_y,
                                                              const uint8_t* __restrict__ relu_mask,
                                                              const float* __restrict__ bn_mean,
                                                              const float* __restrict__ bn_invstd, float* __restrict__ sums,
                                                              int HW, int C, int out_f) {
    constexpr int CH = Chunk<T>::N;
    extern __shared__ float sh[];              // [C] gradient | [256 / cpr][2][C] partial sums
    const int n = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int j = 0; j < out_f; ++j) a += dy[(long)n * out_f + j] * w[(long)j * C + c];
        sh[c] = a / (float)HW;
    }
    __syncthreads();
    const int cpr = C / CH;                    // (256 % cpr == 0: a thread keeps its channel chunk)
    const int c0 = (threadIdx.x % cpr) * CH;
    float v[CH], vr[CH], mu[CH], cnt[CH], sx[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        v[i] = sh[c0 + i];
        mu[i] = bn_mean[c0 + i];
        cnt[i] = sx[i] = 0.f;
    }
    const u32x4 packed = Chunk<T>::pack(v);
    Chunk<T>::unpack(packed, vr);              // the value as stored
    for (int q = threadIdx.x; q < HW * cpr; q += 256) {
        const long off = (long)n * HW * C + (long)q * CH;
        *(u32x4*)(dx + off) = packed;
        float yv[CH];
        Chunk<T>::unpack(*(const u32x4*)(bn_y + off), yv);
        const unsigned m = relu_mask[off / CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bool on = (m >> i) & 1u;
            cnt[i] += on ? 1.f : 0.f;
            sx[i] += on ? yv[i] - mu[i] : 0.f;
        }
    }
    float* red = sh + C;
    const int grp = threadIdx.x / cpr;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        red[(grp * 2 + 0) * C + c0 + i] = vr[i] * cnt[i];
        red[(grp * 2 + 1) * C + c0 + i] = vr[i] * sx[i];
    }
    __syncthreads();
    const int ngrp = 256 / cpr;
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int qq = c / C, cc = c - qq * C;
        float a = 0.f;
        for (int g = 0; g < ngrp; ++g) a += red[(g * 2 + qq) * C + cc];
        if (qq) a *= bn_invstd[cc];
        sums[((long)n * 2 + qq) * C + cc] = a;
    }
}

// ---- cross entropy (single block; N is a batch size) ---------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

constexpr int kMaxClasses = 16;

template <bool SOFT>
__global__ __launch_bounds__(256) void xent_kernel(const float* __restrict__ logits, const void* target,
                                                   const float* __restrict__ cw, float* __restrict__ loss,
                                                   float* __restrict__ dlogits, int N, int C) {
    __shared__ float sh[4];
    // pass 1: weighted loss numerator and (hard) the weight normaliser
    float num = 0.f, den = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* o = logits + (long)n * C;
        float mx = o[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, o[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(o[c] - mx);
        const float lse = mx + logf(se);
        if (SOFT) {
            const float* t = (const float*)target + (long)n * C;
            float a = cw ? 0.f : 1.f, l = 0.f;
            for (int c = 0; c < C; ++c) {
                if (cw) a += cw[c] * t[c];
                l -= t[c] * (o[c] - lse);
            }
            num += a * l;
        } else {
            const int64_t t64 = ((const int64_t*)target)[n];
            if (t64 < 0 || t64 >= C) {      // torch raises here; a kernel cannot: poison the loss instead of reading
                num = __builtin_nanf("");   // out of bounds (and the sample's gradient row below)
                continue;
            }
            const int t = (int)t64;
            const float wt = cw ? cw[t] : 1.f;
            num += wt * (lse - o[t]);
            den += wt;
        }
    }
    num = block_sum_256(num, sh);
    den = SOFT ? (float)N : block_sum_256(den, sh);
    if (threadIdx.x == 0) loss[0] = num / den;
    if (!dlogits) return;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* o = logits + (long)n * C;
        float mx = o[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, o[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(o[c] - mx);
        const float inv = 1.f / se;
        if (SOFT) {
            const float* t = (const float*)target + (long)n * C;
            float a = cw ? 0.f : 1.f, ts = 0.f;
            for (int c = 0; c < C; ++c) {
                if (cw) a += cw[c] * t[c];
                ts += t[c];
            }
            for (int c = 0; c < C; ++c)
                dlogits[(long)n * C + c] = a / den * (expf(o[c] - mx) * inv * ts - t[c]);
        } else {
            const int64_t t64 = ((const int64_t*)target)[n];
            const bool bad = t64 < 0 || t64 >= C;
            const int t = bad ? 0 : (int)t64;
            const float wt = bad ? __builtin_nanf("") : (cw ? cw[t] : 1.f) / den;
            for (int c = 0; c < C; ++c)
                dlogits[(long)n * C + c] = wt * (expf(o[c] - mx) * inv - (c == t ? 1.f : 0.f));
        }
    }
}

template <typename T>
static int pool_launch(bool is_max, bool fwd, const void* a, void* b, uint8_t* argmax, int N, int H, int W,
                       int C, hipStream_t st) {
    constexpr int CH = Chunk<T>::N;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % CH) return PRIMIA_ERR_ARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * (fwd ? (long)Ho * Wo : (long)H * W) * (C / CH);
    const int grid = ceil_div(total, 256);
    if (fwd) {
        if (is_max)
            pool3x3s2_fwd_kernel<T, true><<<grid, 256, 0, st>>>((const T*)a, (T*)b, argmax, N, H, W, C, Ho, Wo);
        else
            pool3x3s2_fwd_kernel<T, false><<<grid, 256, 0, st>>>((const T*)a, (T*)b, nullptr, N, H, W, C, Ho, Wo);
    } else {
        if (is_max)
            pool3x3s2_bwd_kernel<T, true><<<grid, 256, 0, st>>>((const T*)a, argmax, (T*)b, N, H, W, C, Ho, Wo);
        else
            pool3x3s2_bwd_kernel<T, false><<<grid, 256, 0, st>>>((const T*)a, nullptr, (T*)b, N, H, W, C, Ho, Wo);
    }
    return launch_status();
}

}  // namespace primia

using namespace primia;

extern "C" {

int primia_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C,
                            int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(x && y && argmax);
    if (dtype == PRIMIA_F32) return pool_launch<float>(true, true, x, y, argmax, N, H, W, C, (hipStream_t)stream);
    if (dtype == PRIMIA_BF16) return pool_launch<bf16>(true, true, x, y, argmax, N, H, W, C, (hipStream_t)stream);
    return PRIMIA_ERR_ARG;
}
int primia_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int N, int H, int W,
                            int C, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(dy && dx && argmax);
    if (dtype == PRIMIA_F32) return pool_launch<float>(true, false, dy, dx, (uint8_t*)argmax, N, H, W, C, (hipStream_t)stream);
    if (dtype == PRIMIA_BF16) return pool_launch<bf16>(true, false, dy, dx, (uint8_t*)argmax, N, H, W, C, (hipStream_t)stream);
    return PRIMIA_ERR_ARG;
}
int primia_avgpool3x3s2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype,
                            primia_stream_t stream) {
    PRIMIA_REQUIRE(x && y);
    if (dtype == PRIMIA_F32) return pool_launch<float>(false, true, x, y, nullptr, N, H, W, C, (hipStream_t)stream);
    if (dtype == PRIMIA_BF16) return pool_launch<bf16>(false, true, x, y, nullptr, N, H, W, C, (hipStream_t)stream);
    return PRIMIA_ERR_ARG;
}
int primia_avgpool3x3s2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype,
                            primia_stream_t stream) {
    PRIMIA_REQUIRE(dy && dx);
    if (dtype == PRIMIA_F32) return pool_launch<float>(false, false, dy, dx, nullptr, N, H, W, C, (hipStream_t)stream);
    if (dtype == PRIMIA_BF16) return pool_launch<bf16>(false, false, dy, dx, nullptr, N, H, W, C, (hipStream_t)stream);
    return PRIMIA_ERR_ARG;
}

int primia_global_avgpool_fwd(const void* x, float* feat, int N, int HW, int C, int dtype,
                              primia_stream_t stream) {
    PRIMIA_REQUIRE(x && feat && N > 0 && HW > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        gap_fwd_kernel<float><<<N, 256, 0, st>>>((const float*)x, feat, HW, C);
    else if (dtype == PRIMIA_BF16)
        gap_fwd_kernel<bf16><<<N, 256, 0, st>>>((const bf16*)x, feat, HW, C);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}
int primia_global_avgpool_bwd(const float* dfeat, void* dx, int N, int HW, int C, int dtype,
                              primia_stream_t stream) {
    PRIMIA_REQUIRE(dfeat && dx && N > 0 && HW > 0 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)N * HW * C;
    if (dtype == PRIMIA_F32)
        gap_bwd_kernel<float><<<ceil_div(total, 256), 256, 0, st>>>(dfeat, (float*)dx, HW, C, total);
    else if (dtype == PRIMIA_BF16)
        gap_bwd_kernel<bf16><<<ceil_div(total, 256), 256, 0, st>>>(dfeat, (bf16*)dx, HW, C, total);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

static bool head_shape_ok(int C, int dtype) {
    const int ch = dtype == PRIMIA_F32 ? 4 : 8;
    return C % ch == 0 && C / ch <= 256 && 256 % (C / ch) == 0 && C <= 2048;
}
int primia_head_fwd(const void* x, const float* w, const float* b, float* feat, float* logits, int N, int HW, int C,
                    int out_f, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(x && w && feat && logits && N > 0 && HW > 0 && C > 0 && out_f > 0);
    PRIMIA_REQUIRE(dtype == PRIMIA_F32 || dtype == PRIMIA_BF16);
    hipStream_t st = (hipStream_t)stream;
    if (!head_shape_ok(C, dtype)) {   // odd channel counts: the two general kernels
        const int rc = primia_global_avgpool_fwd(x, feat, N, HW, C, dtype, stream);
        return rc ? rc : primia_linear_fwd(feat, w, b, logits, N, C, out_f, stream);
    }
    const int ch = dtype == PRIMIA_F32 ? 4 : 8;
    const size_t lds = (size_t)(256 / (C / ch) + 1) * C * sizeof(float);
    if (dtype == PRIMIA_F32)
        head_fwd_kernel<float><<<N, 256, lds, st>>>((const float*)x, w, b, feat, logits, HW, C, out_f);
    else
        head_fwd_kernel<bf16><<<N, 256, lds, st>>>((const bf16*)x, w, b, feat, logits, HW, C, out_f);
    return launch_status();
}
int primia_head_bwd(const float* w, const float* dlogits, void* dx, int N, int HW, int C, int out_f, int dtype,
                    primia_stream_t stream) {
    PRIMIA_REQUIRE(w && dlogits && dx && N > 0 && HW > 0 && C > 0 && out_f > 0);
    PRIMIA_REQUIRE((dtype == PRIMIA_F32 || dtype == PRIMIA_BF16) && head_shape_ok(C, dtype));
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)C * sizeof(float);
    if (dtype == PRIMIA_F32)
        head_bwd_kernel<float><<<N, 256, lds, st>>>(w, dlogits, (float*)dx, HW, C, out_f);
    else
        head_bwd_kernel<bf16><<<N, 256, lds, st>>>(w, dlogits, (bf16*)dx, HW, C, out_f);
    return launch_status();
}

int primia_head_bwd_bnsums(const float* w, const float* dlogits, void* dx, const void* bn_y, const uint8_t* relu_mask,
                           const float* bn_mean, const float* bn_invstd, float* sums, int N, int HW, int C, int out_f, int dtype,
                           primia_stream_t stream) {
    PRIMIA_REQUIRE(w && dlogits && dx && bn_y && relu_mask && bn_mean && bn_invstd && sums && N > 0 && HW > 0 && C > 0 && out_f > 0);
    PRIMIA_REQUIRE((dtype == PRIMIA_F32 || dtype == PRIMIA_BF16) && head_shape_ok(C, dtype));
    const int cpr = C / (dtype == PRIMIA_F32 ? 4 : 8);
    if (cpr > 256 || 256 % cpr) return PRIMIA_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(1 + 2 * (256 / cpr)) * C * sizeof(float);
    if (lds > 48 * 1024) return PRIMIA_ERR_UNSUPPORTED;
    if (dtype == PRIMIA_F32)
        head_bwd_bnsums_kernel<float><<<N, 256, lds, st>>>(w, dlogits, (float*)dx, (const float*)bn_y, relu_mask, bn_mean,
                                                           bn_invstd, sums, HW, C, out_f);
    else
        head_bwd_bnsums_kernel<bf16><<<N, 256, lds, st>>>(w, dlogits, (bf16*)dx, (const bf16*)bn_y, relu_mask, bn_mean,
                                                          bn_invstd, sums, HW, C, out_f);
    return launch_status();
}

int primia_linear_fwd(const float* x, const float* w, const float* b, float* y, int N, int in_f,
                      int out_f, primia_stream_t stream) {
    PRIMIA_REQUIRE(x && w && y && N > 0 && in_f > 0 && out_f > 0);
    linear_fwd_kernel<<<N, 64, 0, (hipStream_t)stream>>>(x, w, b, y, in_f, out_f);
    return launch_status();
}
int primia_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                      float* db, int N, int in_f, int out_f, primia_stream_t stream) {
    PRIMIA_REQUIRE(x && w && dy && dw && N > 0 && in_f > 0 && out_f > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dx) linear_bwd_dx_kernel<<<ceil_div((long)N * in_f, 256), 256, 0, st>>>(w, dy, dx, N, in_f, out_f);
    linear_bwd_dw_kernel<<<dim3((in_f + 1 + 63) / 64, out_f), 1024, 0, st>>>(x, dy, dw, db, N, in_f, out_f);
    return launch_status();
}

int primia_xent_hard(const float* logits, const int64_t* target, const float* class_weight,
                     float* loss, float* dlogits, int N, int C, primia_stream_t stream) {
    PRIMIA_REQUIRE(logits && target && loss && N > 0 && C > 0);
    xent_kernel<false><<<1, 256, 0, (hipStream_t)stream>>>(logits, target, class_weight, loss, dlogits, N, C);
    return launch_status();
}
int primia_xent_soft(const float* logits, const float* target, const float* class_weight,
                     float* loss, float* dlogits, int N, int C, primia_stream_t stream) {
    PRIMIA_REQUIRE(logits && target && loss && N > 0 && C > 0);
    xent_kernel<true><<<1, 256, 0, (hipStream_t)stream>>>(logits, target, class_weight, loss, dlogits, N, C);
    return launch_status();
}

}  // extern "C"
