// GroupNorm (+ fused ReLU / residual) on NHWC tensors and the per-sample pieces of DP-SGD.
//
// The reference's DP wiring (train.py:304-334: pytorch-dp PrivacyEngine, max_grad_norm 1.0,
// noise_multiplier 1.3) rejects BatchNorm (train.py:308), and ResNet takes a `norm_layer` hook
// (torchlib/models.py:355,362-364, GroupNorm init :411) — so the DP configuration of the build
// (BASELINE.json configs[3]) runs ResNet-18 with GroupNorm(32, C), whose statistics are per sample
// and therefore keep per-sample gradients independent.
//
// Layout: x [N, HW, C]; group g owns channels g*cpg .. g*cpg+cpg-1.  All kernels are HBM streaming
// passes; reductions are two-level (per-slab fp32 partials -> fp64 combine), like csrc/bn.hip.
#include <stdlib.h>

#include "common.h"

namespace primia {

constexpr int kGnSlabs = 32;  // pixel slabs per sample in the reduction passes

// partials[n][slab][q][c], q in {0,1}: two per-element quantities reduced over a slab of pixels
template <typename T, typename F>
__global__ __launch_bounds__(256) void gn_colreduce2_kernel(F f, int HW, int C, int rows_per_slab,
                                                            float* __restrict__ partials) {
    constexpr int CH = Chunk<T>::N;
    const int tpr = C / CH, rpp = 256 / tpr;
    const int rg = threadIdx.x / tpr, cc = threadIdx.x % tpr;
    const int n = blockIdx.y, slab = blockIdx.x;
    const int r0 = slab * rows_per_slab;
    int r1 = r0 + rows_per_slab;
    if (r1 > HW) r1 = HW;
    float s1[CH], s2[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) s1[i] = s2[i] = 0.f;
    F lf = f;
    lf.prepare(n, cc * CH);
    for (int r = r0 + rg; r < r1; r += rpp) lf(((long)n * HW + r) * C + cc * CH, n, cc * CH, s1, s2);
    __shared__ float red[2][256 * CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        red[0][rg * C + cc * CH + i] = s1[i];
        red[1][rg * C + cc * CH + i] = s2[i];
    }
    __syncthreads();
    float* out = partials + ((long)n * gridDim.x + slab) * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int g = 0; g < rpp; ++g) {
            a += red[0][g * C + c];
            b += red[1][g * C + c];
        }
        out[c] = a;
        out[C + c] = b;
    }
}

template <typename T>
struct GnStatsFn {
    const T* y;
    __device__ __forceinline__ void prepare(int, int) {}
    __device__ __forceinline__ void operator()(long off, int, int, float* s1, float* s2) const {
        constexpr int CH = Chunk<T>::N;
        float v[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + off), v);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            s1[i] += v[i];
            s2[i] += v[i] * v[i];
        }
    }
};

// the value gn_apply_kernel stores, before its ReLU (ONE expression for the forward pass and the recomputed masks)
__device__ __forceinline__ float gn_value(float y, float mean, float scale, float shift) {
    return __fmaf_rn(y - mean, scale, shift);
}
// ... and whether the STORED relu of it is positive (bf16: a positive value below half the smallest subnormal stores 0)
template <typename T>
__device__ __forceinline__ bool gn_relu_open(float y, float mean, float scale, float shift) {
    const float v = gn_value(y, mean, scale, shift);
    if constexpr (sizeof(T) == 2) return v > 0.f && f32_to_bf16(v) != 0;
    return v > 0.f;
}

template <typename T>
struct GnBwdFn {
    const T* y;
    const T* z;  // null: no relu, or (gamma_beta set) the mask is recomputed from y
    const T* dz;
    const float* mean;    // [N][G]
    const float* invstd;  // [N][G]
    int G, cpg;
    // z = relu(gn(y)) without a residual: z > 0 exactly where the value gn_apply_kernel rounded was > 0, and that value
    // is a function of y alone — the pass then reads two tensors instead of three
    const float* gamma;   // both set: recompute the mask
    const float* beta;
    const uint8_t* mask;  // or: one byte per 16-byte chunk written by the forward pass (bit i = stored z_i > 0)
    // a thread's sample and channels never change: their group statistics are fetched ONCE (the first version divided
    // by cpg and loaded both statistics per element and row)
    float k_mean[Chunk<T>::N], k_invstd[Chunk<T>::N], k_s[Chunk<T>::N], k_b[Chunk<T>::N];
    __device__ __forceinline__ void prepare(int n, int c0) {
#pragma unroll
        for (int i = 0; i < Chunk<T>::N; ++i) {
            const int g = (c0 + i) / cpg;
            k_mean[i] = mean[n * G + g];
            k_invstd[i] = invstd[n * G + g];
            if (gamma) {
                k_s[i] = k_invstd[i] * gamma[c0 + i];
                k_b[i] = beta[c0 + i];
            }
        }
    }
    __device__ __forceinline__ void operator()(long off, int, int, float* s1, float* s2) const {
        constexpr int CH = Chunk<T>::N;
        float vy[CH], vg[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + off), vy);
        Chunk<T>::unpack(*(const u32x4*)(dz + off), vg);
        if (z) {
            float vz[CH];
            Chunk<T>::unpack(*(const u32x4*)(z + off), vz);
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = vz[i] > 0.f ? vg[i] : 0.f;
        } else if (mask) {
            const unsigned m = mask[off / CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = (m >> i) & 1u ? vg[i] : 0.f;
        } else if (gamma) {
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = gn_relu_open<T>(vy[i], k_mean[i], k_s[i], k_b[i]) ? vg[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float xh = (vy[i] - k_mean[i]) * k_invstd[i];
            s1[i] += vg[i];
            s2[i] += vg[i] * xh;
        }
    }
};

// mode 0: partial (sum, sumsq) -> mean / invstd per (n, g).
// One thread per (sample, channel): the slab sums of a channel are read coalesced, the C/G (2..16, a power of
// two) channels of a group sit in adjacent lanes and are combined with xor-shuffles in a fixed order.
// (One thread per (sample, group) walking slabs x channels with strided reads took ~48 us per call, 40 calls
// per DP-SGD step.)
__device__ __forceinline__ double group_sum(double v, int cpg) {
    for (int o = 1; o < cpg; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void gn_stats_finalize_kernel(const float* __restrict__ partials, int nslab, int C,
                                                                int G, long count, float eps, float* __restrict__ mean,
                                                                float* __restrict__ invstd, int NC) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // NC is a multiple of 64: whole waves stay active for the shuffles
    if (i >= NC) return;
    const int n = i / C, c = i - n * C, cpg = C / G;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < nslab; ++s) {
        const float* p = partials + ((long)n * nslab + s) * 2 * C;
        a += (double)p[c];
        b += (double)p[C + c];
    }
    a = group_sum(a, cpg);
    b = group_sum(b, cpg);
    if (c % cpg == 0) {
        const double m = a / (double)count;
        double var = b / (double)count - m * m;
        if (var < 0.0) var = 0.0;
        mean[n * G + c / cpg] = (float)m;
        invstd[n * G + c / cpg] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// mode 1: partial (sum g, sum g*xhat) per (n, c) -> per-sample dbeta/dgamma [N][C] and the group sums
// A[n,g] = sum_c gamma_c * s1, B[n,g] = sum_c gamma_c * s2 used by the input gradient.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partials, int nslab, int C,
                                                              int G, const float* __restrict__ gamma,
                                                              float* __restrict__ ps_dbeta, float* __restrict__ ps_dgamma,
                                                              float* __restrict__ gA, float* __restrict__ gB, int NC) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    const int n = i / C, c = i - n * C, cpg = C / G;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < nslab; ++s) {
        const float* p = partials + ((long)n * nslab + s) * 2 * C;
        a += (double)p[c];
        b += (double)p[C + c];
    }
    ps_dbeta[(long)n * C + c] = (float)a;
    ps_dgamma[(long)n * C + c] = (float)b;
    const double A = group_sum((double)gamma[c] * a, cpg), B = group_sum((double)gamma[c] * b, cpg);
    if (c % cpg == 0) {
        gA[n * G + c / cpg] = (float)A;
        gB[n * G + c / cpg] = (float)B;
    }
}

// The same two reductions with ONE 1024-thread block per sample and the finalize folded in (batches that fill the chip
// with one block per sample: N >= kGnSampleBlockMinN).  The statistics of GroupNorm never leave a sample, so nothing
// has to cross blocks: the slab partials, their table in HBM and the finalize launch (10-12 us each, 40 per DP-SGD
// step) all go.  Row groups are added in a fixed order in fp64, like the slab path's second level.
//   MODE 0: (sum, sumsq) -> mean / invstd [N][G]
//   MODE 1: (sum g, sum g*xhat) -> ps_dbeta / ps_dgamma [N][C] and the group sums gA / gB [N][G]
constexpr int kGnSampleBlockMinN = 128;

template <typename T, typename F, int MODE>
__global__ __launch_bounds__(1024) void gn_sample_reduce_kernel(F f, int HW, int C, int G, double count, float eps,
                                                                const float* __restrict__ gamma, float* __restrict__ o0,
                                                                float* __restrict__ o1, float* __restrict__ o2,
                                                                float* __restrict__ o3) {
    constexpr int CH = Chunk<T>::N;
    const int tpr = C / CH, rpp = 1024 / tpr;
    const int rg = threadIdx.x / tpr, cc = threadIdx.x % tpr;
    const int n = blockIdx.x;
    float s1[CH], s2[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) s1[i] = s2[i] = 0.f;
    F lf = f;
    lf.prepare(n, cc * CH);
    const long base = (long)n * HW * C + cc * CH;
    // rows per trip = loads in flight per thread: a block is alone on its CU, and a tensor that does not come out of the
    // Infinity Cache (the stem's 411 MB) wants ~128 KB in flight per CU (statistics pass: one load per row -> 8 rows)
    if constexpr (MODE == 0) {
#pragma unroll 8
        for (int r = rg; r < HW; r += rpp) lf(base + (long)r * C, n, cc * CH, s1, s2);
    } else {
#pragma unroll 4
        for (int r = rg; r < HW; r += rpp) lf(base + (long)r * C, n, cc * CH, s1, s2);
    }
    __shared__ float red[2][1024 * CH];       // [q][row group][channel]
    __shared__ double chs[2][512];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        red[0][rg * C + cc * CH + i] = s1[i];
        red[1][rg * C + cc * CH + i] = s2[i];
    }
    __syncthreads();
    // 2C (quantity, channel) columns over the 1024 threads: thread -> (column, part of the row groups)
    const int parts = 1024 / (2 * C);                       // 1 (C = 512) .. 8 (C = 64)
    const int col = threadIdx.x % (2 * C), part = threadIdx.x / (2 * C);
    const int q = col / C, c = col - q * C;
    double a = 0.0;
    if (part < parts) {
        const int per = rpp / parts;                       // rpp and parts are powers of two, rpp >= parts
        for (int g = part * per; g < (part + 1) * per; ++g) a += (double)red[q][g * C + c];
    }
    __syncthreads();
    double* dred = (double*)&red[0][0];                     // [part][2C]
    if (part < parts) dred[part * 2 * C + col] = a;
    __syncthreads();
    if (threadIdx.x < 2 * C) {
        double t = 0.0;
        for (int k = 0; k < parts; ++k) t += dred[k * 2 * C + threadIdx.x];
        if (MODE == 1) {
            (q == 0 ? o0 : o1)[(long)n * C + c] = (float)t;   // ps_dbeta | ps_dgamma
            t *= (double)gamma[c];
        }
        chs[q][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int cpg = C / G;
        double A = 0.0, B = 0.0;
        for (int k = 0; k < cpg; ++k) {
            A += chs[0][threadIdx.x * cpg + k];
            B += chs[1][threadIdx.x * cpg + k];
        }
        if (MODE == 0) {
            const double m = A / count;
            double var = B / count - m * m;
            if (var < 0.0) var = 0.0;
            o0[n * G + threadIdx.x] = (float)m;
            o1[n * G + threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
        } else {
            o2[n * G + threadIdx.x] = (float)A;
            o3[n * G + threadIdx.x] = (float)B;
        }
    }
}

// z = act((y - mean[n,g]) * invstd[n,g] * gamma_c + beta_c [+ residual])
// Grid (blocks per sample, N): blockIdx.y is the sample, and because the thread stride is a multiple of the chunks per
// pixel a thread's 8 (4) channels never change — its per-channel factors are fetched once.  (The first version decoded
// sample and channel of every chunk with 64-bit divisions and loaded four per-channel values per ELEMENT: 15.8 M vector
// instructions per launch against 4.2 M of the BatchNorm apply pass over the same bytes.)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                       T* __restrict__ z, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, int HW, int C, int G, int relu,
                                                       uint8_t* __restrict__ mask_out = nullptr) {
    constexpr int CH = Chunk<T>::N;
    const int cpr = C / CH, cpg = C / G;
    const int n = blockIdx.y;
    const int c0 = (int)(threadIdx.x % cpr) * CH;       // (256 and the block stride are multiples of cpr)
    float km[CH], ks[CH], kb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int g = (c0 + i) / cpg;
        km[i] = mean[n * G + g];
        ks[i] = invstd[n * G + g] * gamma[c0 + i];
        kb[i] = beta[c0 + i];
    }
    const int cps = HW * cpr;                            // chunks per sample
    const long base = (long)n * cps;
    const int stride = gridDim.x * 256;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < cps; q += stride) {
        float v[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + (base + q) * CH), v);
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = gn_value(v[i], km[i], ks[i], kb[i]);
        if (res) {
            float r[CH];
            Chunk<T>::unpack(*(const u32x4*)(res + (base + q) * CH), r);
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] += r[i];
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        *(u32x4*)(z + (base + q) * CH) = Chunk<T>::pack(v);
        if (mask_out) {   // bit i = (stored z_i > 0): the backward passes of a residual layer read it instead of z
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                bool open = v[i] > 0.f;
                if constexpr (sizeof(T) == 2) open = open && f32_to_bf16(v[i]) != 0;
                m |= (open ? 1u : 0u) << i;
            }
            mask_out[base + q] = (uint8_t)m;
        }
    }
}

// dy = invstd * (gamma_c*g - A/m - xhat*B/m); optionally g_out = masked g.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ y, const T* __restrict__ z, const T* dz,
                                                           T* __restrict__ dy, T* g_out, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta_mask,   // set: mask from y
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gA, const float* __restrict__ gB,
                                                           float inv_m, int HW, int C, int G,
                                                           const uint8_t* __restrict__ mask = nullptr) {
    constexpr int CH = Chunk<T>::N;
    const int cpr = C / CH, cpg = C / G;
    const int n = blockIdx.y;
    const int c0 = (int)(threadIdx.x % cpr) * CH;
    float km[CH], ki[CH], kg[CH], ka[CH], kbb[CH], kbeta[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int ng = n * G + (c0 + i) / cpg;
        km[i] = mean[ng];
        ki[i] = invstd[ng];
        kg[i] = gamma[c0 + i];
        ka[i] = gA[ng] * inv_m;
        kbb[i] = gB[ng] * inv_m;
        kbeta[i] = beta_mask ? beta_mask[c0 + i] : 0.f;
    }
    const int cps = HW * cpr;
    const long base = (long)n * cps;
    const int stride = gridDim.x * 256;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < cps; q += stride) {
        float vy[CH], vg[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + (base + q) * CH), vy);
        Chunk<T>::unpack(*(const u32x4*)(dz + (base + q) * CH), vg);
        if (z) {
            float vz[CH];
            Chunk<T>::unpack(*(const u32x4*)(z + (base + q) * CH), vz);
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = vz[i] > 0.f ? vg[i] : 0.f;
        } else if (mask) {
            const unsigned m = mask[base + q];
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = (m >> i) & 1u ? vg[i] : 0.f;
        } else if (beta_mask) {
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = gn_relu_open<T>(vy[i], km[i], ki[i] * kg[i], kbeta[i]) ? vg[i] : 0.f;
        }
        if (g_out) *(u32x4*)(g_out + (base + q) * CH) = Chunk<T>::pack(vg);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float xh = (vy[i] - km[i]) * ki[i];
            vy[i] = ki[i] * (kg[i] * vg[i] - ka[i] - xh * kbb[i]);
        }
        *(u32x4*)(dy + (base + q) * CH) = Chunk<T>::pack(vy);
    }
}

// ---- stem tail under GroupNorm: gn1 -> relu -> maxpool(3, 2, 1) without the full-resolution z / dz tensors ----------
// (the BatchNorm path's fusion, csrc/bn.hip bn_relu_pool_*; the forward apply pass IS that kernel, in its group mode)
// Backward reductions at POOLED resolution, per sample: a window passes its gradient to one input element, and where
// ReLU was active xhat of that element is recoverable from the pooled value, xhat = (p - beta) / gamma.
template <typename T>
struct GnPoolScatterFn {
    const T* p;
    const T* dp;
    const uint8_t* argmax;
    const T* y;
    const float* mean;     // [N][G]
    const float* invstd;
    const float* gamma;
    const float* beta;
    int H, W, C, Ho, Wo, G;
    float k_beta[Chunk<T>::N], k_rgamma[Chunk<T>::N], k_mean[Chunk<T>::N], k_invstd[Chunk<T>::N];
    bool any_zero_gamma;
    __device__ __forceinline__ void prepare(int n, int c0) {
        any_zero_gamma = false;
        const int cpg = C / G;
#pragma unroll
        for (int i = 0; i < Chunk<T>::N; ++i) {
            const float g = gamma[c0 + i];
            k_beta[i] = beta[c0 + i];
            const bool rec = pool_xhat_recoverable(g, k_beta[i]);
            k_rgamma[i] = rec ? 1.f / g : 0.f;
            k_mean[i] = mean[n * G + (c0 + i) / cpg];
            k_invstd[i] = invstd[n * G + (c0 + i) / cpg];
            any_zero_gamma |= !rec;
        }
    }
    __device__ __forceinline__ void operator()(long off, int n, int c0, float* s1, float* s2) const {
        constexpr int CH = Chunk<T>::N;
        float vp[CH], vd[CH];
        Chunk<T>::unpack(*(const u32x4*)(p + off), vp);
        Chunk<T>::unpack(*(const u32x4*)(dp + off), vd);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float g = vp[i] > 0.f ? vd[i] : 0.f;
            s1[i] += g;
            s2[i] += g * ((vp[i] - k_beta[i]) * k_rgamma[i]);
        }
        if (any_zero_gamma) {  // rare: xhat of a channel whose gamma is 0 or tiny against beta, from y at the argmax position
            const long row = (off - c0) / C - (long)n * Ho * Wo;
            const int ho = (int)(row / Wo), wo = (int)(row - (long)ho * Wo);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (k_rgamma[i] != 0.f || !(vp[i] > 0.f)) continue;
                const int code = argmax[off + i];
                const int h = 2 * ho - 1 + code / 3, w = 2 * wo - 1 + code % 3;
                const float yv = Elem<T>::load(y + (((long)n * H + h) * W + w) * C + c0 + i);
                s2[i] += vd[i] * ((yv - k_mean[i]) * k_invstd[i]);
            }
        }
    }
};

// Apply pass (even H, W): one thread per 2 x 2 block of input pixels x one 16-byte channel chunk.  Input pixel
// (2a + i, 2b + j) can only be the argmax of the windows (a + di, b + dj), di <= i, dj <= j, at tap
// (1 + i - 2 di, 1 + j - 2 dj): the four windows are loaded once.  dy = invstd (gamma g - A/m - xhat B/m).
template <typename T>
__global__ __launch_bounds__(256) void gn_relu_pool_bwd_apply2x2_kernel(
    const T* __restrict__ y, const T* __restrict__ dp, const uint8_t* __restrict__ argmax, T* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gA, const float* __restrict__ gB, float inv_m, int N, int H,
    int W, int C, int G, int Ho, int Wo) {
    constexpr int CH = Chunk<T>::N;
    const int cpr = C / CH, H2 = H >> 1, W2 = W >> 1, cpg = C / G;
    const long total = (long)N * H2 * W2 * cpr;
    // a block that lies inside ONE sample (nearly all) stages the sample's per-channel constants in LDS once; each thread
    // fetched its 8 x 6 constants from global memory itself before its first data load (282 us for a pass whose bytes are
    // worth ~190)
    __shared__ float sk[6][512];
    bool staged = false;
    {
        const long per_n = (long)H2 * W2 * cpr;
        const long q0 = (long)blockIdx.x * 256;
        long q1 = q0 + 255;
        if (q1 >= total) q1 = total - 1;
        const int n0 = (int)(q0 / per_n);
        if (q0 < total && n0 == (int)(q1 / per_n) && C <= 512) {
            staged = true;
            for (int c = threadIdx.x; c < C; c += 256) {
                const int ng = n0 * G + c / cpg;
                sk[0][c] = mean[ng];
                sk[1][c] = invstd[ng];
                sk[2][c] = gamma[c];
                sk[3][c] = beta[c];
                sk[4][c] = gA[ng] * inv_m;
                sk[5][c] = gB[ng] * inv_m;
            }
        }
    }
    __syncthreads();
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int c0 = (int)(q % cpr) * CH;
    long t = q / cpr;
    const int b = (int)(t % W2);
    t /= W2;
    const int a = (int)(t % H2), n = (int)(t / H2);
    float km[CH], ki[CH], ks[CH], kg[CH], kb[CH], ka[CH], kbb[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        if (staged) {
            km[k] = sk[0][c0 + k];
            ki[k] = sk[1][c0 + k];
            kg[k] = sk[2][c0 + k];
            kb[k] = sk[3][c0 + k];
            ka[k] = sk[4][c0 + k];
            kbb[k] = sk[5][c0 + k];
        } else {
            const int ng = n * G + (c0 + k) / cpg;
            km[k] = mean[ng];
            ki[k] = invstd[ng];
            kg[k] = gamma[c0 + k];
            kb[k] = beta[c0 + k];
            ka[k] = gA[ng] * inv_m;
            kbb[k] = gB[ng] * inv_m;
        }
        ks[k] = ki[k] * kg[k];
    }
    u32x4 wraw[2][2];      // the four windows, packed: gradient chunk and argmax codes (invalid: 0xff never matches)
    u32x2 craw[2][2];
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const bool ok = a + di < Ho && b + dj < Wo;
            const long o = (((long)n * Ho + (ok ? a + di : 0)) * Wo + (ok ? b + dj : 0)) * C + c0;
            wraw[di][dj] = *(const u32x4*)(dp + o);
            if (CH == 8) {
                craw[di][dj] = *(const u32x2*)(argmax + o);
            } else {
                craw[di][dj][0] = *(const uint32_t*)(argmax + o);
                craw[di][dj][1] = 0xffffffffu;
            }
            if (!ok) craw[di][dj] = u32x2{0xffffffffu, 0xffffffffu};
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long off = ((((long)n * H + 2 * a + i) * W) + 2 * b + j) * C + c0;
            float vy[CH], g[CH];
            Chunk<T>::unpack(*(const u32x4*)(y + off), vy);
#pragma unroll
            for (int k = 0; k < CH; ++k) g[k] = 0.f;
#pragma unroll
            for (int di = i; di >= 0; --di)
#pragma unroll
                for (int dj = j; dj >= 0; --dj) {
                    const unsigned want = (unsigned)((1 + i - 2 * di) * 3 + (1 + j - 2 * dj));
                    float wv[CH];
                    Chunk<T>::unpack(wraw[di][dj], wv);
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const unsigned code = (craw[di][dj][k >> 2] >> (8 * (k & 3))) & 0xffu;
                        g[k] += code == want ? wv[k] : 0.f;
                    }
                }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const float gi = gn_relu_open<T>(vy[k], km[k], ks[k], kb[k]) ? g[k] : 0.f;
                const float xh = (vy[k] - km[k]) * ki[k];
                vy[k] = ki[k] * (kg[k] * gi - ka[k] - xh * kbb[k]);
            }
            *(u32x4*)(dy + off) = Chunk<T>::pack(vy);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
}

// ---- DP-SGD per-sample pieces -------------------------------------------------------------------------
// out[n] += sum_j x[n][j]^2   (block per (sample, chunk); fp64 block sum, one atomic per block)
__global__ __launch_bounds__(256) void persample_sqnorm_kernel(const float* __restrict__ x, long per, double* out) {
    const int n = blockIdx.y;
    const float* p = x + (long)n * per;
    double s = 0.0;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < per; j += (long)gridDim.x * 256) {
        const double v = (double)p[j];
        s += v * v;
    }
    s = wave_sum(s);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + n, sh[0] + sh[1] + sh[2] + sh[3]);
}

// the same for `count` small tensors x_k [N][per_k] in one launch (blockIdx.x = k)
__global__ __launch_bounds__(256) void persample_sqnorm_many_kernel(const float* const* __restrict__ xs,
                                                                    const int* __restrict__ pers, double* out) {
    const int n = blockIdx.y, per = pers[blockIdx.x];
    const float* p = xs[blockIdx.x] + (long)n * per;
    double s = 0.0;
    for (int j = threadIdx.x; j < per; j += 256) {
        const double v = (double)p[j];
        s += v * v;
    }
    s = wave_sum(s);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + n, sh[0] + sh[1] + sh[2] + sh[3]);
}

// clip[n] = min(1, C / (sqrt(sq[n]) + 1e-6))   (pytorch-dp's per-sample clip factor)
__global__ void dp_clip_factor_kernel(const double* __restrict__ sq, float* __restrict__ clip, int N, float max_norm) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double f = (double)max_norm / (sqrt(sq[n]) + 1e-6);
    clip[n] = (float)(f < 1.0 ? f : 1.0);
}

// x[row][:] *= s[row / rows_per_sample]
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_kernel(T* x, const float* __restrict__ s, long nchunks, long chunks_per_sample) {
    constexpr int CH = Chunk<T>::N;
    // grid (blocks per sample, N): the sample is blockIdx.y (no 64-bit division per chunk)
    (void)nchunks;
    const float a = s[blockIdx.y];
    const long base = (long)blockIdx.y * chunks_per_sample;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < chunks_per_sample; q += stride) {
        float v[CH];
        Chunk<T>::unpack(*(const u32x4*)(x + (base + q) * CH), v);
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] *= a;
        *(u32x4*)(x + (base + q) * CH) = Chunk<T>::pack(v);
    }
}

// ... of up to 16 tensors in one launch (blockIdx.z = tensor): the DP-SGD batch pass scales the dy of every layer whose
// per-sample tiles are not kept — a dozen launches of 7-19 us each, none of which fills the chip for long
constexpr int kScaleMany = 16;
struct ScaleManyArgs {
    void* x[kScaleMany];
    long cps[kScaleMany];
};
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_many_kernel(ScaleManyArgs a, const float* __restrict__ s) {
    constexpr int CH = Chunk<T>::N;
    T* x = (T*)a.x[blockIdx.z];
    const long cps = a.cps[blockIdx.z];
    const float f = s[blockIdx.y];
    const long base = (long)blockIdx.y * cps;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < cps; q += stride) {
        float v[CH];
        Chunk<T>::unpack(*(const u32x4*)(x + (base + q) * CH), v);
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] *= f;
        *(u32x4*)(x + (base + q) * CH) = Chunk<T>::pack(v);
    }
}

// out[c] = sum_n w[n] * x[n][c]
__global__ __launch_bounds__(256) void weighted_colsum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ out, int N, int C) {
    // block = 16 columns x 16 row slices (a single thread per column walking all N rows took 95 us per call,
    // 42 calls per DP-SGD step); slices are combined in a fixed order: deterministic
    __shared__ double part[16][17];
    const int cl = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < C)
        for (int n = rs; n < N; n += 16) s += (double)w[n] * (double)x[(long)n * C + c];
    part[rs][cl] = s;
    __syncthreads();
    if (rs == 0 && c < C) {
        for (int k = 1; k < 16; ++k) s += part[k][cl];
        out[c] = (float)s;
    }
}

// the same for `count` (x, out, C) jobs in one launch (blockIdx.y = job): the 42 tiny launches of a DP-SGD step
__global__ __launch_bounds__(256) void weighted_colsum_many_kernel(const float* const* __restrict__ xs,
                                                                   const float* __restrict__ w,
                                                                   float* const* __restrict__ outs,
                                                                   const int* __restrict__ Cs, int N) {
    __shared__ double part[16][17];
    const int job = blockIdx.y, C = Cs[job];
    if ((int)blockIdx.x * 16 >= C) return;
    const float* __restrict__ x = xs[job];
    const int cl = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < C)
        for (int n = rs; n < N; n += 16) s += (double)w[n] * (double)x[(long)n * C + c];
    part[rs][cl] = s;
    __syncthreads();
    if (rs == 0 && c < C) {
        for (int k = 1; k < 16; ++k) s += part[k][cl];
        outs[job][c] = (float)s;
    }
}

// ps[n] = [ outer(dy[n], x[n]) (out_f x in_f) | dy[n] (out_f) ]: per-sample fc gradients
__global__ __launch_bounds__(256) void fc_persample_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ ps, int N, int in_f, int out_f) {
    const long per = (long)out_f * in_f + out_f;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * per) return;
    const long n = i / per, j = i - n * per;
    if (j < (long)out_f * in_f)
        ps[i] = dy[n * out_f + j / in_f] * x[n * in_f + j % in_f];
    else
        ps[i] = dy[n * out_f + (j - (long)out_f * in_f)];
}

// g = (g + noise * sigma) * inv_b
__global__ __launch_bounds__(256) void dp_noise_kernel(float* __restrict__ g, const float* __restrict__ noise, long n,
                                                       float sigma, float inv_b) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) g[i] = (g[i] + noise[i] * sigma) * inv_b;
}

static inline int gn_stream_blocks(long nchunks) {
    long b = (nchunks + 255) / 256;
    return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}
// (blocks per sample, N): ~2048 blocks of 256 threads over the launch
static inline dim3 gn_sample_grid(int N, int HW, int cpr) {
    long per = ((long)HW * cpr + 255) / 256;     // blocks that cover one sample once
    long want = (2048 + N - 1) / N;
    if (want < 1) want = 1;
    if (per > want) per = want;
    if (per < 1) per = 1;
    return dim3((unsigned)per, (unsigned)N);
}
static inline dim3 gn_rows_grid(int N, long cps) {
    long per = (cps + 255) / 256, want = (2048 + N - 1) / N;
    if (per > want) per = want;
    if (per < 1) per = 1;
    return dim3((unsigned)per, (unsigned)N);
}
static inline bool gn_shape_ok(int N, int HW, int C, int G, int dtype) {
    const int ch = dtype == PRIMIA_F32 ? 4 : 8;
    const int cpg = (G > 0 && C % G == 0) ? C / G : 0;  // channels per group: a power of two <= 64 (finalize shuffles)
    return N > 0 && HW > 0 && C > 0 && C <= 512 && cpg > 0 && cpg <= 64 && (cpg & (cpg - 1)) == 0 && C % ch == 0 &&
           256 % (C / ch) == 0;
}

// one block per sample (gn_sample_reduce_kernel): the batch fills the chip that way and 2C columns fit the block
static inline bool gn_sample_blocks(int N, int C, int ch) {
    const int sw = PRIMIA_OPT(gn_sample);
    const int tpr = C / ch;
    return sw && N >= kGnSampleBlockMinN && 2 * C <= 1024 && 1024 % tpr == 0 && 1024 / tpr >= 1024 / (2 * C);
}

template <typename T>
static void gn_stats(const void* y, float* mean, float* invstd, int N, int HW, int C, int G, float eps, float* partials,
                     hipStream_t st) {
    const int rps = (HW + kGnSlabs - 1) / kGnSlabs;
    const int nslab = (HW + rps - 1) / rps;
    GnStatsFn<T> f{(const T*)y};
    if (gn_sample_blocks(N, C, Chunk<T>::N)) {
        gn_sample_reduce_kernel<T, GnStatsFn<T>, 0><<<N, 1024, 0, st>>>(f, HW, C, G, (double)HW * (C / G), eps, nullptr,
                                                                        mean, invstd, nullptr, nullptr);
    } else {
        gn_colreduce2_kernel<T, GnStatsFn<T>><<<dim3(nslab, N), 256, 0, st>>>(f, HW, C, rps, partials);
        gn_stats_finalize_kernel<<<(N * C + 255) / 256, 256, 0, st>>>(partials, nslab, C, G, (long)HW * (C / G), eps, mean,
                                                                      invstd, N * C);
    }
}

template <typename T>
static int gn_fwd_impl(const void* y, const void* res, void* z, const float* gamma, const float* beta, float* mean,
                       float* invstd, int N, int HW, int C, int G, float eps, int relu, float* partials, hipStream_t st,
                       uint8_t* mask_out = nullptr) {
    gn_stats<T>(y, mean, invstd, N, HW, C, G, eps, partials, st);
    gn_apply_kernel<T><<<gn_sample_grid(N, HW, C / Chunk<T>::N), 256, 0, st>>>((const T*)y, (const T*)res, (T*)z, gamma, beta,
                                                                               mean, invstd, HW, C, G, relu, mask_out);
    return launch_status();
}

template <typename T>
static int gn_bwd_impl(const void* y, const void* z, const void* dz, void* dy, void* g_out, const float* gamma,
                       const float* mean, const float* invstd, float* ps_dgamma, float* ps_dbeta, int N, int HW, int C,
                       int G, int relu, float* partials, hipStream_t st, const float* beta_mask = nullptr,
                       const uint8_t* mask = nullptr) {
    const int rps = (HW + kGnSlabs - 1) / kGnSlabs;
    const int nslab = (HW + rps - 1) / rps;
    GnBwdFn<T> f{(const T*)y, relu ? (const T*)z : nullptr, (const T*)dz, mean, invstd, G, C / G,
                 beta_mask ? gamma : nullptr, beta_mask, mask, {}, {}, {}, {}};
    float* gA = partials + (long)N * nslab * 2 * C;  // group sums live behind the partials
    float* gB = gA + (long)N * G;
    if (gn_sample_blocks(N, C, Chunk<T>::N)) {
        gn_sample_reduce_kernel<T, GnBwdFn<T>, 1><<<N, 1024, 0, st>>>(f, HW, C, G, 0.0, 0.f, gamma, ps_dbeta, ps_dgamma, gA,
                                                                      gB);
    } else {
        gn_colreduce2_kernel<T, GnBwdFn<T>><<<dim3(nslab, N), 256, 0, st>>>(f, HW, C, rps, partials);
        gn_bwd_finalize_kernel<<<(N * C + 255) / 256, 256, 0, st>>>(partials, nslab, C, G, gamma, ps_dbeta, ps_dgamma, gA,
                                                                    gB, N * C);
    }
    gn_bwd_apply_kernel<T><<<gn_sample_grid(N, HW, C / Chunk<T>::N), 256, 0, st>>>(
        (const T*)y, relu ? (const T*)z : nullptr, (const T*)dz, (T*)dy, (T*)g_out, gamma, beta_mask, mean, invstd, gA, gB,
        (float)(1.0 / ((double)HW * (C / G))), HW, C, G, mask);
    return launch_status();
}

template <typename T>
static int gn_relu_pool_bwd_impl(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax, void* dy,
                                 const float* gamma, const float* beta, const float* mean, const float* invstd,
                                 float* ps_dgamma, float* ps_dbeta, int N, int H, int W, int C, int G, float* partials,
                                 hipStream_t st) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, HWp = Ho * Wo;
    const int rps = (HWp + kGnSlabs - 1) / kGnSlabs;
    const int nslab = (HWp + rps - 1) / rps;
    GnPoolScatterFn<T> f{(const T*)pooled, (const T*)dpooled, argmax, (const T*)y, mean, invstd, gamma, beta,
                         H, W, C, Ho, Wo, G, {}, {}, {}, {}, false};
    float* gA = partials + (long)N * nslab * 2 * C;
    float* gB = gA + (long)N * G;
    if (gn_sample_blocks(N, C, Chunk<T>::N)) {
        gn_sample_reduce_kernel<T, GnPoolScatterFn<T>, 1><<<N, 1024, 0, st>>>(f, HWp, C, G, 0.0, 0.f, gamma, ps_dbeta,
                                                                              ps_dgamma, gA, gB);
    } else {
        gn_colreduce2_kernel<T, GnPoolScatterFn<T>><<<dim3(nslab, N), 256, 0, st>>>(f, HWp, C, rps, partials);
        gn_bwd_finalize_kernel<<<(N * C + 255) / 256, 256, 0, st>>>(partials, nslab, C, G, gamma, ps_dbeta, ps_dgamma, gA,
                                                                    gB, N * C);
    }
    const long total = (long)N * (H / 2) * (W / 2) * (C / Chunk<T>::N);
    gn_relu_pool_bwd_apply2x2_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
        (const T*)y, (const T*)dpooled, argmax, (T*)dy, gamma, beta, mean, invstd, gA, gB,
        (float)(1.0 / ((double)H * W * (C / G))), N, H, W, C, G, Ho, Wo);
    return launch_status();
}

}  // namespace primia

using namespace primia;

extern "C" {

int64_t primia_gn_workspace_bytes(int N, int C, int G) {
    return ((int64_t)N * kGnSlabs * 2 * C + 2 * (int64_t)N * G) * sizeof(float);
}

int primia_gn_fwd(const void* y, const void* residual, void* z, const float* gamma, const float* beta,
                  float* save_mean, float* save_invstd, int N, int HW, int C, int G, float eps, int relu,
                  void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && z && gamma && beta && save_mean && save_invstd && workspace);
    PRIMIA_REQUIRE(gn_shape_ok(N, HW, C, G, dtype));
    if (workspace_bytes < primia_gn_workspace_bytes(N, C, G)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return gn_fwd_impl<float>(y, residual, z, gamma, beta, save_mean, save_invstd, N, HW, C, G, eps, relu,
                                  (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return gn_fwd_impl<bf16>(y, residual, z, gamma, beta, save_mean, save_invstd, N, HW, C, G, eps, relu,
                                 (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

int primia_gn_bwd(const void* y, const void* z, const void* dz, void* dy, void* g_out, const float* gamma,
                  const float* save_mean, const float* save_invstd, float* ps_dgamma, float* ps_dbeta, int N, int HW,
                  int C, int G, int relu, void* workspace, int64_t workspace_bytes, int dtype,
                  primia_stream_t stream) {
    PRIMIA_REQUIRE(y && dz && dy && gamma && save_mean && save_invstd && ps_dgamma && ps_dbeta && workspace);
    PRIMIA_REQUIRE(!relu || z);
    PRIMIA_REQUIRE(gn_shape_ok(N, HW, C, G, dtype));
    if (workspace_bytes < primia_gn_workspace_bytes(N, C, G)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return gn_bwd_impl<float>(y, z, dz, dy, g_out, gamma, save_mean, save_invstd, ps_dgamma, ps_dbeta, N, HW, C, G,
                                  relu, (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return gn_bwd_impl<bf16>(y, z, dz, dy, g_out, gamma, save_mean, save_invstd, ps_dgamma, ps_dbeta, N, HW, C, G,
                                 relu, (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

int primia_gn_fwd_mask(const void* y, const void* residual, void* z, uint8_t* relu_mask, const float* gamma,
                       const float* beta, float* save_mean, float* save_invstd, int N, int HW, int C, int G, float eps,
                       void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && residual && z && relu_mask && gamma && beta && save_mean && save_invstd && workspace);
    PRIMIA_REQUIRE(gn_shape_ok(N, HW, C, G, dtype));
    if (workspace_bytes < primia_gn_workspace_bytes(N, C, G)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return gn_fwd_impl<float>(y, residual, z, gamma, beta, save_mean, save_invstd, N, HW, C, G, eps, 1,
                                  (float*)workspace, st, relu_mask);
    if (dtype == PRIMIA_BF16)
        return gn_fwd_impl<bf16>(y, residual, z, gamma, beta, save_mean, save_invstd, N, HW, C, G, eps, 1,
                                 (float*)workspace, st, relu_mask);
    return PRIMIA_ERR_ARG;
}

int primia_gn_bwd_mask(const void* y, const uint8_t* relu_mask, const void* dz, void* dy, void* g_out,
                       const float* gamma, const float* save_mean, const float* save_invstd, float* ps_dgamma,
                       float* ps_dbeta, int N, int HW, int C, int G, void* workspace, int64_t workspace_bytes, int dtype,
                       primia_stream_t stream) {
    PRIMIA_REQUIRE(y && relu_mask && dz && dy && gamma && save_mean && save_invstd && ps_dgamma && ps_dbeta && workspace);
    PRIMIA_REQUIRE(gn_shape_ok(N, HW, C, G, dtype));
    if (workspace_bytes < primia_gn_workspace_bytes(N, C, G)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return gn_bwd_impl<float>(y, nullptr, dz, dy, g_out, gamma, save_mean, save_invstd, ps_dgamma, ps_dbeta, N, HW, C, G,
                                  0, (float*)workspace, st, nullptr, relu_mask);
    if (dtype == PRIMIA_BF16)
        return gn_bwd_impl<bf16>(y, nullptr, dz, dy, g_out, gamma, save_mean, save_invstd, ps_dgamma, ps_dbeta, N, HW, C, G,
                                 0, (float*)workspace, st, nullptr, relu_mask);
    return PRIMIA_ERR_ARG;
}

int primia_gn_relu_bwd(const void* y, const void* dz, void* dy, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, float* ps_dgamma, float* ps_dbeta, int N, int HW,
                       int C, int G, void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && dz && dy && gamma && beta && save_mean && save_invstd && ps_dgamma && ps_dbeta && workspace);
    PRIMIA_REQUIRE(gn_shape_ok(N, HW, C, G, dtype));
    if (workspace_bytes < primia_gn_workspace_bytes(N, C, G)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return gn_bwd_impl<float>(y, nullptr, dz, dy, nullptr, gamma, save_mean, save_invstd, ps_dgamma, ps_dbeta, N, HW, C,
                                  G, 0, (float*)workspace, st, beta);
    if (dtype == PRIMIA_BF16)
        return gn_bwd_impl<bf16>(y, nullptr, dz, dy, nullptr, gamma, save_mean, save_invstd, ps_dgamma, ps_dbeta, N, HW, C,
                                 G, 0, (float*)workspace, st, beta);
    return PRIMIA_ERR_ARG;
}

int primia_gn_relu_maxpool_fwd(const void* y, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                               float* save_mean, float* save_invstd, int N, int H, int W, int C, int G, float eps,
                               void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && pooled && argmax && gamma && beta && save_mean && save_invstd && workspace);
    PRIMIA_REQUIRE(H > 0 && W > 0 && gn_shape_ok(N, H * W, C, G, dtype));
    if (workspace_bytes < primia_gn_workspace_bytes(N, C, G)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        gn_stats<float>(y, save_mean, save_invstd, N, H * W, C, G, eps, (float*)workspace, st);
    else if (dtype == PRIMIA_BF16)
        gn_stats<bf16>(y, save_mean, save_invstd, N, H * W, C, G, eps, (float*)workspace, st);
    else
        return PRIMIA_ERR_ARG;
    launch_gn_relu_pool_fwd(y, pooled, argmax, gamma, beta, save_mean, save_invstd, N, H, W, C, G, dtype, st);
    return launch_status();
}

int primia_gn_relu_maxpool_bwd(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax, void* dy,
                               const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                               float* ps_dgamma, float* ps_dbeta, int N, int H, int W, int C, int G, void* workspace,
                               int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && pooled && dpooled && argmax && dy && gamma && beta && save_mean && save_invstd && ps_dgamma &&
                   ps_dbeta && workspace);
    PRIMIA_REQUIRE(H > 0 && W > 0 && gn_shape_ok(N, H * W, C, G, dtype));
    if (H % 2 || W % 2) return PRIMIA_ERR_UNSUPPORTED;
    if (workspace_bytes < primia_gn_workspace_bytes(N, C, G)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return gn_relu_pool_bwd_impl<float>(y, pooled, dpooled, argmax, dy, gamma, beta, save_mean, save_invstd, ps_dgamma,
                                            ps_dbeta, N, H, W, C, G, (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return gn_relu_pool_bwd_impl<bf16>(y, pooled, dpooled, argmax, dy, gamma, beta, save_mean, save_invstd, ps_dgamma,
                                           ps_dbeta, N, H, W, C, G, (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

int primia_persample_sqnorm(const float* x, int N, int64_t per_sample, double* sq_acc, primia_stream_t stream) {
    PRIMIA_REQUIRE(x && sq_acc && N > 0 && per_sample > 0);
    long b = (per_sample + 255) / 256;
    if (b > 64) b = 64;
    persample_sqnorm_kernel<<<dim3((int)b, N), 256, 0, (hipStream_t)stream>>>(x, per_sample, sq_acc);
    return launch_status();
}

int primia_persample_sqnorm_many(const void* xs_dev, const int* widths_dev, int count, int N, double* sq_acc,
                                 primia_stream_t stream) {
    if (count == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(xs_dev && widths_dev && sq_acc && count > 0 && N > 0);
    persample_sqnorm_many_kernel<<<dim3(count, N), 256, 0, (hipStream_t)stream>>>((const float* const*)xs_dev, widths_dev,
                                                                                 sq_acc);
    return launch_status();
}

int primia_dp_clip_factors(const double* sq, float* clip, int N, float max_grad_norm, primia_stream_t stream) {
    PRIMIA_REQUIRE(sq && clip && N > 0 && max_grad_norm > 0.f);
    dp_clip_factor_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(sq, clip, N, max_grad_norm);
    return launch_status();
}

int primia_scale_rows(void* x, const float* s, int N, int64_t elems_per_sample, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(x && s && N > 0 && elems_per_sample > 0);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32) {
        PRIMIA_REQUIRE(elems_per_sample % 4 == 0);
        const long cps = elems_per_sample / 4, nch = cps * N;
        scale_rows_kernel<float><<<gn_rows_grid(N, cps), 256, 0, st>>>((float*)x, s, nch, cps);
    } else if (dtype == PRIMIA_BF16) {
        PRIMIA_REQUIRE(elems_per_sample % 8 == 0);
        const long cps = elems_per_sample / 8, nch = cps * N;
        scale_rows_kernel<bf16><<<gn_rows_grid(N, cps), 256, 0, st>>>((bf16*)x, s, nch, cps);
    } else {
        return PRIMIA_ERR_ARG;
    }
    return launch_status();
}

int primia_scale_rows_many(void* const* xs_host, const int64_t* elems_per_sample_host, int count, const float* s, int N,
                           int dtype, primia_stream_t stream) {
    if (count == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(xs_host && elems_per_sample_host && s && N > 0 && count > 0 && count <= kScaleMany);
    PRIMIA_REQUIRE(dtype == PRIMIA_F32 || dtype == PRIMIA_BF16);
    const int ch = dtype == PRIMIA_F32 ? 4 : 8;
    ScaleManyArgs a;
    long cmax = 0;
    for (int i = 0; i < count; ++i) {
        PRIMIA_REQUIRE(xs_host[i] && elems_per_sample_host[i] > 0 && elems_per_sample_host[i] % ch == 0);
        a.x[i] = xs_host[i];
        a.cps[i] = elems_per_sample_host[i] / ch;
        if (a.cps[i] > cmax) cmax = a.cps[i];
    }
    dim3 g = gn_rows_grid(N, cmax);
    g.z = (unsigned)count;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        scale_rows_many_kernel<float><<<g, 256, 0, st>>>(a, s);
    else
        scale_rows_many_kernel<bf16><<<g, 256, 0, st>>>(a, s);
    return launch_status();
}

int primia_weighted_colsum(const float* x, const float* w, float* out, int N, int C, primia_stream_t stream) {
    PRIMIA_REQUIRE(x && w && out && N > 0 && C > 0);
    weighted_colsum_kernel<<<(C + 15) / 16, 256, 0, (hipStream_t)stream>>>(x, w, out, N, C);
    return launch_status();
}

int primia_weighted_colsum_many(const void* xs_dev, const float* w, const void* outs_dev, const int* widths_dev, int count,
                                int max_width, int N, primia_stream_t stream) {
    if (count == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(xs_dev && w && outs_dev && widths_dev && count > 0 && max_width > 0 && N > 0);
    weighted_colsum_many_kernel<<<dim3((max_width + 15) / 16, count), 256, 0, (hipStream_t)stream>>>(
        (const float* const*)xs_dev, w, (float* const*)outs_dev, widths_dev, N);
    return launch_status();
}

int primia_fc_persample_grads(const float* x, const float* dy, float* ps, int N, int in_f, int out_f,
                              primia_stream_t stream) {
    PRIMIA_REQUIRE(x && dy && ps && N > 0 && in_f > 0 && out_f > 0);
    const long total = (long)N * ((long)out_f * in_f + out_f);
    fc_persample_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(x, dy, ps, N, in_f, out_f);
    return launch_status();
}

int primia_dp_add_noise(float* g, const float* noise, int64_t n, float sigma, float inv_batch, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(g && noise && n >= 0);
    if (n == 0) return PRIMIA_OK;
    long b = (n + 255) / 256;
    dp_noise_kernel<<<(int)(b > 4096 ? 4096 : b), 256, 0, (hipStream_t)stream>>>(g, noise, n, sigma, inv_batch);
    return launch_status();
}

}  // extern "C"
