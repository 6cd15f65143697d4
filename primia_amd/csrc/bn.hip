// BatchNorm2d training/eval forward and backward on [M, C] (NHWC) tensors, fused with ReLU and
// the residual add.  All kernels are HBM streaming passes with 16-byte accesses; the per-channel
// reductions are two-level and deterministic: each block reduces a contiguous slab of rows to an
// fp32 partial (wavefront-local accumulation, LDS tree across row groups), a one-block finalize
// kernel combines the <= 1024 partials per channel in fp64.
//
// Reference semantics: torch.nn.functional.batch_norm as called from nn.BatchNorm2d
// (torchlib/models.py:261-264, 382): biased variance for normalisation, unbiased for the running
// estimate, momentum 0.1, eps 1e-5.
#include <stdlib.h>

#include "common.h"

namespace primia {

constexpr int kMaxPartialBlocks = 1024;

// z before activation: one explicit fma, so the backward pass can recompute the ReLU mask from y
// bit-identically to what the forward pass stored (see primia_bn_relu_bwd).
__device__ __forceinline__ float bn_affine(float y, float mean, float scale, float beta) {
    return __builtin_fmaf(y - mean, scale, beta);
}

// ---- generic column reduction of two per-element quantities ------------------------------------
// Threads are laid out [rows_per_pass][C/CH]; thread (rg, cc) owns channels cc*CH..+CH-1.
template <typename T, typename F, int UNROLL = 0>
__global__ __launch_bounds__(256) void colreduce2_kernel(F f, long M, int C, long rows_per_block,
                                                         float* __restrict__ partials) {
    constexpr int CH = Chunk<T>::N;
    const int tpr = C / CH;        // threads per row
    const int rpp = 256 / tpr;     // rows per pass
    const int rg = threadIdx.x / tpr, cc = threadIdx.x % tpr;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;

    float s1[CH], s2[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) s1[i] = s2[i] = 0.f;
    if (rg < rpp) {
        F lf = f;
        lf.prepare(cc * CH);  // per-channel constants -> registers (the functor's stores must not force reloads)
        long r = r0 + rg;
        constexpr int U = UNROLL ? UNROLL : F::kUnroll;  // rows per trip: independent loads overlap (measured per functor)
        for (; r + (long)(U - 1) * rpp < r1; r += (long)U * rpp) {
#pragma unroll
            for (int u = 0; u < U; ++u) lf(r + (long)u * rpp, (r + (long)u * rpp) * C + cc * CH, cc * CH, s1, s2);
        }
        for (; r < r1; r += rpp) lf(r, r * C + cc * CH, cc * CH, s1, s2);
    }
    // cross-row-group reduction through LDS: [rpp][C] floats x 2 (rpp*C <= 256*CH)
    __shared__ float red[2][256 * CH];
    if (rg < rpp) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            red[0][rg * C + cc * CH + i] = s1[i];
            red[1][rg * C + cc * CH + i] = s2[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int g = 0; g < rpp; ++g) {
            a += red[0][g * C + c];
            b += red[1][g * C + c];
        }
        partials[((long)blockIdx.x * 2 + 0) * C + c] = a;
        partials[((long)blockIdx.x * 2 + 1) * C + c] = b;
    }
}

template <typename T>
struct StatsFn {
    static constexpr int kUnroll = 4;
    const T* y;
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void operator()(long, long off, int c0, float* s1, float* s2) const {
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

template <typename T>
struct BwdFn {
    static constexpr int kUnroll = 2;
    const T* y;
    const T* z;   // may be null (no relu)
    const T* dz;
    const float* mean;
    const float* invstd;
    const float* gamma;  // with beta: no z, the mask (z > 0) is recomputed from y
    const float* beta;
    const uint8_t* mask;  // or: one byte per chunk written by the forward pass (bit i = z_i > 0)
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void operator()(long, long off, int c0, float* s1, float* s2) const {
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
        } else if (beta) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float zz = bn_affine(vy[i], mean[c0 + i], invstd[c0 + i] * gamma[c0 + i], beta[c0 + i]);
                vg[i] = zz > 0.f ? vg[i] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float xh = (vy[i] - mean[c0 + i]) * invstd[c0 + i];
            s1[i] += vg[i];
            s2[i] += vg[i] * xh;
        }
    }
};

// Combine the per-block partials in fp64.  Block = 16 channels x 64 partial-slices (1024 threads: the
// kernel is pure load latency, <= 16 dependent-free trips per thread); the slices are reduced through LDS.  mode 0: batch statistics -> mean / invstd / running stats.
// mode 1: backward sums -> dbeta (sum g) / dgamma (sum g*xhat).
constexpr int kFinSlices = 64;
// One thread's share of the partial rows (k = ks, ks + 64, ...), NQ sums per row: eight rows = 8 * NQ independent loads
// are in flight before the first add (the adds keep the row order, so the result does not depend on the batching).
// With `#pragma unroll 4` over dependent double adds the compiler kept ~4 loads in flight and the kernel was four
// memory latencies long (5.8 us per call, 34 calls per training step).
template <int NQ>
__device__ __forceinline__ void fin_gather(const float* __restrict__ partials, int nblk, int C, int c, int ks,
                                           double (&acc)[NQ]) {
    for (int k0 = ks; k0 < nblk; k0 += 8 * kFinSlices) {
        float v[8][NQ];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u * kFinSlices;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                // unconditional load from a clamped row, then a select: no branch between the loads
                const float t = partials[((long)(k < nblk ? k : nblk - 1) * NQ + q) * C + c];
                v[u][q] = k < nblk ? t : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] += (double)v[u][q];
    }
}

__global__ __launch_bounds__(16 * kFinSlices) void bn_finalize_kernel(const float* __restrict__ partials, int nblk,
                                                          int C, long M, int mode, float eps,
                                                          float momentum, float* out1, float* out2,
                                                          float* running_mean, float* running_var,
                                                          const float* scale2 = nullptr) {
    __shared__ double sa[kFinSlices][17], sb[kFinSlices][17];
    const int cl = threadIdx.x & 15, ks = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    // what the last line needs from memory besides the sums is requested FIRST: left where they are used, these loads start
    // after the reduction and the kernel — pure latency, 33 launches per training step — is one memory round trip longer
    // (round 6, same-box A/B: 4.713 -> 4.696 ms per step.  The same idea in the streaming apply kernels — first trip's rows
    // requested before the constants' barrier — measured SLOWER, 4.740 -> 4.761 ms: the rows then live in registers across the
    // barrier and the loop carries a select; reverted)
    float rm0 = 0.f, rv0 = 0.f, sc2v = 1.f;
    if (ks == 0 && c < C) {
        if (mode == 0 && running_mean) {
            rm0 = running_mean[c];
            rv0 = running_var[c];
        } else if (mode != 0 && scale2) {
            sc2v = scale2[c];
        }
    }
    double acc2[2] = {0.0, 0.0};
    if (c < C) fin_gather<2>(partials, nblk, C, c, ks, acc2);
    double a = acc2[0], b = acc2[1];
    sa[ks][cl] = a;
    sb[ks][cl] = b;
    __syncthreads();
    if (ks >= 4) return;
    // 64 -> 4 slices by 4 threads per channel (wave 0), then two shuffles
    for (int k = ks + 4; k < kFinSlices; k += 4) {
        a += sa[k][cl];
        b += sb[k][cl];
    }
    // the four survivors of a channel sit in one wave at lanes cl, cl + 16, cl + 32, cl + 48
    a += __shfl_down(a, 32);
    b += __shfl_down(b, 32);
    a += __shfl_down(a, 16);
    b += __shfl_down(b, 16);
    if (ks != 0 || c >= C) return;
    if (mode == 0) {
        const double mean = a / (double)M;
        double var = b / (double)M - mean * mean;
        if (var < 0.0) var = 0.0;
        out1[c] = (float)mean;
        out2[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)rm0 + (double)momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)rv0 + (double)momentum * unb);
        }
    } else {
        out1[c] = (float)a;  // dbeta
        // dgamma; mode 1 with scale2: the second sum is sum g*(y - mean) and still lacks the factor invstd
        out2[c] = scale2 ? (float)(b * (double)sc2v) : (float)b;
    }
}

// Finalize of the stem's BatchNorm backward sums when their partials came out of a data-gradient write-back at POOLED resolution
// (conv3x3_c64_kernel<true, 3, 3>: sum g, sum g * (p - beta) / gamma with g = dpooled * [p > 0]).  For a channel whose gamma is 0
// or tiny against beta (common.h pool_xhat_recoverable) xhat cannot be recovered from the stored p: the producer wrote a zero
// partial and this kernel reads y at the argmax positions for that channel (what PoolScatterFn does in its rare branch) — one
// block walks the pooled tensor for it.
template <typename T>
__global__ __launch_bounds__(16 * kFinSlices) void bn_finalize_pool_kernel(
    const float* __restrict__ partials, int nblk, int C, float* __restrict__ dbeta, float* __restrict__ dgamma,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, const T* __restrict__ p, const T* __restrict__ dp, const uint8_t* __restrict__ argmax,
    const T* __restrict__ y, int N, int H, int W, int Ho, int Wo) {
    __shared__ double sa[kFinSlices][17], sb[kFinSlices][17];
    __shared__ double sfix[16 * kFinSlices / 64];
    const int cl = threadIdx.x & 15, ks = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double acc2[2] = {0.0, 0.0};
    if (c < C) fin_gather<2>(partials, nblk, C, c, ks, acc2);
    sa[ks][cl] = acc2[0];
    sb[ks][cl] = acc2[1];
    __syncthreads();
    if (ks == 0 && c < C) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < kFinSlices; ++k) {
            a += sa[k][cl];
            b += sb[k][cl];
        }
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
    }
    for (int j = 0; j < 16; ++j) {          // (block-uniform)
        const int cj = blockIdx.x * 16 + j;
        if (cj >= C || pool_xhat_recoverable(gamma[cj], beta[cj])) continue;
        const long Mp = (long)N * Ho * Wo;
        const float mu = mean[cj], is = invstd[cj];
        double t = 0.0;
        for (long r = threadIdx.x; r < Mp; r += blockDim.x) {
            const float pv = Elem<T>::load(p + r * C + cj);
            if (!(pv > 0.f)) continue;
            const int wo = (int)(r % Wo);
            const long q = r / Wo;
            const int ho = (int)(q % Ho), n = (int)(q / Ho);
            const int code = argmax[r * C + cj];
            const int h = 2 * ho - 1 + code / 3, w = 2 * wo - 1 + code % 3;
            const float yv = Elem<T>::load(y + (((long)n * H + h) * W + w) * C + cj);
            t += (double)(Elem<T>::load(dp + r * C + cj) * ((yv - mu) * is));
        }
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sfix[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0.0;
            for (int k = 0; k < 16 * kFinSlices / 64; ++k) a += sfix[k];
            dgamma[cj] = (float)a;
        }
    }
}

// ---- the finalize launch folded into its consumer (round 5) -------------------------------------------------------
// The one-block-per-16-channels bn_finalize_kernel between a statistics producer and the apply pass costs 5.6 us of pure
// latency 37 times per step.  Here the apply kernel's FIRST C / 4 blocks do that work themselves — 4 channels x 64
// partial-slices each, the same slicing and the same order of additions as bn_finalize_kernel: bit-identical mean / invstd /
// running statistics — publish the constants with write-through (agent-scope) stores and raise one flag word per block;
// every block of the kernel then waits for the flags of the channels it reads.  No fence anywhere: the partial sums were
// written by an EARLIER kernel (visible at the boundary), the constants and flags travel as agent-scope atomics, the
// order "constants before flag" is a vmcnt wait + a block barrier.  (Round 3's hand-offs sat on the PRODUCER side and
// needed either a device-scope fence per block behind 400 MB of dirty L2 lines or one block pulling the whole table;
// int64 atomics from the producers serialise at 23 M/s per address: profiles/r05_bn_finalize_atomics.txt.)
// `flags`: C / 4 words, zero when the kernel starts (the engine zeroes every layer's words once per step).
struct BnInline {
    const float* partials;   // [nblk][2][C]; null: the caller already ran the finalize launch
    int nblk;
    long M;
    float eps, momentum;
    float* running_mean;
    float* running_var;
    float* save_mean;
    float* save_invstd;
    unsigned* flags;
};

__device__ __forceinline__ void bn_inline_finalize_stats(const BnInline& q, int C, float* lds_mean, float* lds_invstd) {
    const int nfin = (C + 3) >> 2;
    if ((int)blockIdx.x < nfin) {
        __shared__ double sa[kFinSlices][5], sb[kFinSlices][5];
        const int cl = threadIdx.x & 3, ks = threadIdx.x >> 2;      // 256 threads: 4 channels x 64 slices
        const int c = blockIdx.x * 4 + cl;
        double acc2[2] = {0.0, 0.0};
        if (c < C) fin_gather<2>(q.partials, q.nblk, C, c, ks, acc2);
        double a = acc2[0], b = acc2[1];
        sa[ks][cl] = a;
        sb[ks][cl] = b;
        __syncthreads();
        if (ks < 4) {
            for (int k = ks + 4; k < kFinSlices; k += 4) {
                a += sa[k][cl];
                b += sb[k][cl];
            }
        }
        __syncthreads();
        if (ks < 4) {
            sa[ks][cl] = a;
            sb[ks][cl] = b;
        }
        __syncthreads();
        if (ks == 0 && c < C) {
            // bn_finalize_kernel's two shuffles: (g0 + g2) + (g1 + g3)
            a = (sa[0][cl] + sa[2][cl]) + (sa[1][cl] + sa[3][cl]);
            b = (sb[0][cl] + sb[2][cl]) + (sb[1][cl] + sb[3][cl]);
            const double mean = a / (double)q.M;
            double var = b / (double)q.M - mean * mean;
            if (var < 0.0) var = 0.0;
            __hip_atomic_store(q.save_mean + c, (float)mean, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q.save_invstd + c, (float)(1.0 / sqrt(var + (double)q.eps)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q.running_mean) {
                const double unb = q.M > 1 ? var * (double)q.M / (double)(q.M - 1) : var;
                q.running_mean[c] = (float)((1.0 - q.momentum) * (double)q.running_mean[c] + (double)q.momentum * mean);
                q.running_var[c] = (float)((1.0 - q.momentum) * (double)q.running_var[c] + (double)q.momentum * unb);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the constants have reached the coherence point
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(q.flags + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        while (__hip_atomic_load(q.flags + (c >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
        asm volatile("" ::: "memory");
        lds_mean[c] = __hip_atomic_load(q.save_mean + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_invstd[c] = __hip_atomic_load(q.save_invstd + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T>
__device__ __forceinline__ float round_to(float v) {
    return v;
}
template <>
__device__ __forceinline__ float round_to<bf16>(float v) {
    return bf16_to_f32(f32_to_bf16(v));
}

// z = act((y - mean) * (invstd * gamma) + beta [+ residual]); optionally one mask byte per 16-byte chunk with
// bit i = (stored z_i > 0): the backward passes of a residual layer read it instead of z (1/16 of the bytes)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                       T* __restrict__ z, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd_or_var, float eps,
                                                       int eval_mode, long nchunks, int C, int relu,
                                                       uint8_t* __restrict__ mask_out = nullptr, BnInline inl = BnInline{}) {
    constexpr int CH = Chunk<T>::N;
    __shared__ float sm[3][512];
    if (inl.partials) {
        // the finalize folded into this kernel: mean -> sm[0], invstd -> sm[1] (then scaled by gamma below)
        bn_inline_finalize_stats(inl, C, sm[0], sm[1]);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            sm[1][c] = sm[1][c] * gamma[c];
            sm[2][c] = beta[c];
        }
    } else {
    for (int c = threadIdx.x; c < C; c += 256) {
        const float is = eval_mode ? 1.f / sqrtf(invstd_or_var[c] + eps) : invstd_or_var[c];
        sm[0][c] = mean[c];
        sm[1][c] = is * gamma[c];
        sm[2][c] = beta[c];
    }
    }
    __syncthreads();
    const int cpr = C / CH;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nchunks; q += stride) {
        const int c0 = (int)(q % cpr) * CH;
        float v[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + q * CH), v);
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = bn_affine(v[i], sm[0][c0 + i], sm[1][c0 + i], sm[2][c0 + i]);
        if (res) {
            float r[CH];
            Chunk<T>::unpack(*(const u32x4*)(res + q * CH), r);
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] += r[i];
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        // (non-temporal: the activation is read again a kernel later from HBM / the Infinity Cache anyway, and not
        // allocating it in the L2 is worth -0.5 % on the step — same-box A/B, profiles/r04_lh3_experiments.txt)
        __builtin_nontemporal_store(Chunk<T>::pack(v), (u32x4*)(z + q * CH));
        if (mask_out) {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < CH; ++i) m |= (round_to<T>(v[i]) > 0.f ? 1u : 0u) << i;
            mask_out[q] = (uint8_t)m;
        }
    }
}

// Transition block forward: z = relu(bn2(y2) + bn_d(yd)) with the downsample BatchNorm applied on the fly (its output,
// rounded to the storage type exactly as if it had been stored, is never written or read back) + the 1-bit ReLU mask.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_pair_kernel(const T* __restrict__ y2, const T* __restrict__ yd,
                                                            T* __restrict__ z, const float* __restrict__ gamma2,
                                                            const float* __restrict__ beta2,
                                                            const float* __restrict__ mean2,
                                                            const float* __restrict__ invstd2,
                                                            const float* __restrict__ gammad,
                                                            const float* __restrict__ betad,
                                                            const float* __restrict__ meand,
                                                            const float* __restrict__ invstdd, long nchunks, int C,
                                                            uint8_t* __restrict__ mask_out) {
    constexpr int CH = Chunk<T>::N;
    __shared__ float sm[6][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        sm[0][c] = mean2[c];
        sm[1][c] = invstd2[c] * gamma2[c];
        sm[2][c] = beta2[c];
        sm[3][c] = meand[c];
        sm[4][c] = invstdd[c] * gammad[c];
        sm[5][c] = betad[c];
    }
    __syncthreads();
    const int cpr = C / CH;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nchunks; q += stride) {
        const int c0 = (int)(q % cpr) * CH;
        float v[CH], r[CH];
        Chunk<T>::unpack(*(const u32x4*)(y2 + q * CH), v);
        Chunk<T>::unpack(*(const u32x4*)(yd + q * CH), r);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            v[i] = bn_affine(v[i], sm[0][c0 + i], sm[1][c0 + i], sm[2][c0 + i]);
            v[i] += round_to<T>(bn_affine(r[i], sm[3][c0 + i], sm[4][c0 + i], sm[5][c0 + i]));
            v[i] = fmaxf(v[i], 0.f);
        }
        *(u32x4*)(z + q * CH) = Chunk<T>::pack(v);
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < CH; ++i) m |= (round_to<T>(v[i]) > 0.f ? 1u : 0u) << i;
        mask_out[q] = (uint8_t)m;
    }
}

// dy = gamma*invstd * (g - dbeta/M - xhat*dgamma/M); optionally g_out = g.
// U chunks per thread and trip: their loads are requested together (U = 2: twice the bytes in flight per CU).
template <typename T, int U = 1>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ y, const T* __restrict__ z,
                                                           const T* dz, T* __restrict__ dy, T* g_out,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, float inv_m,
                                                           long nchunks, int C, const float* __restrict__ beta,
                                                           const uint8_t* __restrict__ mask = nullptr) {
    constexpr int CH = Chunk<T>::N;
    __shared__ float sm[7][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        sm[0][c] = mean[c];
        sm[1][c] = invstd[c];
        sm[2][c] = gamma[c] * invstd[c];
        sm[3][c] = dbeta[c] * inv_m;
        sm[4][c] = dgamma[c] * inv_m;
        sm[5][c] = invstd[c] * gamma[c];   // the forward pass's scale (same operand order)
        sm[6][c] = beta ? beta[c] : 0.f;
    }
    __syncthreads();
    const int cpr = C / CH;
    const long stride = (long)gridDim.x * 256;
    for (long q0 = (long)blockIdx.x * 256 + threadIdx.x; q0 < nchunks; q0 += stride * U) {
        u32x4 ry[U], rg[U], rz[U];
        unsigned rm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = q0 + u * stride;
            const long qc = q < nchunks ? q : q0;          // clamped: the loads of all U chunks go out unconditionally
            ry[u] = *(const u32x4*)(y + qc * CH);
            rg[u] = *(const u32x4*)(dz + qc * CH);
            if (z) rz[u] = *(const u32x4*)(z + qc * CH);
            else if (mask) rm[u] = mask[qc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long q = q0 + u * stride;
            if (q >= nchunks) break;
            const int c0 = (int)(q % cpr) * CH;
            float vy[CH], vg[CH];
            Chunk<T>::unpack(ry[u], vy);
            Chunk<T>::unpack(rg[u], vg);
            if (z) {
                float vz[CH];
                Chunk<T>::unpack(rz[u], vz);
#pragma unroll
                for (int i = 0; i < CH; ++i) vg[i] = vz[i] > 0.f ? vg[i] : 0.f;
            } else if (mask) {
                const unsigned m = rm[u];
#pragma unroll
                for (int i = 0; i < CH; ++i) vg[i] = (m >> i) & 1u ? vg[i] : 0.f;
            } else if (beta) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const float zz = bn_affine(vy[i], sm[0][c0 + i], sm[5][c0 + i], sm[6][c0 + i]);
                    vg[i] = zz > 0.f ? vg[i] : 0.f;
                }
            }
            if (g_out) *(u32x4*)(g_out + q * CH) = Chunk<T>::pack(vg);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float xh = (vy[i] - sm[0][c0 + i]) * sm[1][c0 + i];
                vy[i] = sm[2][c0 + i] * (vg[i] - sm[3][c0 + i] - xh * sm[4][c0 + i]);
            }
            *(u32x4*)(dy + q * CH) = Chunk<T>::pack(vy);
        }
    }
}

// =================================================================================================
// BatchNorm + ReLU + MaxPool(3, 2, 1) fused (the stem: conv1 -> bn1 -> relu -> maxpool,
// torchlib/models.py:466-471).  The unfused chain writes z = relu(bn(y)) (the largest activation of
// the network), reads it back for the pool, and in the backward pass materialises the pool's input
// gradient dz only for the two BatchNorm backward passes to read it again.  Fused:
//   forward : pooled, argmax = maxpool(relu(bn(y)))   straight from y              (z never exists)
//   backward: the BatchNorm sums are formed at POOLED resolution (a window's gradient reaches exactly its
//             argmax, whose activation is the pooled value itself: PoolScatterFn); only the apply pass gathers
//             g(h, w) = [z(h, w) > 0] * sum over the <= 4 windows whose argmax is (h, w) of dpooled.
//             The pool's own backward pass, its 411 MB output and both re-reads of it are gone.
// Values are rounded to the storage type before the window comparison, so argmax / ties are exactly
// those of the unfused kernels (first maximum in scan order, like torch's max_pool2d).
// =================================================================================================

// PW x PH = output pixels per thread (along W: 1, or 2 when Wo is even — the two windows share a column, 5 loads per
// row instead of 6; along H: 1, or 2 when Ho is even — the two windows share input row 2*ho + 1, 5 rows instead of 6).
// Default: PW = 2 (when Wo is even), PH = 1 — see launch_bn_relu_pool_fwd for the measurement behind it.
template <typename T, int PW, int PH>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const T* __restrict__ y, T* __restrict__ pooled,
                                                               uint8_t* __restrict__ argmax,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, int N, int H, int W,
                                                               int C, int Ho, int Wo, int G) {
    constexpr int CH = Chunk<T>::N;
    constexpr int NCOL = 2 * PW + 1, NROW = 2 * PH + 1;
    __shared__ float sm[3][512];
    // G > 0: GroupNorm — mean / invstd are [N][G] (per sample and group), read by each thread for its own sample
    if (G == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            sm[0][c] = mean[c];
            sm[1][c] = invstd[c] * gamma[c];
            sm[2][c] = beta[c];
        }
    }
    __syncthreads();
    const int cpr = C / CH;
    const int Wq = Wo / PW, Hq = Ho / PH;
    const long total = (long)N * Hq * Wq * cpr;
    // (an XCD-contiguous block order was measured slower, 161 -> 178 us: the eight L2s then stream eight distant
    // regions instead of sharing one)
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int cc = (int)(q % cpr);
    long t = q / cpr;
    const int wq = (int)(t % Wq);
    t /= Wq;
    const int hq = (int)(t % Hq);
    const int n = (int)(t / Hq);
    const int c0 = cc * CH;
    float best[PH][PW][CH];
    int pos[PH][PW][CH];
#pragma unroll
    for (int kh = 0; kh < PH; ++kh)
#pragma unroll
        for (int k = 0; k < PW; ++k)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                best[kh][k][i] = 0.f;
                pos[kh][k][i] = -1;
            }
    float mu[CH], sc[CH], be[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (G == 0) {
            mu[i] = sm[0][c0 + i];
            sc[i] = sm[1][c0 + i];
            be[i] = sm[2][c0 + i];
        } else {
            const int ng = n * G + (c0 + i) / (C / G);
            mu[i] = mean[ng];
            sc[i] = invstd[ng] * gamma[c0 + i];
            be[i] = beta[c0 + i];
        }
    }
    const int w0 = wq * PW * 2 - 1;   // leftmost input column of the first window
    const int h0 = hq * PH * 2 - 1;   // topmost input row of the first window
#pragma unroll
    for (int rr = 0; rr < NROW; ++rr) {
        const int h = h0 + rr;
        if (h < 0 || h >= H) continue;
        const T* row = y + (((long)n * H + h) * W) * C + c0;
        u32x4 raw[NCOL];
#pragma unroll
        for (int col = 0; col < NCOL; ++col) {
            const int w = w0 + col;
            raw[col] = u32x4{0, 0, 0, 0};
            if (w >= 0 && w < W) raw[col] = *(const u32x4*)(row + (long)w * C);
        }
#pragma unroll
        for (int col = 0; col < NCOL; ++col) {
            const int w = w0 + col;
            if (w < 0 || w >= W) continue;
            float v[CH];
            Chunk<T>::unpack(raw[col], v);
#pragma unroll
            for (int i = 0; i < CH; ++i) v[i] = round_to<T>(fmaxf(bn_affine(v[i], mu[i], sc[i], be[i]), 0.f));
#pragma unroll
            for (int kh = 0; kh < PH; ++kh) {
                const int r = rr - 2 * kh;       // tap row of window kh (a window sees its taps in scan order r, s)
                if (r < 0 || r > 2) continue;
#pragma unroll
                for (int k = 0; k < PW; ++k) {
                    const int s_ = col - 2 * k;  // tap column of window k
                    if (s_ < 0 || s_ > 2) continue;
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const float z = v[i];
                        if (pos[kh][k][i] < 0 || z > best[kh][k][i] || z != z) {
                            best[kh][k][i] = z;
                            pos[kh][k][i] = r * 3 + s_;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int kh = 0; kh < PH; ++kh)
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const long o = (((long)n * Ho + hq * PH + kh) * Wo + wq * PW + k) * C + c0;
            *(u32x4*)(pooled + o) = Chunk<T>::pack(best[kh][k]);
            // the chunk's CH argmax bytes leave as one 4- / 8-byte store (o is a multiple of CH)
            uint32_t pk[CH / 4];
#pragma unroll
            for (int j = 0; j < CH / 4; ++j)
                pk[j] = (uint32_t)pos[kh][k][4 * j] | ((uint32_t)pos[kh][k][4 * j + 1] << 8) |
                        ((uint32_t)pos[kh][k][4 * j + 2] << 16) | ((uint32_t)pos[kh][k][4 * j + 3] << 24);
            if constexpr (CH == 8)
                *(u32x2*)(argmax + o) = u32x2{pk[0], pk[1]};
            else
                *(uint32_t*)(argmax + o) = pk[0];
        }
}

// The same kernel for bf16, PH = 1, on PACKED KEYS.  The instruction counters of round 3 showed the generic form bound
// by vector-instruction issue, not by memory (87 M instructions per launch at batch 256 = ~140 us of issue in a 170-us
// kernel): per window tap and channel it spends two compares, an or and two selects on the (best, argmax) pair.  A
// stored activation is a non-negative bf16 (or NaN), whose 16 bits order like the value, so
//     key = bits(z) << 4 | (15 - tap)            (tap = 3 r + s in scan order)
// orders first by value, then by EARLIER tap: max over the window's keys = (maximum, first position attaining it) — two
// instructions per tap (v_lshl_or_b32, v_max_u32), and the same result as the generic kernel bit for bit (a NaN, the
// largest pattern, wins as it does there).
// Block size: 256 threads (whole output rows per block were measured: no gain); consecutive blocks go to different
// XCDs, so the input row two vertically neighbouring windows share is fetched once per block that touches it; a block of
// whole rows shares them inside one CU.  Measured: step 4.925 / 4.937 ms (256 threads) vs 4.938 / 4.944 (2 rows) and
// 4.944 / 4.972 (4 rows) — the 1.5x fetch of this pass is served by the Infinity Cache and is not what bounds it.
template <int PW, int PH>
__global__ __launch_bounds__(1024) void bn_relu_pool_fwd_key_kernel(const bf16* __restrict__ y, bf16* __restrict__ pooled,
                                                                   uint8_t* __restrict__ argmax,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, int N, int H, int W,
                                                                   int C, int Ho, int Wo, int G) {
    constexpr int CH = 8, NCOL = 2 * PW + 1, NROW = 2 * PH + 1;
    __shared__ float sm[3][512];
    // G > 0: GroupNorm — mean / invstd are [N][G] (per sample and group), read by each thread for its own sample
    const int cpr = C / CH;
    const int Wq = Wo / PW, Hq = Ho / PH;
    const long total = (long)N * Hq * Wq * cpr;
    // group mode: a block that lies inside ONE sample (all but one in ~43 at 56 x 56) stages that sample's per-channel
    // constants in LDS like the BatchNorm mode — each thread fetched 8 x (mean, invstd, gamma, beta) itself before its
    // first data load (183 us against the BatchNorm mode's 140 for the same bytes)
    bool staged = G == 0;
    if (G == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            sm[0][c] = mean[c];
            sm[1][c] = invstd[c] * gamma[c];
            sm[2][c] = beta[c];
        }
    } else {
        const long per_n = (long)Hq * Wq * cpr;
        const long q0 = (long)blockIdx.x * blockDim.x;
        long q1 = q0 + blockDim.x - 1;
        if (q1 >= total) q1 = total - 1;
        const int n0 = (int)(q0 / per_n);
        if (q0 < total && n0 == (int)(q1 / per_n)) {
            staged = true;
            for (int c = threadIdx.x; c < C; c += blockDim.x) {
                const int ng = n0 * G + c / (C / G);
                sm[0][c] = mean[ng];
                sm[1][c] = invstd[ng] * gamma[c];
                sm[2][c] = beta[c];
            }
        }
    }
    __syncthreads();
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const int cc = (int)(q % cpr);
    long t = q / cpr;
    const int wq = (int)(t % Wq);
    t /= Wq;
    const int hq = (int)(t % Hq);
    const int n = (int)(t / Hq);
    const int c0 = cc * CH;
    unsigned key[PH][PW][CH];
#pragma unroll
    for (int kh = 0; kh < PH; ++kh)
#pragma unroll
        for (int k = 0; k < PW; ++k)
#pragma unroll
            for (int i = 0; i < CH; ++i) key[kh][k][i] = 0u;     // below every real key (15 - tap >= 7)
    float mu[CH], sc[CH], be[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (staged) {
            mu[i] = sm[0][c0 + i];
            sc[i] = sm[1][c0 + i];
            be[i] = sm[2][c0 + i];
        } else {
            const int ng = n * G + (c0 + i) / (C / G);
            mu[i] = mean[ng];
            sc[i] = invstd[ng] * gamma[c0 + i];
            be[i] = beta[c0 + i];
        }
    }
    const int w0 = wq * PW * 2 - 1;   // leftmost input column of the first window
    const int h0 = hq * PH * 2 - 1;   // topmost input row of the first window
#pragma unroll
    for (int rr = 0; rr < NROW; ++rr) {
        const int h = h0 + rr;
        if (h < 0 || h >= H) continue;
        const bf16* row = y + (((long)n * H + h) * W) * C + c0;
        u32x4 raw[NCOL];
#pragma unroll
        for (int col = 0; col < NCOL; ++col) {
            const int w = w0 + col;
            raw[col] = u32x4{0, 0, 0, 0};
            if (w >= 0 && w < W) raw[col] = *(const u32x4*)(row + (long)w * C);
        }
#pragma unroll
        for (int col = 0; col < NCOL; ++col) {
            const int w = w0 + col;
            if (w < 0 || w >= W) continue;
            float v[CH];
            Chunk<bf16>::unpack(raw[col], v);
            unsigned zb[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i)   // the stored activation's bits (sign cleared: -0 orders like +0, NaN stays NaN)
                zb[i] = (unsigned)f32_to_bf16(fmaxf(bn_affine(v[i], mu[i], sc[i], be[i]), 0.f)) & 0x7fffu;
#pragma unroll
            for (int kh = 0; kh < PH; ++kh) {
                const int r = rr - 2 * kh;       // tap row of window kh
                if (r < 0 || r > 2) continue;
#pragma unroll
                for (int k = 0; k < PW; ++k) {
                    const int s_ = col - 2 * k;  // tap column of window k
                    if (s_ < 0 || s_ > 2) continue;
                    const unsigned tail = 15u - (unsigned)(r * 3 + s_);
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const unsigned cand = (zb[i] << 4) | tail;
                        key[kh][k][i] = cand > key[kh][k][i] ? cand : key[kh][k][i];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int kh = 0; kh < PH; ++kh)
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const long o = (((long)n * Ho + hq * PH + kh) * Wo + wq * PW + k) * C + c0;
            u32x4 pv;
            uint32_t pk[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) pv[j] = (key[kh][k][2 * j] >> 4) | ((key[kh][k][2 * j + 1] >> 4) << 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                pk[j] = (15u - (key[kh][k][4 * j] & 15u)) | ((15u - (key[kh][k][4 * j + 1] & 15u)) << 8) |
                        ((15u - (key[kh][k][4 * j + 2] & 15u)) << 16) | ((15u - (key[kh][k][4 * j + 3] & 15u)) << 24);
            *(u32x4*)(pooled + o) = pv;
            *(u32x2*)(argmax + o) = u32x2{pk[0], pk[1]};
        }
}

// gradient w.r.t. z(n, h, w, c0..c0+CH-1) coming back through the pool.  A pixel lies in at most 2 x 2
// windows: along each axis candidate A = ((h + 1) >> 1, tap h + 1 - 2*ho) always exists (if in range) and
// candidate B = ((h - 1) >> 1, tap 2) only for odd h.  Branch-free: loads go to a clamped address and are
// discarded by the validity flag, so the compiler can overlap the loads of several pixels.
template <typename T>
__device__ __forceinline__ void pool_gather(const T* __restrict__ dp, const uint8_t* __restrict__ argmax, int n, int h,
                                            int w, int c0, int C, int Ho, int Wo, float* g) {
    constexpr int CH = Chunk<T>::N;
#pragma unroll
    for (int i = 0; i < CH; ++i) g[i] = 0.f;
    int hoc[2], rc[2], woc[2], sc[2];
    bool hv[2], wv[2];
    hoc[0] = (h + 1) >> 1; rc[0] = h + 1 - 2 * hoc[0]; hv[0] = hoc[0] < Ho;
    hoc[1] = (h - 1) >> 1; rc[1] = 2;                  hv[1] = (h & 1) != 0;
    woc[0] = (w + 1) >> 1; sc[0] = w + 1 - 2 * woc[0]; wv[0] = woc[0] < Wo;
    woc[1] = (w - 1) >> 1; sc[1] = 2;                  wv[1] = (w & 1) != 0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool ok = hv[a] && wv[b];
            const int ho = ok ? hoc[a] : 0, wo = ok ? woc[b] : 0;
            const long o = (((long)n * Ho + ho) * Wo + wo) * C + c0;
            float v[CH];
            Chunk<T>::unpack(*(const u32x4*)(dp + o), v);
            const unsigned code = ok ? (unsigned)(rc[a] * 3 + sc[b]) : 0xffu;  // 0xff never matches
            if (CH == 8) {
                const u32x2 m = *(const u32x2*)(argmax + o);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    g[i] += ((m[0] >> (8 * i)) & 0xffu) == code ? v[i] : 0.f;
                    g[4 + i] += ((m[1] >> (8 * i)) & 0xffu) == code ? v[4 + i] : 0.f;
                }
            } else {
                const uint32_t m = *(const uint32_t*)(argmax + o);
#pragma unroll
                for (int i = 0; i < CH; ++i) g[i] += ((m >> (8 * i)) & 0xffu) == code ? v[i] : 0.f;
            }
        }
}

// Reduction pass of the fused stem backward, over the POOLED pixels (a quarter of the input pixels, no gather):
// a window's gradient reaches exactly one input element, its argmax, whose activation IS the pooled value p, so
//     sum g        = sum over windows [p > 0] * dp
//     sum g * xhat = sum over windows [p > 0] * dp * xhat(argmax),   xhat = (p - beta) / gamma   (ReLU active)
// — two tensors at pooled resolution instead of y plus a 4-window gather per input element (400 -> ~50 us at
// batch 256).  (gamma == 0 makes z constant and xhat unrecoverable from p: those channels read y at the argmax.)
template <typename T>
struct PoolScatterFn {
    static constexpr int kUnroll = 4;
    const T* p;
    const T* dp;
    const uint8_t* argmax;
    const T* y;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    int H, W, C, Ho, Wo;
    float k_beta[Chunk<T>::N], k_rgamma[Chunk<T>::N], k_mean[Chunk<T>::N], k_invstd[Chunk<T>::N];
    bool any_zero_gamma;
    __device__ __forceinline__ void prepare(int c0) {
        any_zero_gamma = false;
#pragma unroll
        for (int i = 0; i < Chunk<T>::N; ++i) {
            const float g = gamma[c0 + i];
            k_beta[i] = beta[c0 + i];
            const bool rec = pool_xhat_recoverable(g, k_beta[i]);
            k_rgamma[i] = rec ? 1.f / g : 0.f;
            k_mean[i] = mean[c0 + i];
            k_invstd[i] = invstd[c0 + i];
            any_zero_gamma |= !rec;
        }
    }
    __device__ __forceinline__ void operator()(long row, long off, int c0, float* s1, float* s2) const {
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
            const int wo = (int)(row % Wo);
            const long t = row / Wo;
            const int ho = (int)(t % Ho), n = (int)(t / Ho);
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

// Apply pass of the fused stem backward: the pool gradient of every input element is gathered here (it is
// needed per element only now), masked with the ReLU mask recomputed from y, and turned into dy.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_apply_kernel(
    const T* __restrict__ y, const T* __restrict__ dp, const uint8_t* __restrict__ argmax, T* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_m,
    int N, int H, int W, int C, int Ho, int Wo, float rcp_w, float rcp_h) {
    constexpr int CH = Chunk<T>::N;
    __shared__ float sm[7][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        sm[0][c] = mean[c];
        sm[1][c] = invstd[c];
        sm[2][c] = gamma[c] * invstd[c];
        sm[3][c] = dbeta[c] * inv_m;
        sm[4][c] = dgamma[c] * inv_m;
        sm[5][c] = invstd[c] * gamma[c];
        sm[6][c] = beta[c];
    }
    __syncthreads();
    const int cpr = C / CH;
    const long total = (long)N * H * W * cpr;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += stride) {
        const int m = (int)(q / cpr);
        const int c0 = (int)(q - (long)m * cpr) * CH;
        int w = m % W;
        const int t = m / W;
        const int h = t % H, n = t / H;
        float vy[CH], vg[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + q * CH), vy);
        pool_gather<T>(dp, argmax, n, h, w, c0, C, Ho, Wo, vg);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float zz = bn_affine(vy[i], sm[0][c0 + i], sm[5][c0 + i], sm[6][c0 + i]);
            const float gi = zz > 0.f ? vg[i] : 0.f;
            const float xh = (vy[i] - sm[0][c0 + i]) * sm[1][c0 + i];
            vy[i] = sm[2][c0 + i] * (gi - sm[3][c0 + i] - xh * sm[4][c0 + i]);
        }
        *(u32x4*)(dy + q * CH) = Chunk<T>::pack(vy);
    }
}

static inline void reduce_geometry(long M, int C, int& nblk, long& rows_per_block) {
    // <= 1024 blocks; at least `min_rows` rows per block (option bn_minrows, default 32: the small late
    // layers are latency-bound with few blocks — 256 rows per block left layer4 with 49 blocks; measured
    // 6.518 / 6.492 / 6.508 ms per step at 64 / 32 / 16).
    const long min_rows = PRIMIA_OPT(bn_minrows) > 0 ? PRIMIA_OPT(bn_minrows) : 32;
    long nb = (M + min_rows - 1) / min_rows;
    if (nb > kMaxPartialBlocks) nb = kMaxPartialBlocks;
    if (nb < 1) nb = 1;
    rows_per_block = (M + nb - 1) / nb;
    nblk = (int)((M + rows_per_block - 1) / rows_per_block);
}

static inline int stream_blocks(long nchunks) {
    long b = (nchunks + 255) / 256;
    return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

static inline bool bn_shape_ok(long M, int C, int dtype) {
    const int ch = dtype == PRIMIA_F32 ? 4 : 8;
    return M > 0 && C > 0 && C <= 512 && C % ch == 0 && 256 % (C / ch) == 0;
}

template <typename T>
static int bn_fwd_train_impl(const void* y, const void* residual, void* z, const float* gamma,
                             const float* beta, float* running_mean, float* running_var,
                             float* save_mean, float* save_invstd, long M, int C, float eps,
                             float momentum, int relu, float* partials, hipStream_t st) {
    int nblk;
    long rpb;
    reduce_geometry(M, C, nblk, rpb);
    StatsFn<T> f{(const T*)y};
    colreduce2_kernel<T, StatsFn<T>><<<nblk, 256, 0, st>>>(f, M, C, rpb, partials);
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(partials, nblk, C, M, 0, eps, momentum, save_mean, save_invstd,
                                           running_mean, running_var);
    const long nchunks = M * C / Chunk<T>::N;
    bn_apply_kernel<T><<<stream_blocks(nchunks), 256, 0, st>>>((const T*)y, (const T*)residual, (T*)z, gamma,
                                                               beta, save_mean, save_invstd, eps, 0, nchunks,
                                                               C, relu);
    return launch_status();
}

template <typename T>
static int bn_bwd_impl(const void* y, const void* z, const void* dz, void* dy, void* g_out,
                       const float* gamma, const float* save_mean, const float* save_invstd,
                       float* dgamma, float* dbeta, long M, int C, int relu, float* partials,
                       hipStream_t st, const float* beta = nullptr, const uint8_t* mask = nullptr) {
    int nblk;
    long rpb;
    reduce_geometry(M, C, nblk, rpb);
    BwdFn<T> f{(const T*)y, relu ? (const T*)z : nullptr, (const T*)dz, save_mean, save_invstd, gamma, beta, mask};
    colreduce2_kernel<T, BwdFn<T>><<<nblk, 256, 0, st>>>(f, M, C, rpb, partials);
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(partials, nblk, C, M, 1, 0.f, 0.f, dbeta, dgamma, nullptr, nullptr);
    const long nchunks = M * C / Chunk<T>::N;
    const int unroll = PRIMIA_OPT(bn_unroll);
    if (unroll == 2 && nchunks >= 4L * 2048 * 256)
        bn_bwd_apply_kernel<T, 2><<<stream_blocks(nchunks), 256, 0, st>>>(
            (const T*)y, relu ? (const T*)z : nullptr, (const T*)dz, (T*)dy, (T*)g_out, gamma, save_mean,
            save_invstd, dbeta, dgamma, (float)(1.0 / (double)M), nchunks, C, beta, mask);
    else
        bn_bwd_apply_kernel<T><<<stream_blocks(nchunks), 256, 0, st>>>(
            (const T*)y, relu ? (const T*)z : nullptr, (const T*)dz, (T*)dy, (T*)g_out, gamma, save_mean,
            save_invstd, dbeta, dgamma, (float)(1.0 / (double)M), nchunks, C, beta, mask);
    return launch_status();
}

// =================================================================================================
// Transition block (conv2 -> bn2, + downsample -> bn_d, ReLU): both BatchNorm backward passes read the SAME incoming
// gradient g = dz * relu_mask.  Separately they cost: bn2 (reduce: y2, dz | apply: y2, dz -> dy2, g) and then bn_d
// (reduce: yd, g | apply: yd, g -> dyd) = 7 tensor reads + 3 writes.  Together: one reduction with three sums (sum g,
// sum g*xhat2, sum g*xhat_d) over y2, yd, dz and one apply pass y2, yd, dz -> dy2, dyd = 6 reads + 2 writes, and g is
// never materialised (the block's input gradient comes from primia_conv2d_dgrad_pair(dy1, dyd)).  Same reduction
// geometry and arithmetic as the separate kernels: bit-identical results.
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_pair_reduce_kernel(const T* __restrict__ y2, const T* __restrict__ yd,
                                                                 const T* __restrict__ dz,
                                                                 const uint8_t* __restrict__ mask,
                                                                 const float* __restrict__ mean2,
                                                                 const float* __restrict__ invstd2,
                                                                 const float* __restrict__ meand,
                                                                 const float* __restrict__ invstdd, long M, int C,
                                                                 long rows_per_block, float* __restrict__ partials) {
    constexpr int CH = Chunk<T>::N;
    const int tpr = C / CH, rpp = 256 / tpr;
    const int rg = threadIdx.x / tpr, cc = threadIdx.x % tpr;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float s1[CH], s2[CH], s3[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) s1[i] = s2[i] = s3[i] = 0.f;
    const int c0 = cc * CH;
    if (rg < rpp) {
        for (long r = r0 + rg; r < r1; r += rpp) {
            const long off = r * C + c0;
            float v2[CH], vd[CH], vg[CH];
            Chunk<T>::unpack(*(const u32x4*)(y2 + off), v2);
            Chunk<T>::unpack(*(const u32x4*)(yd + off), vd);
            Chunk<T>::unpack(*(const u32x4*)(dz + off), vg);
            const unsigned m = mask[off / CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                vg[i] = (m >> i) & 1u ? vg[i] : 0.f;
                const float xh2 = (v2[i] - mean2[c0 + i]) * invstd2[c0 + i];
                const float xhd = (vd[i] - meand[c0 + i]) * invstdd[c0 + i];
                s1[i] += vg[i];
                s2[i] += vg[i] * xh2;
                s3[i] += vg[i] * xhd;
            }
        }
    }
    __shared__ float red[3][256 * CH];
    if (rg < rpp) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            red[0][rg * C + c0 + i] = s1[i];
            red[1][rg * C + c0 + i] = s2[i];
            red[2][rg * C + c0 + i] = s3[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f, d = 0.f;
        for (int g = 0; g < rpp; ++g) {
            a += red[0][g * C + c];
            b += red[1][g * C + c];
            d += red[2][g * C + c];
        }
        partials[((long)blockIdx.x * 3 + 0) * C + c] = a;
        partials[((long)blockIdx.x * 3 + 1) * C + c] = b;
        partials[((long)blockIdx.x * 3 + 2) * C + c] = d;
    }
}

// the fp64 combine of bn_finalize_kernel (same slicing, same order) for three sums
__global__ __launch_bounds__(16 * kFinSlices) void bn_finalize3_kernel(const float* __restrict__ partials, int nblk, int C,
                                                                      float* dbeta2, float* dgamma2, float* dbetad,
                                                                      float* dgammad) {
    __shared__ double sa[kFinSlices][17], sb[kFinSlices][17], sc[kFinSlices][17];
    const int cl = threadIdx.x & 15, ks = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double acc3[3] = {0.0, 0.0, 0.0};
    if (c < C) fin_gather<3>(partials, nblk, C, c, ks, acc3);
    double a = acc3[0], b = acc3[1], d = acc3[2];
    sa[ks][cl] = a;
    sb[ks][cl] = b;
    sc[ks][cl] = d;
    __syncthreads();
    if (ks >= 4) return;
    for (int k = ks + 4; k < kFinSlices; k += 4) {
        a += sa[k][cl];
        b += sb[k][cl];
        d += sc[k][cl];
    }
    a += __shfl_down(a, 32);
    b += __shfl_down(b, 32);
    d += __shfl_down(d, 32);
    a += __shfl_down(a, 16);
    b += __shfl_down(b, 16);
    d += __shfl_down(d, 16);
    if (ks != 0 || c >= C) return;
    dbeta2[c] = (float)a;
    dgamma2[c] = (float)b;
    dbetad[c] = (float)a;
    dgammad[c] = (float)d;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_pair_apply_kernel(
    const T* __restrict__ y2, const T* __restrict__ yd, const T* __restrict__ dz, const uint8_t* __restrict__ mask,
    T* __restrict__ dy2, T* __restrict__ dyd, const float* __restrict__ gamma2, const float* __restrict__ mean2,
    const float* __restrict__ invstd2, const float* __restrict__ gammad, const float* __restrict__ meand,
    const float* __restrict__ invstdd, const float* __restrict__ dbeta, const float* __restrict__ dgamma2,
    const float* __restrict__ dgammad, float inv_m, long nchunks, int C) {
    constexpr int CH = Chunk<T>::N;
    __shared__ float sm[9][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        sm[0][c] = mean2[c];
        sm[1][c] = invstd2[c];
        sm[2][c] = gamma2[c] * invstd2[c];
        sm[3][c] = dbeta[c] * inv_m;
        sm[4][c] = dgamma2[c] * inv_m;
        sm[5][c] = meand[c];
        sm[6][c] = invstdd[c];
        sm[7][c] = gammad[c] * invstdd[c];
        sm[8][c] = dgammad[c] * inv_m;
    }
    __syncthreads();
    const int cpr = C / CH;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nchunks; q += stride) {
        const int c0 = (int)(q % cpr) * CH;
        float v2[CH], vd[CH], vg[CH];
        Chunk<T>::unpack(*(const u32x4*)(y2 + q * CH), v2);
        Chunk<T>::unpack(*(const u32x4*)(yd + q * CH), vd);
        Chunk<T>::unpack(*(const u32x4*)(dz + q * CH), vg);
        const unsigned m = mask[q];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            vg[i] = (m >> i) & 1u ? vg[i] : 0.f;
            const float xh2 = (v2[i] - sm[0][c0 + i]) * sm[1][c0 + i];
            const float xhd = (vd[i] - sm[5][c0 + i]) * sm[6][c0 + i];
            v2[i] = sm[2][c0 + i] * (vg[i] - sm[3][c0 + i] - xh2 * sm[4][c0 + i]);
            vd[i] = sm[7][c0 + i] * (vg[i] - sm[3][c0 + i] - xhd * sm[8][c0 + i]);
        }
        *(u32x4*)(dy2 + q * CH) = Chunk<T>::pack(v2);
        *(u32x4*)(dyd + q * CH) = Chunk<T>::pack(vd);
    }
}

template <typename T>
static int bn_bwd_pair_impl(const void* y2, const void* yd, const void* dz, const uint8_t* mask, void* dy2, void* dyd,
                            const float* gamma2, const float* mean2, const float* invstd2, const float* gammad,
                            const float* meand, const float* invstdd, float* dgamma2, float* dbeta2, float* dgammad,
                            float* dbetad, long M, int C, float* partials, hipStream_t st) {
    int nblk;
    long rpb;
    reduce_geometry(M, C, nblk, rpb);
    bn_bwd_pair_reduce_kernel<T><<<nblk, 256, 0, st>>>((const T*)y2, (const T*)yd, (const T*)dz, mask, mean2, invstd2,
                                                       meand, invstdd, M, C, rpb, partials);
    bn_finalize3_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(partials, nblk, C, dbeta2, dgamma2, dbetad, dgammad);
    const long nchunks = M * C / Chunk<T>::N;
    bn_bwd_pair_apply_kernel<T><<<stream_blocks(nchunks), 256, 0, st>>>(
        (const T*)y2, (const T*)yd, (const T*)dz, mask, (T*)dy2, (T*)dyd, gamma2, mean2, invstd2, gammad, meand,
        invstdd, dbeta2, dgamma2, dgammad, (float)(1.0 / (double)M), nchunks, C);
    return launch_status();
}

template <typename T>
static void launch_bn_relu_pool_fwd(const void* y, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                                    const float* mean, const float* invstd, int N, int H, int W, int C, int Ho, int Wo,
                                    hipStream_t st, int G = 0) {
    // Two windows along W per thread where the pooled row is even.  Variants measured and not kept (profiles/
    // r03_negative_results.txt): two windows along H as well (265 vs 258 us generic, 236 vs 225 us packed keys, batch
    // 256), whole output rows per block (no gain).
    const int pw = Wo % 2 == 0 ? 2 : 1;
    if constexpr (sizeof(T) == 2) {
        // bf16: packed (value, first position) keys — half the vector instructions of the generic form
        const long ktotal = (long)N * Ho * (Wo / pw) * (C / 8);
        const unsigned kgrid = (unsigned)((ktotal + 255) / 256);
        auto kk = pw == 2 ? bn_relu_pool_fwd_key_kernel<2, 1> : bn_relu_pool_fwd_key_kernel<1, 1>;
        kk<<<kgrid, 256, 0, st>>>((const bf16*)y, (bf16*)pooled, argmax, gamma, beta, mean, invstd, N, H, W, C, Ho, Wo, G);
        return;
    }
    const long total = (long)N * Ho * (Wo / pw) * (C / Chunk<T>::N);
    const unsigned grid = (unsigned)((total + 255) / 256);
    auto kern = pw == 2 ? bn_relu_pool_fwd_kernel<T, 2, 1> : bn_relu_pool_fwd_kernel<T, 1, 1>;
    kern<<<grid, 256, 0, st>>>((const T*)y, (T*)pooled, argmax, gamma, beta, mean, invstd, N, H, W, C, Ho, Wo, G);
}

// GroupNorm + ReLU + max-pool apply pass (statistics [N][G] already formed): csrc/gn.hip
void launch_gn_relu_pool_fwd(const void* y, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                             const float* mean, const float* invstd, int N, int H, int W, int C, int G, int dtype,
                             hipStream_t st) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (dtype == PRIMIA_F32)
        launch_bn_relu_pool_fwd<float>(y, pooled, argmax, gamma, beta, mean, invstd, N, H, W, C, Ho, Wo, st, G);
    else
        launch_bn_relu_pool_fwd<bf16>(y, pooled, argmax, gamma, beta, mean, invstd, N, H, W, C, Ho, Wo, st, G);
}

template <typename T>
static int bn_relu_pool_fwd_impl(const void* y, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float* save_mean, float* save_invstd, int N,
                                 int H, int W, int C, float eps, float momentum, float* partials, hipStream_t st) {
    const long M = (long)N * H * W;
    int nblk;
    long rpb;
    reduce_geometry(M, C, nblk, rpb);
    StatsFn<T> f{(const T*)y};
    colreduce2_kernel<T, StatsFn<T>><<<nblk, 256, 0, st>>>(f, M, C, rpb, partials);
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(partials, nblk, C, M, 0, eps, momentum, save_mean, save_invstd,
                                                       running_mean, running_var);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    launch_bn_relu_pool_fwd<T>(y, pooled, argmax, gamma, beta, save_mean, save_invstd, N, H, W, C, Ho, Wo, st);
    return launch_status();
}

// The same apply pass for even H and W, one thread per 2 x 2 block of input pixels (x one 16-byte channel chunk).
// Input pixel (2a + i, 2b + j) can only be the argmax of the windows (a + di, b + dj) with di <= i, dj <= j, at tap
// (1 + i - 2 di, 1 + j - 2 dj): the block shares 4 windows, loaded once (8 loads instead of up to 32 for four
// independent gathers), and needs 9 code comparisons instead of 16.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_apply2x2_kernel(
    const T* __restrict__ y, const T* __restrict__ dp, const uint8_t* __restrict__ argmax, T* __restrict__ dy,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_m,
    int N, int H, int W, int C, int Ho, int Wo) {
    constexpr int CH = Chunk<T>::N;
    __shared__ float sm[7][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        sm[0][c] = mean[c];
        sm[1][c] = invstd[c];
        sm[2][c] = gamma[c] * invstd[c];
        sm[3][c] = dbeta[c] * inv_m;
        sm[4][c] = dgamma[c] * inv_m;
        sm[5][c] = invstd[c] * gamma[c];
        sm[6][c] = beta[c];
    }
    __syncthreads();
    const int cpr = C / CH, H2 = H >> 1, W2 = W >> 1;
    const long total = (long)N * H2 * W2 * cpr;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total) return;
    const int c0 = (int)(q % cpr) * CH;
    long t = q / cpr;
    const int b = (int)(t % W2);
    t /= W2;
    const int a = (int)(t % H2), n = (int)(t / H2);
    // the 4 windows (a + di, b + dj), kept PACKED (24 registers; unpacked they would be 64 and the occupancy of this
    // latency-bound gather halves): gradient chunk and argmax codes (invalid windows: code 0xff never matches)
    u32x4 wraw[2][2];
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
            // same summation order as the generic gather: window row a + i first ... i.e. candidates A then B per axis
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
                const float zz = bn_affine(vy[k], sm[0][c0 + k], sm[5][c0 + k], sm[6][c0 + k]);
                const float gi = zz > 0.f ? g[k] : 0.f;
                const float xh = (vy[k] - sm[0][c0 + k]) * sm[1][c0 + k];
                vy[k] = sm[2][c0 + k] * (gi - sm[3][c0 + k] - xh * sm[4][c0 + k]);
            }
            *(u32x4*)(dy + off) = Chunk<T>::pack(vy);
            // keep the four pixels' work apart: hoisting all of it to the top costs 174 VGPRs (occupancy 2) and
            // this gather lives on waves in flight
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
}

template <typename T>
static int bn_relu_pool_bwd_impl(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax,
                                 void* dy, const float* gamma, const float* beta, const float* save_mean,
                                 const float* save_invstd, float* dgamma, float* dbeta, int N, int H, int W, int C,
                                 float* partials, hipStream_t st) {
    const long M = (long)N * H * W;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long Mp = (long)N * Ho * Wo;
    int nblk;
    long rpb;
    reduce_geometry(Mp, C, nblk, rpb);
    PoolScatterFn<T> f{(const T*)pooled, (const T*)dpooled, argmax, (const T*)y, save_mean, save_invstd, gamma, beta,
                       H, W, C, Ho, Wo, {}, {}, {}, {}, false};
    colreduce2_kernel<T, PoolScatterFn<T>><<<nblk, 256, 0, st>>>(f, Mp, C, rpb, partials);
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(partials, nblk, C, M, 1, 0.f, 0.f, dbeta, dgamma, nullptr, nullptr);
    if (!dy) return launch_status();   // sums only: the apply pass runs inside the consumer (primia_stem_bwd_fused)
    const long nchunks = M * C / Chunk<T>::N;
    if (H % 2 == 0 && W % 2 == 0) {
        const long total = nchunks / 4;
        bn_relu_pool_bwd_apply2x2_kernel<T><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
            (const T*)y, (const T*)dpooled, argmax, (T*)dy, gamma, beta, save_mean, save_invstd, dbeta, dgamma,
            (float)(1.0 / (double)M), N, H, W, C, Ho, Wo);
    } else {
        bn_relu_pool_bwd_apply_kernel<T><<<stream_blocks(nchunks) * 2, 256, 0, st>>>(
            (const T*)y, (const T*)dpooled, argmax, (T*)dy, gamma, beta, save_mean, save_invstd, dbeta, dgamma,
            (float)(1.0 / (double)M), N, H, W, C, Ho, Wo, 1.0f / (float)W, 1.0f / (float)H);
    }
    return launch_status();
}

}  // namespace primia

using namespace primia;

template <typename T>
static int bn_fwd_train_pair_impl(const void* y2, const void* yd, void* z, uint8_t* relu_mask, const float* gamma2,
                                  const float* beta2, float* rm2, float* rv2, float* sm2, float* si2, const float* sums2,
                                  int slots2, const float* gammad, const float* betad, float* rmd, float* rvd, float* smd,
                                  float* sid, const float* sumsd, int slotsd, long M, int C, float eps, float momentum,
                                  float* ws, hipStream_t st) {
    int nblk;
    long rpb;
    reduce_geometry(M, C, nblk, rpb);
    const float* partd = sumsd;
    int nd = slotsd;
    if (!sumsd) {
        StatsFn<T> fd{(const T*)yd};
        colreduce2_kernel<T, StatsFn<T>><<<nblk, 256, 0, st>>>(fd, M, C, rpb, ws);
        partd = ws;
        nd = nblk;
    }
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(partd, nd, C, M, 0, eps, momentum, smd, sid, rmd, rvd);
    const float* part = sums2;
    int n2 = slots2;
    if (!sums2) {
        StatsFn<T> f2{(const T*)y2};
        colreduce2_kernel<T, StatsFn<T>><<<nblk, 256, 0, st>>>(f2, M, C, rpb, ws);
        part = ws;
        n2 = nblk;
    }
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(part, n2, C, M, 0, eps, momentum, sm2, si2, rm2, rv2);
    const long nchunks = M * C / Chunk<T>::N;
    bn_apply_pair_kernel<T><<<stream_blocks(nchunks), 256, 0, st>>>((const T*)y2, (const T*)yd, (T*)z, gamma2, beta2, sm2,
                                                                    si2, gammad, betad, smd, sid, nchunks, C, relu_mask);
    return launch_status();
}

extern "C" {

int64_t primia_bn_workspace_bytes(int64_t M, int C) {
    (void)M;
    return (int64_t)kMaxPartialBlocks * 2 * C * sizeof(float);
}

int primia_bn_fwd_train(const void* y, const void* residual, void* z, const float* gamma,
                        const float* beta, float* running_mean, float* running_var,
                        float* save_mean, float* save_invstd, int64_t M, int C, float eps,
                        float momentum, int relu, void* workspace, int64_t workspace_bytes,
                        int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && z && gamma && beta && save_mean && save_invstd && workspace);
    PRIMIA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    if (workspace_bytes < primia_bn_workspace_bytes(M, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_fwd_train_impl<float>(y, residual, z, gamma, beta, running_mean, running_var, save_mean,
                                        save_invstd, M, C, eps, momentum, relu, (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return bn_fwd_train_impl<bf16>(y, residual, z, gamma, beta, running_mean, running_var, save_mean,
                                       save_invstd, M, C, eps, momentum, relu, (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

int primia_bn_fwd_train_from_sums(const void* y, const void* residual, void* z, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float* save_mean,
                                  float* save_invstd, const float* sums, int slots, int64_t M, int C, float eps,
                                  float momentum, int relu, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && z && gamma && beta && save_mean && save_invstd && sums && slots >= 1);
    PRIMIA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    hipStream_t st = (hipStream_t)stream;
    // `sums` has the layout of the partial blocks: [slots][2][C]
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(sums, slots, C, M, 0, eps, momentum, save_mean, save_invstd,
                                                       running_mean, running_var);
    if (dtype == PRIMIA_F32) {
        const long nchunks = M * C / 4;
        bn_apply_kernel<float><<<stream_blocks(nchunks), 256, 0, st>>>((const float*)y, (const float*)residual,
                                                                       (float*)z, gamma, beta, save_mean, save_invstd,
                                                                       eps, 0, nchunks, C, relu);
    } else if (dtype == PRIMIA_BF16) {
        const long nchunks = M * C / 8;
        bn_apply_kernel<bf16><<<stream_blocks(nchunks), 256, 0, st>>>((const bf16*)y, (const bf16*)residual, (bf16*)z,
                                                                      gamma, beta, save_mean, save_invstd, eps, 0,
                                                                      nchunks, C, relu);
    } else {
        return PRIMIA_ERR_ARG;
    }
    return launch_status();
}

// primia_bn_fwd_train_from_sums / primia_bn_fwd_train_mask with the finalize launch folded into the apply kernel
// (bn_inline_finalize_stats): same outputs, bit for bit.
int primia_bn_fwd_train_apply_inline(const void* y, const void* residual, void* z, uint8_t* relu_mask, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float* save_mean,
                                     float* save_invstd, const float* sums, int slots, int64_t M, int C, float eps,
                                     float momentum, int relu, uint32_t* flags, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && z && gamma && beta && save_mean && save_invstd && sums && slots >= 1 && flags);
    PRIMIA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    hipStream_t st = (hipStream_t)stream;
    const long nchunks = M * C / (dtype == PRIMIA_F32 ? 4 : 8);
    const int grid = stream_blocks(nchunks);
    if (relu_mask) relu = 1;
    if (grid < (C + 3) / 4) {      // fewer blocks than finalize slices (tiny tensors): the two-launch form
        bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(sums, slots, C, M, 0, eps, momentum, save_mean, save_invstd,
                                                           running_mean, running_var);
        if (dtype == PRIMIA_F32)
            bn_apply_kernel<float><<<grid, 256, 0, st>>>((const float*)y, (const float*)residual, (float*)z, gamma, beta, save_mean,
                                                          save_invstd, eps, 0, nchunks, C, relu, relu_mask);
        else
            bn_apply_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)y, (const bf16*)residual, (bf16*)z, gamma, beta, save_mean,
                                                         save_invstd, eps, 0, nchunks, C, relu, relu_mask);
        return launch_status();
    }
    BnInline q{sums, slots, (long)M, eps, momentum, running_mean, running_var, save_mean, save_invstd, flags};
    if (dtype == PRIMIA_F32)
        bn_apply_kernel<float><<<grid, 256, 0, st>>>((const float*)y, (const float*)residual, (float*)z, gamma, beta, save_mean,
                                                      save_invstd, eps, 0, nchunks, C, relu, relu_mask, q);
    else if (dtype == PRIMIA_BF16)
        bn_apply_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)y, (const bf16*)residual, (bf16*)z, gamma, beta, save_mean,
                                                     save_invstd, eps, 0, nchunks, C, relu, relu_mask, q);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

int primia_bn_fwd_eval(const void* y, const void* residual, void* z, const float* gamma,
                       const float* beta, const float* running_mean, const float* running_var,
                       int64_t M, int C, float eps, int relu, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && z && gamma && beta && running_mean && running_var);
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32) {
        const long nchunks = M * C / 4;
        bn_apply_kernel<float><<<stream_blocks(nchunks), 256, 0, st>>>(
            (const float*)y, (const float*)residual, (float*)z, gamma, beta, running_mean, running_var, eps, 1,
            nchunks, C, relu);
    } else if (dtype == PRIMIA_BF16) {
        const long nchunks = M * C / 8;
        bn_apply_kernel<bf16><<<stream_blocks(nchunks), 256, 0, st>>>(
            (const bf16*)y, (const bf16*)residual, (bf16*)z, gamma, beta, running_mean, running_var, eps, 1,
            nchunks, C, relu);
    } else {
        return PRIMIA_ERR_ARG;
    }
    return launch_status();
}

int primia_bn_bwd(const void* y, const void* z, const void* dz, void* dy, void* g_out,
                  const float* gamma, const float* save_mean, const float* save_invstd,
                  float* dgamma, float* dbeta, int64_t M, int C, int relu, void* workspace,
                  int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && dz && dy && gamma && save_mean && save_invstd && dgamma && dbeta && workspace);
    PRIMIA_REQUIRE(!relu || z);
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    if (workspace_bytes < primia_bn_workspace_bytes(M, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_bwd_impl<float>(y, z, dz, dy, g_out, gamma, save_mean, save_invstd, dgamma, dbeta, M, C, relu,
                                  (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return bn_bwd_impl<bf16>(y, z, dz, dy, g_out, gamma, save_mean, save_invstd, dgamma, dbeta, M, C, relu,
                                 (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}


int primia_bn_relu_maxpool_fwd(const void* y, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* save_mean, float* save_invstd, int N,
                               int H, int W, int C, float eps, float momentum, void* workspace,
                               int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && pooled && argmax && gamma && beta && save_mean && save_invstd && workspace);
    PRIMIA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    PRIMIA_REQUIRE(N > 0 && H > 0 && W > 0 && bn_shape_ok((long)N * H * W, C, dtype));
    if (workspace_bytes < primia_bn_workspace_bytes((int64_t)N * H * W, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_relu_pool_fwd_impl<float>(y, pooled, argmax, gamma, beta, running_mean, running_var, save_mean,
                                            save_invstd, N, H, W, C, eps, momentum, (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return bn_relu_pool_fwd_impl<bf16>(y, pooled, argmax, gamma, beta, running_mean, running_var, save_mean,
                                           save_invstd, N, H, W, C, eps, momentum, (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

int primia_bn_finalize_stats(const float* sums, int slots, int64_t M, int C, float eps, float momentum, float* running_mean,
                             float* running_var, float* save_mean, float* save_invstd, primia_stream_t stream) {
    PRIMIA_REQUIRE(sums && slots >= 1 && M > 0 && C > 0 && save_mean && save_invstd);
    PRIMIA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, (hipStream_t)stream>>>(sums, slots, C, M, 0, eps, momentum, save_mean,
                                                                                  save_invstd, running_mean, running_var);
    return launch_status();
}

int primia_bn_relu_maxpool_fwd_from_sums(const void* y, void* pooled, uint8_t* argmax, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var,
                                         float* save_mean, float* save_invstd, const float* sums, int slots, int N,
                                         int H, int W, int C, float eps, float momentum, int dtype,
                                         primia_stream_t stream) {
    PRIMIA_REQUIRE(y && pooled && argmax && gamma && beta && save_mean && save_invstd && sums && slots >= 1);
    PRIMIA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    PRIMIA_REQUIRE(N > 0 && H > 0 && W > 0 && bn_shape_ok((long)N * H * W, C, dtype));
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)N * H * W;
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(sums, slots, C, M, 0, eps, momentum, save_mean, save_invstd,
                                                       running_mean, running_var);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (dtype == PRIMIA_F32) {
        launch_bn_relu_pool_fwd<float>(y, pooled, argmax, gamma, beta, save_mean, save_invstd, N, H, W, C, Ho, Wo, st);
    } else if (dtype == PRIMIA_BF16) {
        launch_bn_relu_pool_fwd<bf16>(y, pooled, argmax, gamma, beta, save_mean, save_invstd, N, H, W, C, Ho, Wo, st);
    } else {
        return PRIMIA_ERR_ARG;
    }
    return launch_status();
}

int primia_bn_relu_maxpool_bwd(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax,
                               void* dy, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_invstd, float* dgamma, float* dbeta, int N, int H, int W, int C,
                               void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && pooled && dpooled && argmax && gamma && beta && save_mean && save_invstd && dgamma &&
                   dbeta && workspace);   // dy may be null: dgamma / dbeta only (see primia_stem_bwd_fused)
    PRIMIA_REQUIRE(N > 0 && H > 0 && W > 0 && bn_shape_ok((long)N * H * W, C, dtype));
    PRIMIA_REQUIRE((long)N * H * W < (1L << 31));
    if (workspace_bytes < primia_bn_workspace_bytes((int64_t)N * H * W, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_relu_pool_bwd_impl<float>(y, pooled, dpooled, argmax, dy, gamma, beta, save_mean, save_invstd, dgamma,
                                            dbeta, N, H, W, C, (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return bn_relu_pool_bwd_impl<bf16>(y, pooled, dpooled, argmax, dy, gamma, beta, save_mean, save_invstd, dgamma,
                                           dbeta, N, H, W, C, (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

// primia_bn_relu_maxpool_bwd(dy = null) whose reduction pass over (pooled, dpooled) already happened in the write-back of the data
// gradient that produced dpooled (primia_conv2d_dgrad_masked_acc_bnsums, mode 3): `sums` = its partials [slots][2][C].
int primia_bn_relu_maxpool_bwd_from_sums(const void* y, const void* pooled, const void* dpooled, const uint8_t* argmax,
                                         const float* gamma, const float* beta, const float* save_mean,
                                         const float* save_invstd, float* dgamma, float* dbeta, const float* sums, int slots,
                                         int N, int H, int W, int C, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && pooled && dpooled && argmax && gamma && beta && save_mean && save_invstd && dgamma && dbeta && sums);
    PRIMIA_REQUIRE(N > 0 && H > 0 && W > 0 && slots >= 1 && bn_shape_ok((long)N * H * W, C, dtype));
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_BF16)
        bn_finalize_pool_kernel<bf16><<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(
            sums, slots, C, dbeta, dgamma, gamma, beta, save_mean, save_invstd, (const bf16*)pooled, (const bf16*)dpooled, argmax,
            (const bf16*)y, N, H, W, Ho, Wo);
    else if (dtype == PRIMIA_F32)
        bn_finalize_pool_kernel<float><<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(
            sums, slots, C, dbeta, dgamma, gamma, beta, save_mean, save_invstd, (const float*)pooled, (const float*)dpooled, argmax,
            (const float*)y, N, H, W, Ho, Wo);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

int primia_bn_fwd_train_mask(const void* y, const void* residual, void* z, uint8_t* relu_mask, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float* save_mean,
                             float* save_invstd, const float* sums, int slots, int64_t M, int C, float eps,
                             float momentum, void* workspace, int64_t workspace_bytes, int dtype,
                             primia_stream_t stream) {
    PRIMIA_REQUIRE(y && z && relu_mask && gamma && beta && save_mean && save_invstd && (sums || workspace));
    PRIMIA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype) && (!sums || slots >= 1));
    if (!sums && workspace_bytes < primia_bn_workspace_bytes(M, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const float* part = sums;
    int nblk = slots;
    if (!sums) {
        long rpb;
        reduce_geometry(M, C, nblk, rpb);
        part = (const float*)workspace;
        if (dtype == PRIMIA_F32) {
            StatsFn<float> f{(const float*)y};
            colreduce2_kernel<float, StatsFn<float>><<<nblk, 256, 0, st>>>(f, M, C, rpb, (float*)workspace);
        } else {
            StatsFn<bf16> f{(const bf16*)y};
            colreduce2_kernel<bf16, StatsFn<bf16>><<<nblk, 256, 0, st>>>(f, M, C, rpb, (float*)workspace);
        }
    }
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(part, nblk, C, M, 0, eps, momentum, save_mean, save_invstd,
                                                       running_mean, running_var);
    if (dtype == PRIMIA_F32) {
        const long nchunks = M * C / 4;
        bn_apply_kernel<float><<<stream_blocks(nchunks), 256, 0, st>>>((const float*)y, (const float*)residual,
                                                                       (float*)z, gamma, beta, save_mean, save_invstd,
                                                                       eps, 0, nchunks, C, 1, relu_mask);
    } else {
        const long nchunks = M * C / 8;
        bn_apply_kernel<bf16><<<stream_blocks(nchunks), 256, 0, st>>>((const bf16*)y, (const bf16*)residual, (bf16*)z,
                                                                      gamma, beta, save_mean, save_invstd, eps, 0,
                                                                      nchunks, C, 1, relu_mask);
    }
    return launch_status();
}

int primia_bn_fwd_train_pair(const void* y2, const void* yd, void* z, uint8_t* relu_mask, const float* gamma2,
                             const float* beta2, float* running_mean2, float* running_var2, float* save_mean2,
                             float* save_invstd2, const float* sums2, int slots2, const float* gamma_d,
                             const float* beta_d, float* running_mean_d, float* running_var_d, float* save_mean_d,
                             float* save_invstd_d, const float* sums_d, int slots_d, int64_t M, int C, float eps,
                             float momentum, void* workspace, int64_t workspace_bytes, int dtype,
                             primia_stream_t stream) {
    PRIMIA_REQUIRE(y2 && yd && z && relu_mask && gamma2 && beta2 && save_mean2 && save_invstd2 && gamma_d && beta_d &&
                   save_mean_d && save_invstd_d && workspace);
    PRIMIA_REQUIRE((running_mean2 == nullptr) == (running_var2 == nullptr));
    PRIMIA_REQUIRE((running_mean_d == nullptr) == (running_var_d == nullptr));
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype) && (!sums2 || slots2 >= 1) && (!sums_d || slots_d >= 1));
    if (workspace_bytes < primia_bn_workspace_bytes(M, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_fwd_train_pair_impl<float>(y2, yd, z, relu_mask, gamma2, beta2, running_mean2, running_var2, save_mean2,
                                             save_invstd2, sums2, slots2, gamma_d, beta_d, running_mean_d, running_var_d,
                                             save_mean_d, save_invstd_d, sums_d, slots_d, M, C, eps, momentum,
                                             (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return bn_fwd_train_pair_impl<bf16>(y2, yd, z, relu_mask, gamma2, beta2, running_mean2, running_var2, save_mean2,
                                            save_invstd2, sums2, slots2, gamma_d, beta_d, running_mean_d, running_var_d,
                                            save_mean_d, save_invstd_d, sums_d, slots_d, M, C, eps, momentum,
                                            (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

int primia_bn_bwd_mask(const void* y, const uint8_t* relu_mask, const void* dz, void* dy, void* g_out,
                       const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma,
                       float* dbeta, int64_t M, int C, void* workspace, int64_t workspace_bytes, int dtype,
                       primia_stream_t stream) {
    PRIMIA_REQUIRE(y && relu_mask && dz && dy && gamma && save_mean && save_invstd && dgamma && dbeta && workspace);
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    if (workspace_bytes < primia_bn_workspace_bytes(M, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_bwd_impl<float>(y, nullptr, dz, dy, g_out, gamma, save_mean, save_invstd, dgamma, dbeta, M, C, 0,
                                  (float*)workspace, st, nullptr, relu_mask);
    if (dtype == PRIMIA_BF16)
        return bn_bwd_impl<bf16>(y, nullptr, dz, dy, g_out, gamma, save_mean, save_invstd, dgamma, dbeta, M, C, 0,
                                 (float*)workspace, st, nullptr, relu_mask);
    return PRIMIA_ERR_ARG;
}

int primia_bn_bwd_pair(const void* y2, const void* yd, const void* dz, const uint8_t* relu_mask, void* dy2, void* dyd,
                       const float* gamma2, const float* save_mean2, const float* save_invstd2, const float* gamma_d,
                       const float* save_mean_d, const float* save_invstd_d, float* dgamma2, float* dbeta2,
                       float* dgamma_d, float* dbeta_d, int64_t M, int C, void* workspace, int64_t workspace_bytes,
                       int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y2 && yd && dz && relu_mask && dy2 && dyd && gamma2 && save_mean2 && save_invstd2 && gamma_d &&
                   save_mean_d && save_invstd_d && dgamma2 && dbeta2 && dgamma_d && dbeta_d && workspace);
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    if (workspace_bytes < (int64_t)kMaxPartialBlocks * 3 * C * (int64_t)sizeof(float)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_bwd_pair_impl<float>(y2, yd, dz, relu_mask, dy2, dyd, gamma2, save_mean2, save_invstd2, gamma_d,
                                       save_mean_d, save_invstd_d, dgamma2, dbeta2, dgamma_d, dbeta_d, M, C,
                                       (float*)workspace, st);
    if (dtype == PRIMIA_BF16)
        return bn_bwd_pair_impl<bf16>(y2, yd, dz, relu_mask, dy2, dyd, gamma2, save_mean2, save_invstd2, gamma_d,
                                      save_mean_d, save_invstd_d, dgamma2, dbeta2, dgamma_d, dbeta_d, M, C,
                                      (float*)workspace, st);
    return PRIMIA_ERR_ARG;
}

int primia_bn_relu_bwd(const void* y, const void* dz, void* dy, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, int64_t M,
                       int C, void* workspace, int64_t workspace_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && dz && dy && gamma && beta && save_mean && save_invstd && dgamma && dbeta && workspace);
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    if (workspace_bytes < primia_bn_workspace_bytes(M, C)) return PRIMIA_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        return bn_bwd_impl<float>(y, nullptr, dz, dy, nullptr, gamma, save_mean, save_invstd, dgamma, dbeta, M, C, 0,
                                  (float*)workspace, st, beta);
    if (dtype == PRIMIA_BF16)
        return bn_bwd_impl<bf16>(y, nullptr, dz, dy, nullptr, gamma, save_mean, save_invstd, dgamma, dbeta, M, C, 0,
                                 (float*)workspace, st, beta);
    return PRIMIA_ERR_ARG;
}

// primia_bn_relu_bwd with the reduction pass already done by the producer of dz (primia_conv2d_dgrad_bnsums): `sums` =
// [slots][2][C] partials of (sum g, sum g * xhat); finalize + apply pass only.
int primia_bn_relu_bwd_from_sums(const void* y, const void* dz, void* dy, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                 const float* sums, int slots, int64_t M, int C, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(y && dz && dy && gamma && beta && save_mean && save_invstd && dgamma && dbeta && sums && slots >= 1);
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    hipStream_t st = (hipStream_t)stream;
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(sums, slots, C, M, 1, 0.f, 0.f, dbeta, dgamma, nullptr, nullptr);
    const float inv_m = (float)(1.0 / (double)M);
    if (dtype == PRIMIA_F32) {
        const long nchunks = M * C / 4;
        bn_bwd_apply_kernel<float><<<stream_blocks(nchunks), 256, 0, st>>>((const float*)y, nullptr, (const float*)dz, (float*)dy,
                                                                           nullptr, gamma, save_mean, save_invstd, dbeta, dgamma,
                                                                           inv_m, nchunks, C, beta, nullptr);
    } else if (dtype == PRIMIA_BF16) {
        const long nchunks = M * C / 8;
        if (PRIMIA_OPT(bn_unroll) == 2 && nchunks >= 4L * 2048 * 256)
            bn_bwd_apply_kernel<bf16, 2><<<stream_blocks(nchunks), 256, 0, st>>>((const bf16*)y, nullptr, (const bf16*)dz, (bf16*)dy,
                                                                                 nullptr, gamma, save_mean, save_invstd, dbeta,
                                                                                 dgamma, inv_m, nchunks, C, beta, nullptr);
        else
            bn_bwd_apply_kernel<bf16><<<stream_blocks(nchunks), 256, 0, st>>>((const bf16*)y, nullptr, (const bf16*)dz, (bf16*)dy,
                                                                              nullptr, gamma, save_mean, save_invstd, dbeta, dgamma,
                                                                              inv_m, nchunks, C, beta, nullptr);
    } else {
        return PRIMIA_ERR_ARG;
    }
    return launch_status();
}

// primia_bn_bwd_mask with the reduction pass already done by the producer of dz (primia_conv2d_dgrad_pair_bnsums).
int primia_bn_bwd_mask_from_sums(const void* y, const uint8_t* relu_mask, const void* dz, void* dy, void* g_out,
                                 const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma,
                                 float* dbeta, const float* sums, int slots, int64_t M, int C, int dtype,
                                 primia_stream_t stream) {
    PRIMIA_REQUIRE(y && relu_mask && dz && dy && gamma && save_mean && save_invstd && dgamma && dbeta && sums && slots >= 1);
    PRIMIA_REQUIRE(bn_shape_ok(M, C, dtype));
    hipStream_t st = (hipStream_t)stream;
    bn_finalize_kernel<<<(C + 15) / 16, 16 * kFinSlices, 0, st>>>(sums, slots, C, M, 1, 0.f, 0.f, dbeta, dgamma, nullptr, nullptr);
    const float inv_m = (float)(1.0 / (double)M);
    if (dtype == PRIMIA_F32) {
        const long nchunks = M * C / 4;
        bn_bwd_apply_kernel<float><<<stream_blocks(nchunks), 256, 0, st>>>((const float*)y, nullptr, (const float*)dz, (float*)dy,
                                                                           (float*)g_out, gamma, save_mean, save_invstd, dbeta,
                                                                           dgamma, inv_m, nchunks, C, nullptr, relu_mask);
    } else if (dtype == PRIMIA_BF16) {
        const long nchunks = M * C / 8;
        if (PRIMIA_OPT(bn_unroll) == 2 && nchunks >= 4L * 2048 * 256)
            bn_bwd_apply_kernel<bf16, 2><<<stream_blocks(nchunks), 256, 0, st>>>((const bf16*)y, nullptr, (const bf16*)dz, (bf16*)dy,
                                                                                 (bf16*)g_out, gamma, save_mean, save_invstd, dbeta,
                                                                                 dgamma, inv_m, nchunks, C, nullptr, relu_mask);
        else
            bn_bwd_apply_kernel<bf16><<<stream_blocks(nchunks), 256, 0, st>>>((const bf16*)y, nullptr, (const bf16*)dz, (bf16*)dy,
                                                                              (bf16*)g_out, gamma, save_mean, save_invstd, dbeta,
                                                                              dgamma, inv_m, nchunks, C, nullptr, relu_mask);
    } else {
        return PRIMIA_ERR_ARG;
    }
    return launch_status();
}

}  // extern "C"
