// BatchNorm2d training/eval forward and backward on [M, C] (NHWC) tensors, fused with ReLU and
// the residual add.  All kernels are HBM streaming passes with 16-byte accesses; the per-channel
// reductions are two-level and deterministic: each block reduces a contiguous slab of rows to an
// fp32 partial (wavefront-local accumulation, LDS tree across row groups), a one-block finalize
// kernel combines the <= 1024 partials per channel in fp64.
//
// Reference semantics: torch.nn.functional.batch_norm as called from nn.BatchNorm2d
// (torchlib/models.py:261-264, 382): biased variance for normalisation, unbiased for the running
// estimate, momentum 0.1, eps 1e-5.
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
template <typename T, typename F>
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
        for (long r = r0 + rg; r < r1; r += rpp) f(r * C + cc * CH, cc * CH, s1, s2);
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
    const T* y;
    __device__ __forceinline__ void operator()(long off, int c0, float* s1, float* s2) const {
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
    const T* y;
    const T* z;   // may be null (no relu)
    const T* dz;
    const float* mean;
    const float* invstd;
    const float* gamma;  // with beta: no z, the mask (z > 0) is recomputed from y
    const float* beta;
    __device__ __forceinline__ void operator()(long off, int c0, float* s1, float* s2) const {
        constexpr int CH = Chunk<T>::N;
        float vy[CH], vg[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + off), vy);
        Chunk<T>::unpack(*(const u32x4*)(dz + off), vg);
        if (z) {
            float vz[CH];
            Chunk<T>::unpack(*(const u32x4*)(z + off), vz);
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = vz[i] > 0.f ? vg[i] : 0.f;
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

// Combine the per-block partials in fp64.  Block = 16 channels x 16 partial-slices; the slices are
// reduced through LDS.  mode 0: batch statistics -> mean / invstd / running stats.
// mode 1: backward sums -> dbeta (sum g) / dgamma (sum g*xhat).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int nblk,
                                                          int C, long M, int mode, float eps,
                                                          float momentum, float* out1, float* out2,
                                                          float* running_mean, float* running_var) {
    __shared__ double sa[16][17], sb[16][17];
    const int cl = threadIdx.x & 15, ks = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int k = ks; k < nblk; k += 16) {
            a += (double)partials[((long)k * 2 + 0) * C + c];
            b += (double)partials[((long)k * 2 + 1) * C + c];
        }
    }
    sa[ks][cl] = a;
    sb[ks][cl] = b;
    __syncthreads();
    if (ks != 0 || c >= C) return;
    for (int k = 1; k < 16; ++k) {
        a += sa[k][cl];
        b += sb[k][cl];
    }
    if (mode == 0) {
        const double mean = a / (double)M;
        double var = b / (double)M - mean * mean;
        if (var < 0.0) var = 0.0;
        out1[c] = (float)mean;
        out2[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + (double)momentum * unb);
        }
    } else {
        out1[c] = (float)a;  // dbeta
        out2[c] = (float)b;  // dgamma
    }
}

// z = act((y - mean) * (invstd * gamma) + beta [+ residual])
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ res,
                                                       T* __restrict__ z, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd_or_var, float eps,
                                                       int eval_mode, long nchunks, int C, int relu) {
    constexpr int CH = Chunk<T>::N;
    __shared__ float sm[3][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        const float is = eval_mode ? 1.f / sqrtf(invstd_or_var[c] + eps) : invstd_or_var[c];
        sm[0][c] = mean[c];
        sm[1][c] = is * gamma[c];
        sm[2][c] = beta[c];
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
        *(u32x4*)(z + q * CH) = Chunk<T>::pack(v);
    }
}

// dy = gamma*invstd * (g - dbeta/M - xhat*dgamma/M); optionally g_out = g.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ y, const T* __restrict__ z,
                                                           const T* dz, T* __restrict__ dy, T* g_out,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, float inv_m,
                                                           long nchunks, int C, const float* __restrict__ beta) {
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
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nchunks; q += stride) {
        const int c0 = (int)(q % cpr) * CH;
        float vy[CH], vg[CH];
        Chunk<T>::unpack(*(const u32x4*)(y + q * CH), vy);
        Chunk<T>::unpack(*(const u32x4*)(dz + q * CH), vg);
        if (z) {
            float vz[CH];
            Chunk<T>::unpack(*(const u32x4*)(z + q * CH), vz);
#pragma unroll
            for (int i = 0; i < CH; ++i) vg[i] = vz[i] > 0.f ? vg[i] : 0.f;
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

static inline void reduce_geometry(long M, int C, int& nblk, long& rows_per_block) {
    // >= 256 rows per block so the fp32 in-block accumulation stays short; <= 1024 blocks.
    long nb = (M + 255) / 256;
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
    bn_finalize_kernel<<<(C + 15) / 16, 256, 0, st>>>(partials, nblk, C, M, 0, eps, momentum, save_mean, save_invstd,
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
                       hipStream_t st, const float* beta = nullptr) {
    int nblk;
    long rpb;
    reduce_geometry(M, C, nblk, rpb);
    BwdFn<T> f{(const T*)y, relu ? (const T*)z : nullptr, (const T*)dz, save_mean, save_invstd, gamma, beta};
    colreduce2_kernel<T, BwdFn<T>><<<nblk, 256, 0, st>>>(f, M, C, rpb, partials);
    bn_finalize_kernel<<<(C + 15) / 16, 256, 0, st>>>(partials, nblk, C, M, 1, 0.f, 0.f, dbeta, dgamma, nullptr, nullptr);
    const long nchunks = M * C / Chunk<T>::N;
    bn_bwd_apply_kernel<T><<<stream_blocks(nchunks), 256, 0, st>>>(
        (const T*)y, relu ? (const T*)z : nullptr, (const T*)dz, (T*)dy, (T*)g_out, gamma, save_mean,
        save_invstd, dbeta, dgamma, (float)(1.0 / (double)M), nchunks, C, beta);
    return launch_status();
}

}  // namespace primia

using namespace primia;

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
    bn_finalize_kernel<<<(C + 15) / 16, 256, 0, st>>>(sums, slots, C, M, 0, eps, momentum, save_mean, save_invstd,
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

}  // extern "C"
