// Boundary layout conversions: NCHW fp32 <-> NHWC compute dtype, OIHW fp32 master weights ->
// compute-dtype implicit-GEMM layouts, and the wgrad accumulator -> OIHW gradient transpose.
// All HBM-bound; sizes here are tiny next to the activations (11 M weights, one input batch).
#include "common.h"
#include "conv_common.h"

namespace primia {

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src,
                                                           T* __restrict__ dst, int C, int HW,
                                                           int c_pad, long total) {
    // One thread per (n, hw): reads C strided values (coalesced across hw), writes c_pad contiguous.
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long n = i / HW;
    long hw = i - n * HW;
    const float* s = src + n * (long)C * HW + hw;
    T* d = dst + i * c_pad;
    for (int c = 0; c < c_pad; ++c) Elem<T>::store(d + c, c < C ? s[(long)c * HW] : 0.f);
}

// Same, into a spatially padded destination [N][Hp][Wp][c_pad] (interior at (pad_top, pad_left)); the
// padding itself is never written — the caller zeroes the buffer once.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_padded_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                                  int C, int H, int W, int c_pad, int pad_top,
                                                                  int pad_left, int Hp, int Wp, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int HW = H * W;
    const long n = i / HW;
    const int hw = (int)(i - n * HW);
    const int h = hw / W, w = hw - h * W;
    const float* s = src + n * (long)C * HW + hw;
    T* d = dst + ((n * Hp + h + pad_top) * Wp + w + pad_left) * c_pad;
    for (int c = 0; c < c_pad; ++c) Elem<T>::store(d + c, c < C ? s[(long)c * HW] : 0.f);
}

// bf16, 4 channels, W % 4 == 0: one thread converts 4 consecutive pixels (a float4 per channel plane in, 32
// contiguous bytes out) — 4x fewer threads and 16-byte loads for the same bytes
__global__ __launch_bounds__(256) void nchw_to_nhwc4_padded_x4_kernel(const float* __restrict__ src,
                                                                      bf16* __restrict__ dst, int C, int H, int W,
                                                                      int pad_top, int pad_left, int Hp, int Wp,
                                                                      long total4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int W4 = W >> 2, HW = H * W;
    const long n = i / ((long)H * W4);
    const int r = (int)(i - n * (long)H * W4);
    const int h = r / W4, w = (r - h * W4) * 4;
    const float* s = src + n * (long)C * HW + (long)h * W + w;
    f32x4 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = c < C ? *(const f32x4*)(s + (long)c * HW) : f32x4{0.f, 0.f, 0.f, 0.f};
    u32x2* d = (u32x2*)(dst + ((n * Hp + h + pad_top) * Wp + w + pad_left) * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u32x2 o;
        o[0] = (uint32_t)f32_to_bf16(v[0][k]) | ((uint32_t)f32_to_bf16(v[1][k]) << 16);
        o[1] = (uint32_t)f32_to_bf16(v[2][k]);
        d[k] = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src,
                                                           float* __restrict__ dst, int C, int HW,
                                                           int c_pad, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long n = i / HW;
    long hw = i - n * HW;
    const T* s = src + i * c_pad;
    float* d = dst + n * (long)C * HW + hw;
    for (int c = 0; c < C; ++c) d[(long)c * HW] = Elem<T>::load(s + c);
}

template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src,
                                                            T* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) Elem<T>::store(dst + i, src[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* __restrict__ src,
                                                          float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = Elem<T>::load(src + i);
}

// One thread per element of the fwd-layout copy [K][klen] (klen includes zero padding).
template <typename T>
__global__ __launch_bounds__(256) void weight_fwd_kernel(const float* __restrict__ w_oihw,
                                                         T* __restrict__ w_fwd, ConvGeom g,
                                                         int c_real, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int k = (int)(i / g.klen);
    int e = (int)(i - (long)k * g.klen);
    int r, s, c;
    g.decode_k(e, r, s, c);
    float v = 0.f;
    if (r < g.R && s < g.S && c < c_real)
        v = w_oihw[(((long)k * c_real + c) * g.R + r) * g.S + s];
    Elem<T>::store(w_fwd + i, v);
}

// dgrad layout [C][R][S][K]: one thread per element.
template <typename T>
__global__ __launch_bounds__(256) void weight_dgrad_kernel(const float* __restrict__ w_oihw,
                                                           T* __restrict__ w_dg, int K, int C,
                                                           int R, int S, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int k = (int)(i % K);
    long t = i / K;
    int s = (int)(t % S);
    t /= S;
    int r = (int)(t % R);
    int c = (int)(t / R);
    Elem<T>::store(w_dg + i, w_oihw[(((long)k * C + c) * R + r) * S + s]);
}

// dw_oihw[k][c][r][s] = dw_acc[k][e(r,s,c)].
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float* __restrict__ dw_acc,
                                                             float* __restrict__ dw_oihw,
                                                             ConvGeom g, int c_real, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int s = (int)(i % g.S);
    long t = i / g.S;
    int r = (int)(t % g.R);
    t /= g.R;
    int c = (int)(t % c_real);
    int k = (int)(t / c_real);
    dw_oihw[i] = dw_acc[(long)k * g.klen + g.encode_k(r, s, c)];
}

// ---- batched weight refresh / gradient finalize: one launch for the whole network ------------------
constexpr int kMaxConvs = 24;
struct ManyEntry {
    ConvGeom g;
    int c_real;
    const float* src;  // w_oihw | dw_acc
    void* dst;         // w_fwd  | dw_oihw
    void* dst2;        // w_dgrad or null
    long begin;        // first flat work index of this entry
};
struct ManyArgs {
    ManyEntry e[kMaxConvs];
    int n;
    long total;
};

template <typename T>
__global__ __launch_bounds__(256) void weight_prepare_many_kernel(ManyArgs a) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += stride) {
        int c = 0;
        while (c + 1 < a.n && i >= a.e[c + 1].begin) ++c;
        const ManyEntry& en = a.e[c];
        const long j = i - en.begin;
        const long nfwd = (long)en.g.K * en.g.klen;
        if (j < nfwd) {
            const int k = (int)(j / en.g.klen), e = (int)(j - (long)k * en.g.klen);
            int r, s, ch;
            en.g.decode_k(e, r, s, ch);
            float v = 0.f;
            if (r < en.g.R && s < en.g.S && ch < en.c_real)
                v = en.src[(((long)k * en.c_real + ch) * en.g.R + r) * en.g.S + s];
            Elem<T>::store((T*)en.dst + j, v);
        } else {
            // dgrad layout [C][R][S][K]
            const long q = j - nfwd;
            const int K = en.g.K, C = en.g.C, R = en.g.R, S = en.g.S;
            const int k = (int)(q % K);
            long t = q / K;
            const int s = (int)(t % S);
            t /= S;
            const int r = (int)(t % R);
            const int ch = (int)(t / R);
            Elem<T>::store((T*)en.dst2 + q, en.src[(((long)k * C + ch) * R + r) * S + s]);
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_finalize_many_kernel(ManyArgs a) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += stride) {
        int c = 0;
        while (c + 1 < a.n && i >= a.e[c + 1].begin) ++c;
        const ManyEntry& en = a.e[c];
        const long j = i - en.begin;
        const int s = (int)(j % en.g.S);
        long t = j / en.g.S;
        const int r = (int)(t % en.g.R);
        t /= en.g.R;
        const int ch = (int)(t % en.c_real);
        const int k = (int)(t / en.c_real);
        ((float*)en.dst)[j] = en.src[(long)k * en.g.klen + en.g.encode_k(r, s, ch)];
    }
}

// ---- tiled forms for every regular conv (C % 64 == 0, c_real == C): one block moves a 32 out-chan x 64
// in-chan x R*S tile through LDS, so both the OIHW side (contiguous in (c, r, s) per out-chan) and the
// kernel-layout side (contiguous in c per tap, or in k for the dgrad copy) are accessed in whole rows.
// The element-wise kernels above gather with a stride of R*S floats and spend ~100 instructions of index
// arithmetic per element (193 us for the network's 11 M weights; this form: see profiles/).
// Out-channels per tile: 16 for the 3x3 layers (37 KB of LDS, four blocks per CU, ~1,200 tiles for ResNet-18) — with 32
// (74 KB, two blocks per CU, ~600 tiles) a pass was a chain of memory round trips on too few blocks in flight (the fused
// step tail below: 69 -> 43 us) — and 32 for the 1x1 layers.
constexpr int kTileK9 = 16;
static inline int tiles_of(int K, int C, int RS) { return (K / (RS == 9 ? kTileK9 : 32)) * (C / 64); }

struct TileArgs {
    ManyEntry e[kMaxConvs];
    int tile_begin[kMaxConvs + 1];
    int n;
};

// (16-byte loads on the fp32 side and 4-byte stores of channel / out-channel pairs on the bf16 side: the first
// version moved one element per instruction — 72 dependent-free but single-element trips per thread — and ran at
// 2.4 TB/s on 90 MB)
// LDS tile [32 out-channels][64 in-channels * RS taps] in OIHW order -> the forward layout [K][tap][C] and (if dst2) the
// data-gradient layout [C][R][S][K], in the compute dtype
template <typename T, int RS, int KT = 32>
__device__ __forceinline__ void prepare_tile_emit(void* dst_, void* dst2_, int klen, int C, int K, int k0, int c0,
                                                  const float* lds) {
    constexpr int ROW = 64 * RS, PITCH = ROW + 1;
    T* dst = (T*)dst_;
    if constexpr (sizeof(T) == 2) {
        for (int idx = threadIdx.x; idx < KT * ROW / 2; idx += 256) {
            const int k = idx / (ROW / 2), j = (idx - k * (ROW / 2)) * 2;     // j = t * 64 + c, c even
            const int t = j >> 6, c = j & 63;
            const uint32_t lo = f32_to_bf16(lds[k * PITCH + c * RS + t]), hi = f32_to_bf16(lds[k * PITCH + (c + 1) * RS + t]);
            *(uint32_t*)((uint16_t*)dst + (long)(k0 + k) * klen + t * C + c0 + c) = lo | (hi << 16);
        }
        if (dst2_) {
            uint16_t* d2 = (uint16_t*)dst2_;  // [C][R][S][K]
            for (int idx = threadIdx.x; idx < KT / 2 * ROW; idx += 256) {
                const int k = (idx % (KT / 2)) * 2, j = idx / (KT / 2);  // j = c * RS + t
                const uint32_t lo = f32_to_bf16(lds[k * PITCH + j]), hi = f32_to_bf16(lds[(k + 1) * PITCH + j]);
                *(uint32_t*)(d2 + ((long)c0 * RS + j) * K + k0 + k) = lo | (hi << 16);
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < KT * ROW; idx += 256) {
            const int k = idx / ROW, j = idx - k * ROW;
            const int t = j >> 6, c = j & 63;
            Elem<T>::store(dst + (long)(k0 + k) * klen + t * C + c0 + c, lds[k * PITCH + c * RS + t]);
        }
        if (dst2_) {
            T* d2 = (T*)dst2_;  // [C][R][S][K]
            for (int idx = threadIdx.x; idx < KT * ROW; idx += 256) {
                const int k = idx % KT, j = idx / KT;  // j = c * RS + t
                Elem<T>::store(d2 + ((long)c0 * RS + j) * K + k0 + k, lds[k * PITCH + j]);
            }
        }
    }
}

template <typename T, int RS, int KT>
__device__ __forceinline__ void prepare_tile(const ManyEntry& en, int tile, float* lds) {
    constexpr int ROW = 64 * RS, PITCH = ROW + 1;
    const int C = en.g.C, K = en.g.K, nct = C / 64;
    const int kt = tile / nct, ct = tile - kt * nct;
    const int k0 = kt * KT, c0 = ct * 64;
    const float* src = en.src + ((long)k0 * C + c0) * RS;
#pragma unroll
    for (int idx = threadIdx.x; idx < KT * ROW / 4; idx += 256) {
        const int k = idx / (ROW / 4), j = (idx - k * (ROW / 4)) * 4;
        const f32x4 v = *(const f32x4*)(src + (long)k * C * RS + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[k * PITCH + j + e] = v[e];
    }
    __syncthreads();
    prepare_tile_emit<T, RS, KT>(en.dst, en.dst2, en.g.klen, C, K, k0, c0, lds);
}

template <typename T>
__global__ __launch_bounds__(256) void weight_prepare_tiled_kernel(TileArgs a) {
    __shared__ float lds[kTileK9 * (64 * 9 + 1)];
    int c = 0;
    while (c + 1 < a.n && (int)blockIdx.x >= a.tile_begin[c + 1]) ++c;
    const ManyEntry& en = a.e[c];
    const int tile = blockIdx.x - a.tile_begin[c];
    if (en.g.R * en.g.S == 9)
        prepare_tile<T, 9, kTileK9>(en, tile, lds);
    else
        prepare_tile<T, 1, 32>(en, tile, lds);
}

template <int RS, int KT>
__device__ __forceinline__ void finalize_tile(const ManyEntry& en, int tile, float* lds) {
    constexpr int ROW = 64 * RS, PITCH = ROW + 1;
    const int C = en.g.C, nct = C / 64;
    const int kt = tile / nct, ct = tile - kt * nct;
    const int k0 = kt * KT, c0 = ct * 64;
#pragma unroll
    for (int idx = threadIdx.x; idx < KT * ROW / 4; idx += 256) {
        const int k = idx / (ROW / 4), j = (idx - k * (ROW / 4)) * 4;     // j = t * 64 + c, c a multiple of 4
        const int t = j >> 6, c = j & 63;
        const f32x4 v = *(const f32x4*)(en.src + (long)(k0 + k) * en.g.klen + t * C + c0 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[k * PITCH + (c + e) * RS + t] = v[e];
    }
    __syncthreads();
    float* dst = (float*)en.dst + ((long)k0 * C + c0) * RS;
    for (int idx = threadIdx.x; idx < KT * ROW / 4; idx += 256) {
        const int k = idx / (ROW / 4), j = (idx - k * (ROW / 4)) * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = lds[k * PITCH + j + e];
        *(f32x4*)(dst + (long)k * C * RS + j) = v;
    }
}

__global__ __launch_bounds__(256) void wgrad_finalize_tiled_kernel(TileArgs a) {
    __shared__ float lds[kTileK9 * (64 * 9 + 1)];
    int c = 0;
    while (c + 1 < a.n && (int)blockIdx.x >= a.tile_begin[c + 1]) ++c;
    const ManyEntry& en = a.e[c];
    const int tile = blockIdx.x - a.tile_begin[c];
    if (en.g.R * en.g.S == 9)
        finalize_tile<9, kTileK9>(en, tile, lds);
    else
        finalize_tile<1, 32>(en, tile, lds);
}

// ---- gradient finalize + SGD step + weight refresh of every regular conv in ONE pass ------------------------------
// The three tiled passes above move the same 32 x 64 x R*S tile three times (accumulator -> OIHW gradient; master
// weights and gradient -> master weights; master weights -> kernel layouts: 90 + 135 + 90 MB for ResNet-18).  Here a
// block keeps its tile in LDS: the gradient goes out in OIHW order as before (the caller still finds every .grad), the
// master weights are read once, updated with the flat optimizer's expression (sgd_update: bit-identical to
// primia_sgd_step on the same range) and written once, and the compute-dtype copies are formed from the LDS tile.
struct SgdTileEntry {
    int C, K, klen, RS;
    const float* acc;   // [K][klen] weight-gradient accumulator (forward layout)
    float* grad;        // OIHW gradient (out)
    float* w;           // OIHW master weights (in / out)
    void* wf;           // forward-layout copy (out)
    void* wd;           // data-gradient-layout copy (out) or null
};
struct SgdTileArgs {
    SgdTileEntry e[kMaxConvs];
    int tile_begin[kMaxConvs + 1];
    int n;
    float lr, wd;
};

// KT out-channels per tile: 16 for the 3x3 layers (37 KB of LDS: four blocks per CU instead of two, 1,200 tiles instead of
// 600 — the pass is a chain of memory round trips per block and lives on blocks in flight), 32 for the 1x1 layers.
// Both loads of a tile (accumulator rows, master weights) are requested before the first barrier.
template <typename T, int RS, int KT>
__device__ __forceinline__ void sgd_tile(const SgdTileEntry& en, int tile, float* lds, float lr, float wd) {
    constexpr int ROW = 64 * RS, PITCH = ROW + 1;
    constexpr int NV = KT * ROW / 4;                 // float4 pieces of the tile
    constexpr int PER = (NV + 255) / 256;            // per thread
    const int C = en.C, nct = C / 64;
    const int kt = tile / nct, ct = tile - kt * nct;
    const int k0 = kt * KT, c0 = ct * 64;
    const long o0 = ((long)k0 * C + c0) * RS;
    f32x4 ga[PER], w[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = threadIdx.x + 256 * u;
        if (idx < NV) {
            const int k = idx / (ROW / 4), j = (idx - k * (ROW / 4)) * 4;     // accumulator side: j = t * 64 + c
            const int t = j >> 6, c = j & 63;
            ga[u] = *(const f32x4*)(en.acc + (long)(k0 + k) * en.klen + t * C + c0 + c);
            w[u] = *(const f32x4*)(en.w + o0 + (long)k * C * RS + j);        // OIHW side: j = c * RS + t
        }
    }
    // gradient tile -> LDS in OIHW order [k][c * RS + t] (finalize_tile's first half)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = threadIdx.x + 256 * u;
        if (idx < NV) {
            const int k = idx / (ROW / 4), j = (idx - k * (ROW / 4)) * 4;
            const int t = j >> 6, c = j & 63;
#pragma unroll
            for (int e = 0; e < 4; ++e) lds[k * PITCH + (c + e) * RS + t] = ga[u][e];
        }
    }
    __syncthreads();
    // OIHW side: gradient out, master weights out; the LDS tile then holds the NEW weights.  (A thread reads and
    // rewrites its own four LDS words: no barrier inside this loop.)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int idx = threadIdx.x + 256 * u;
        if (idx < NV) {
            const int k = idx / (ROW / 4), j = (idx - k * (ROW / 4)) * 4;
            const long o = o0 + (long)k * C * RS + j;
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = lds[k * PITCH + j + e];
            *(f32x4*)(en.grad + o) = g;
            f32x4 wn = w[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) wn[e] = sgd_update(wn[e], g[e], lr, wd);
            *(f32x4*)(en.w + o) = wn;
#pragma unroll
            for (int e = 0; e < 4; ++e) lds[k * PITCH + j + e] = wn[e];
        }
    }
    __syncthreads();
    prepare_tile_emit<T, RS, KT>(en.wf, en.wd, en.klen, C, en.K, k0, c0, lds);
}

template <typename T>
__global__ __launch_bounds__(256) void conv_sgd_tiled_kernel(SgdTileArgs a) {
    __shared__ float lds[kTileK9 * (64 * 9 + 1)];
    int c = 0;
    while (c + 1 < a.n && (int)blockIdx.x >= a.tile_begin[c + 1]) ++c;
    const SgdTileEntry& en = a.e[c];
    const int tile = blockIdx.x - a.tile_begin[c];
    if (en.RS == 9)
        sgd_tile<T, 9, kTileK9>(en, tile, lds, a.lr, a.wd);
    else
        sgd_tile<T, 1, 32>(en, tile, lds, a.lr, a.wd);
}

// SGD over up to 32 element ranges of a flat arena in one launch (what the fused tiles leave over: BatchNorm / fc
// parameters, the stem filter) — scalar accesses, the ranges are a few thousand elements each
constexpr int kMaxRanges = 32;
struct SgdRanges {
    long begin[kMaxRanges], len[kMaxRanges];
    int n;
    long total;
};
__global__ __launch_bounds__(256) void sgd_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, SgdRanges r,
                                                         float lr, float wd) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < r.total; i += stride) {
        long j = i;
        int k = 0;
        while (k + 1 < r.n && j >= r.len[k]) j -= r.len[k++];
        const long o = r.begin[k] + j;
        p[o] = sgd_update(p[o], g[o], lr, wd);
    }
}

// regular conv that the tiled kernels cover
static inline bool tiled_ok(const ConvGeom& g, int c_real) {
    return !g.stem && c_real == g.C && g.C % 64 == 0 && g.K % 32 == 0 && (g.R * g.S == 9 || g.R * g.S == 1);
}

}  // namespace primia

using namespace primia;

extern "C" {

int primia_abi_version(void) { return 1; }

int primia_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int c_pad,
                        int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && c_pad >= C);
    long total = (long)N * H * W;
    dim3 grid(ceil_div(total, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        nchw_to_nhwc_kernel<float><<<grid, block, 0, st>>>(src, (float*)dst, C, H * W, c_pad, total);
    else if (dtype == PRIMIA_BF16)
        nchw_to_nhwc_kernel<bf16><<<grid, block, 0, st>>>(src, (bf16*)dst, C, H * W, c_pad, total);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

int primia_nchw_to_nhwc_padded(const float* src, void* dst, int N, int C, int H, int W, int c_pad, int pad_top,
                               int pad_left, int Hp, int Wp, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && c_pad >= C);
    PRIMIA_REQUIRE(pad_top >= 0 && pad_left >= 0 && Hp >= H + pad_top && Wp >= W + pad_left);
    long total = (long)N * H * W;
    dim3 grid(ceil_div(total, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        nchw_to_nhwc_padded_kernel<float><<<grid, block, 0, st>>>(src, (float*)dst, C, H, W, c_pad, pad_top, pad_left, Hp,
                                                                  Wp, total);
    else if (dtype == PRIMIA_BF16 && c_pad == 4 && C <= 3 && W % 4 == 0 && ((uintptr_t)src & 15) == 0) {
        const long total4 = total / 4;
        nchw_to_nhwc4_padded_x4_kernel<<<ceil_div(total4, 256), 256, 0, st>>>(src, (bf16*)dst, C, H, W, pad_top, pad_left,
                                                                             Hp, Wp, total4);
    } else if (dtype == PRIMIA_BF16)
        nchw_to_nhwc_padded_kernel<bf16><<<grid, block, 0, st>>>(src, (bf16*)dst, C, H, W, c_pad, pad_top, pad_left, Hp,
                                                                 Wp, total);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

int primia_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int c_pad,
                        int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && c_pad >= C);
    long total = (long)N * H * W;
    dim3 grid(ceil_div(total, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        nhwc_to_nchw_kernel<float><<<grid, block, 0, st>>>((const float*)src, dst, C, H * W, c_pad, total);
    else if (dtype == PRIMIA_BF16)
        nhwc_to_nchw_kernel<bf16><<<grid, block, 0, st>>>((const bf16*)src, dst, C, H * W, c_pad, total);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

int primia_cast_from_f32(const float* src, void* dst, int64_t n, int dtype, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(src && dst && n >= 0);
    if (n == 0) return PRIMIA_OK;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        cast_from_f32_kernel<float><<<blocks, 256, 0, st>>>(src, (float*)dst, n);
    else if (dtype == PRIMIA_BF16)
        cast_from_f32_kernel<bf16><<<blocks, 256, 0, st>>>(src, (bf16*)dst, n);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

int primia_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(src && dst && n >= 0);
    if (n == 0) return PRIMIA_OK;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        cast_to_f32_kernel<float><<<blocks, 256, 0, st>>>((const float*)src, dst, n);
    else if (dtype == PRIMIA_BF16)
        cast_to_f32_kernel<bf16><<<blocks, 256, 0, st>>>((const bf16*)src, dst, n);
    else
        return PRIMIA_ERR_ARG;
    return launch_status();
}

int64_t primia_conv_wfwd_elems(const primia_conv_desc* d) {
    if (!d) return PRIMIA_ERR_ARG;
    ConvGeom g;
    if (!g.init(*d)) return PRIMIA_ERR_ARG;
    return (int64_t)d->K * g.klen;
}

int64_t primia_conv_wdgrad_elems(const primia_conv_desc* d) {
    if (!d) return PRIMIA_ERR_ARG;
    return (int64_t)d->C * d->R * d->S * d->K;
}

int primia_conv_weight_prepare(const primia_conv_desc* d, int c_real, const float* w_oihw,
                               void* w_fwd, void* w_dgrad, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && w_oihw && w_fwd && c_real > 0 && c_real <= d->C);
    ConvGeom g;
    PRIMIA_REQUIRE(g.init(*d));
    hipStream_t st = (hipStream_t)stream;
    long total = (long)d->K * g.klen;
    if (dtype == PRIMIA_F32)
        weight_fwd_kernel<float><<<ceil_div(total, 256), 256, 0, st>>>(w_oihw, (float*)w_fwd, g, c_real, total);
    else if (dtype == PRIMIA_BF16)
        weight_fwd_kernel<bf16><<<ceil_div(total, 256), 256, 0, st>>>(w_oihw, (bf16*)w_fwd, g, c_real, total);
    else
        return PRIMIA_ERR_ARG;
    if (w_dgrad) {
        PRIMIA_REQUIRE(!g.stem && c_real == d->C);
        long t2 = (long)d->C * d->R * d->S * d->K;
        if (dtype == PRIMIA_F32)
            weight_dgrad_kernel<float><<<ceil_div(t2, 256), 256, 0, st>>>(w_oihw, (float*)w_dgrad, d->K, d->C, d->R, d->S, t2);
        else
            weight_dgrad_kernel<bf16><<<ceil_div(t2, 256), 256, 0, st>>>(w_oihw, (bf16*)w_dgrad, d->K, d->C, d->R, d->S, t2);
    }
    return launch_status();
}

int primia_conv_wgrad_finalize(const primia_conv_desc* d, int c_real, const float* dw_acc,
                               float* dw_oihw, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && dw_acc && dw_oihw && c_real > 0 && c_real <= d->C);
    ConvGeom g;
    PRIMIA_REQUIRE(g.init(*d));
    long total = (long)d->K * c_real * d->R * d->S;
    wgrad_finalize_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(dw_acc, dw_oihw, g, c_real, total);
    return launch_status();
}

int primia_conv_weight_prepare_many(const primia_conv_desc* descs, const int* c_real, const float* const* w_oihw,
                                    void* const* w_fwd, void* const* w_dgrad, int n, int dtype,
                                    primia_stream_t stream) {
    PRIMIA_REQUIRE(descs && c_real && w_oihw && w_fwd && w_dgrad && n > 0 && n <= kMaxConvs);
    PRIMIA_REQUIRE(dtype == PRIMIA_F32 || dtype == PRIMIA_BF16);
    ManyArgs a;    // element-wise path: the stem (c_real = 3, padded layout) and anything irregular
    TileArgs ta;   // tiled path: every regular conv
    a.n = 0;
    ta.n = 0;
    long total = 0;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        ManyEntry en;
        PRIMIA_REQUIRE(en.g.init(descs[i]) && w_oihw[i] && w_fwd[i] && c_real[i] > 0 && c_real[i] <= descs[i].C);
        PRIMIA_REQUIRE(!w_dgrad[i] || (!en.g.stem && c_real[i] == descs[i].C));
        en.c_real = c_real[i];
        en.src = w_oihw[i];
        en.dst = w_fwd[i];
        en.dst2 = w_dgrad[i];
        en.begin = 0;
        if (tiled_ok(en.g, c_real[i]) && (((uintptr_t)en.src | (uintptr_t)en.dst | (uintptr_t)en.dst2) & 15) == 0) {   // (16-byte accesses)
            ta.tile_begin[ta.n] = tiles;
            ta.e[ta.n++] = en;
            tiles += tiles_of(descs[i].K, descs[i].C, descs[i].R * descs[i].S);
        } else {
            en.begin = total;
            a.e[a.n++] = en;
            total += (long)descs[i].K * en.g.klen;
            if (w_dgrad[i]) total += (long)descs[i].C * descs[i].R * descs[i].S * descs[i].K;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (ta.n) {
        ta.tile_begin[ta.n] = tiles;
        if (dtype == PRIMIA_F32)
            weight_prepare_tiled_kernel<float><<<tiles, 256, 0, st>>>(ta);
        else
            weight_prepare_tiled_kernel<bf16><<<tiles, 256, 0, st>>>(ta);
    }
    if (a.n) {
        a.total = total;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (dtype == PRIMIA_F32)
            weight_prepare_many_kernel<float><<<blocks, 256, 0, st>>>(a);
        else
            weight_prepare_many_kernel<bf16><<<blocks, 256, 0, st>>>(a);
    }
    return launch_status();
}

int primia_conv_wgrad_finalize_many(const primia_conv_desc* descs, const int* c_real, const float* const* dw_acc,
                                    float* const* dw_oihw, int n, primia_stream_t stream) {
    PRIMIA_REQUIRE(descs && c_real && dw_acc && dw_oihw && n > 0 && n <= kMaxConvs);
    ManyArgs a;
    TileArgs ta;
    a.n = 0;
    ta.n = 0;
    long total = 0;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        ManyEntry en;
        PRIMIA_REQUIRE(en.g.init(descs[i]) && dw_acc[i] && dw_oihw[i] && c_real[i] > 0 && c_real[i] <= descs[i].C);
        en.c_real = c_real[i];
        en.src = dw_acc[i];
        en.dst = dw_oihw[i];
        en.dst2 = nullptr;
        en.begin = 0;
        if (tiled_ok(en.g, c_real[i]) && (((uintptr_t)en.src | (uintptr_t)en.dst | (uintptr_t)en.dst2) & 15) == 0) {   // (16-byte accesses)
            ta.tile_begin[ta.n] = tiles;
            ta.e[ta.n++] = en;
            tiles += tiles_of(descs[i].K, descs[i].C, descs[i].R * descs[i].S);
        } else {
            en.begin = total;
            a.e[a.n++] = en;
            total += (long)descs[i].K * c_real[i] * descs[i].R * descs[i].S;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (ta.n) {
        ta.tile_begin[ta.n] = tiles;
        wgrad_finalize_tiled_kernel<<<tiles, 256, 0, st>>>(ta);
    }
    if (a.n) {
        a.total = total;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        wgrad_finalize_many_kernel<<<blocks, 256, 0, st>>>(a);
    }
    return launch_status();
}

int primia_conv_sgd_fusable(const primia_conv_desc* d, int c_real) {
    if (!d) return PRIMIA_ERR_ARG;
    ConvGeom g;
    if (!g.init(*d)) return PRIMIA_ERR_ARG;
    return tiled_ok(g, c_real) ? 1 : 0;
}

int primia_conv_sgd_step_many(const primia_conv_desc* descs, const int* c_real, const float* const* dw_acc,
                              float* const* dw_oihw, float* const* w_oihw, void* const* w_fwd, void* const* w_dgrad,
                              int n, float lr, float weight_decay, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(descs && c_real && dw_acc && dw_oihw && w_oihw && w_fwd && w_dgrad && n > 0 && n <= kMaxConvs);
    PRIMIA_REQUIRE(dtype == PRIMIA_F32 || dtype == PRIMIA_BF16);
    SgdTileArgs ta;
    ta.n = 0;
    ta.lr = lr;
    ta.wd = weight_decay;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        ConvGeom g;
        PRIMIA_REQUIRE(g.init(descs[i]) && dw_acc[i] && dw_oihw[i] && w_oihw[i] && w_fwd[i]);
        if (!tiled_ok(g, c_real[i])) return PRIMIA_ERR_UNSUPPORTED;      // ask primia_conv_sgd_fusable first
        PRIMIA_REQUIRE((((uintptr_t)dw_acc[i] | (uintptr_t)dw_oihw[i] | (uintptr_t)w_oihw[i] | (uintptr_t)w_fwd[i] |
                         (uintptr_t)w_dgrad[i]) & 15) == 0);
        SgdTileEntry& en = ta.e[ta.n];
        en.C = g.C; en.K = g.K; en.klen = g.klen; en.RS = g.R * g.S;
        en.acc = dw_acc[i]; en.grad = dw_oihw[i]; en.w = w_oihw[i]; en.wf = w_fwd[i]; en.wd = w_dgrad[i];
        ta.tile_begin[ta.n++] = tiles;
        tiles += tiles_of(g.K, g.C, en.RS);
    }
    ta.tile_begin[ta.n] = tiles;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32)
        conv_sgd_tiled_kernel<float><<<tiles, 256, 0, st>>>(ta);
    else
        conv_sgd_tiled_kernel<bf16><<<tiles, 256, 0, st>>>(ta);
    return launch_status();
}

int primia_sgd_step_ranges(float* p, const float* g, const int64_t* begin_host, const int64_t* len_host, int n, float lr,
                           float weight_decay, primia_stream_t stream) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(p && g && begin_host && len_host && n > 0 && n <= kMaxRanges);
    SgdRanges r;
    r.n = n;
    r.total = 0;
    for (int i = 0; i < n; ++i) {
        PRIMIA_REQUIRE(begin_host[i] >= 0 && len_host[i] > 0);
        r.begin[i] = begin_host[i];
        r.len[i] = len_host[i];
        r.total += len_host[i];
    }
    const int blocks = (int)((r.total + 255) / 256 < 1024 ? (r.total + 255) / 256 : 1024);
    sgd_ranges_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(p, g, r, lr, weight_decay);
    return launch_status();
}

}  // extern "C"
