// Shared device/host helpers for the primia_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/primia_hip.h"
#include "options.h"

namespace primia {

// ---- element types -------------------------------------------------------------------------
// bf16 is carried as raw uint16_t; conversion is round-to-nearest-even, NaN preserved, exactly
// what torch's float->bfloat16 cast does, so a CPU oracle that rounds with torch sees the same bits.
struct bf16 {
    uint16_t bits;
};

__host__ __device__ __forceinline__ float bf16_to_f32(uint16_t b) {
    union {
        uint32_t u;
        float f;
    } v;
    v.u = ((uint32_t)b) << 16;
    return v.f;
}

__host__ __device__ __forceinline__ uint16_t f32_to_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, quiet NaN) — the bit-twiddling
    // below costs ~6 VALU operations per value in the store-heavy kernels
    return __builtin_bit_cast(uint16_t, (__bf16)f);
#endif
    union {
        uint32_t u;
        float f;
    } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int kBytes = 4;
    static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16> {
    static constexpr int kBytes = 2;
    static constexpr int kPerChunk = 8;
    __device__ static __forceinline__ float load(const bf16* p) { return bf16_to_f32(p->bits); }
    __device__ static __forceinline__ void store(bf16* p, float v) { p->bits = f32_to_bf16(v); }
};

// 16-byte vector of raw data.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

// Unpack / pack one 16-byte chunk to/from floats. For float the chunk holds 4 values, for bf16 8.
template <typename T>
struct Chunk;
template <>
struct Chunk<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
        return v;
    }
};
template <>
struct Chunk<bf16> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(v[i] << 16);
            f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i)   // one v_cvt_pk_bf16_f32 per pair
            v[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{f[2 * i], f[2 * i + 1]}, bf16x2_));
        return v;
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// The stem's BatchNorm seen through ReLU + max-pool: where ReLU was active the pooled value IS the activation, so
// xhat = (p - beta) / gamma can be read off the pooled tensor (bn.hip PoolScatterFn, gn.hip GnPoolScatterFn,
// conv3x3_c64.hip mode 3).  p is a STORED value: its rounding error |p| * 2^-9 (bf16) is divided by gamma, so a channel
// whose gamma is small against beta (pretrained torchvision bn1 has such channels) would get xhat = rounding noise /
// gamma — those channels, and gamma == 0, take y at the argmax position instead.  ONE predicate for every site.
__host__ __device__ __forceinline__ bool pool_xhat_recoverable(float gamma, float beta) {
    return fabsf(gamma) >= 0x1p-6f * fmaxf(fabsf(beta), 1e-30f);
}

// torch.optim.SGD without momentum: d_p = g + wd*p ; p = p - lr*d_p.  ONE definition for the flat optimizer kernel
// (optim.hip) and the fused gradient-finalize + step + weight-refresh tiles (layout.hip): the two must agree bit for bit.
__device__ __forceinline__ float sgd_update(float p, float g, float lr, float wd) { return p - lr * (g + wd * p); }

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Post-launch error check -> C-ABI code.
static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PRIMIA_OK : PRIMIA_ERR_LAUNCH;
}

#define PRIMIA_REQUIRE(cond)                 \
    do {                                     \
        if (!(cond)) return PRIMIA_ERR_ARG;  \
    } while (0)

// XCD-aware, bijective remap of a 1-D block id: blocks b, b+8, b+16, ... share an XCD (observed
// placement; speed only), so give each XCD a contiguous range of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// GroupNorm + ReLU + max-pool(3, 2, 1) apply pass, statistics [N][G] given (kernel: csrc/bn.hip, caller: csrc/gn.hip)
void launch_gn_relu_pool_fwd(const void* y, void* pooled, uint8_t* argmax, const float* gamma, const float* beta,
                             const float* mean, const float* invstd, int N, int H, int W, int C, int G, int dtype,
                             hipStream_t st);

}  // namespace primia
