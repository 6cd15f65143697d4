// What does it cost to replace BatchNorm's [blocks][2][C] partial table + one-block finalize launch by int64 fixed-point
// atomics into [2][C][2] accumulators that the consumer turns into mean / invstd in its prologue?
//   chain A: reduce (plain partial stores) -> finalize (fp64 combine, 1 launch) -> apply (reads mean / invstd arrays)
//   chain B: memset(acc) -> reduce (2 x 2C int64 atomics per block) -> apply (prologue derives the constants from acc)
//   chain C: chain B without the memset (the engine zeroes every layer's accumulators once per step)
// for the four BatchNorm shapes of ResNet-18 at batch 256.  Same-address contention is the question: up to 1024 blocks
// finish together and add to the same 4C addresses.
//   hipcc --offload-arch=gfx950 -O3 atomic_acc.hip -o atomic_acc
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__device__ __forceinline__ void acc_add(long long* acc2, float v) {
    const double d = (double)v * 1048576.0;       // units of 2^-20
    const double dh = floor(d);
    const long long hi = (long long)dh;
    const long long lo = (long long)((d - dh) * 4503599627370496.0);   // 2^52
    __hip_atomic_fetch_add(acc2, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(acc2 + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double acc_get(const long long* acc2) {
    return ((double)acc2[0] + (double)acc2[1] * (1.0 / 4503599627370496.0)) * (1.0 / 1048576.0);
}

template <int MODE>   // 0: partial stores, 1: atomics
__global__ __launch_bounds__(256) void reduce_kernel(const unsigned short* __restrict__ y, long M, int C, long rpb,
                                                     float* __restrict__ partials, long long* __restrict__ acc) {
    const int tpr = C / 8, rpp = 256 / tpr;
    const int rg = threadIdx.x / tpr, cc = threadIdx.x % tpr;
    const long r0 = (long)blockIdx.x * rpb;
    long r1 = r0 + rpb;
    if (r1 > M) r1 = M;
    float s1[8], s2[8];
    for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
    if (rg < rpp) {
        for (long r = r0 + rg; r < r1; r += rpp) {
            const u32x4 v = *(const u32x4*)(y + r * C + cc * 8);
            for (int i = 0; i < 4; ++i) {
                const float a = bf2f(v[i] & 0xffff), b = bf2f(v[i] >> 16);
                s1[2 * i] += a;
                s2[2 * i] += a * a;
                s1[2 * i + 1] += b;
                s2[2 * i + 1] += b * b;
            }
        }
    }
    __shared__ float red[2][256 * 8];
    if (rg < rpp)
        for (int i = 0; i < 8; ++i) {
            red[0][rg * C + cc * 8 + i] = s1[i];
            red[1][rg * C + cc * 8 + i] = s2[i];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int g = 0; g < rpp; ++g) {
            a += red[0][g * C + c];
            b += red[1][g * C + c];
        }
        if (MODE == 0) {
            partials[((long)blockIdx.x * 2 + 0) * C + c] = a;
            partials[((long)blockIdx.x * 2 + 1) * C + c] = b;
        } else {
            acc_add(acc + 2 * c, a);
            acc_add(acc + 2 * (C + c), b);
        }
    }
}

__global__ __launch_bounds__(1024) void finalize_kernel(const float* __restrict__ partials, int nblk, int C, long M,
                                                        float* mean, float* invstd) {
    __shared__ double sa[64][17], sb[64][17];
    const int cl = threadIdx.x & 15, ks = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double a = 0, b = 0;
    for (int k = ks; k < nblk; k += 64) {
        a += partials[((long)k * 2) * C + c];
        b += partials[((long)k * 2 + 1) * C + c];
    }
    sa[ks][cl] = a;
    sb[ks][cl] = b;
    __syncthreads();
    if (ks) return;
    for (int k = 1; k < 64; ++k) {
        a += sa[k][cl];
        b += sb[k][cl];
    }
    const double m = a / M;
    double var = b / M - m * m;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + 1e-5));
}

template <int MODE>
__global__ __launch_bounds__(256) void apply_kernel(const unsigned short* __restrict__ y, unsigned short* __restrict__ z,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const long long* __restrict__ acc, long M, long nchunks, int C,
                                                    float* save_mean, float* save_invstd) {
    __shared__ float sm[2][512];
    for (int c = threadIdx.x; c < C; c += 256) {
        if (MODE == 0) {
            sm[0][c] = mean[c];
            sm[1][c] = invstd[c];
        } else {
            const double a = acc_get(acc + 2 * c), b = acc_get(acc + 2 * (C + c));
            const double m = a / M;
            double var = b / M - m * m;
            const float fm = (float)m, fi = (float)(1.0 / sqrt(var + 1e-5));
            sm[0][c] = fm;
            sm[1][c] = fi;
            if (blockIdx.x == 0) {
                save_mean[c] = fm;
                save_invstd[c] = fi;
            }
        }
    }
    __syncthreads();
    const int cpr = C / 8;
    const long stride = (long)gridDim.x * 256;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nchunks; q += stride) {
        const int c0 = (int)(q % cpr) * 8;
        u32x4 v = *(const u32x4*)(y + q * 8);
        u32x4 o;
        for (int i = 0; i < 4; ++i) {
            float a = bf2f(v[i] & 0xffff), b = bf2f(v[i] >> 16);
            a = fmaxf((a - sm[0][c0 + 2 * i]) * sm[1][c0 + 2 * i], 0.f);
            b = fmaxf((b - sm[0][c0 + 2 * i + 1]) * sm[1][c0 + 2 * i + 1], 0.f);
            o[i] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
        }
        __builtin_nontemporal_store(o, (u32x4*)(z + q * 8));
    }
}

// atomics alone: `nblk` blocks x 4C atomics, nothing else (the pure contention cost)
__global__ __launch_bounds__(256) void atomics_only_kernel(long long* acc, int C) {
    for (int c = threadIdx.x; c < C; c += 256) {
        acc_add(acc + 2 * c, 1.25f + threadIdx.x);
        acc_add(acc + 2 * (C + c), 3.5f + blockIdx.x);
    }
}

int main() {
    const long Ms[4] = {802816, 200704, 50176, 12544};
    const int Cs[4] = {64, 128, 256, 512};
    unsigned short *y, *z;
    float *partials, *mean, *invstd;
    long long* acc;
    hipMalloc(&y, 802816L * 64 * 2);
    hipMalloc(&z, 802816L * 64 * 2);
    hipMalloc(&partials, 1024L * 2 * 512 * 4);
    hipMalloc(&mean, 512 * 4);
    hipMalloc(&invstd, 512 * 4);
    hipMalloc(&acc, 4 * 512 * 8);
    std::vector<unsigned short> h(802816L * 64);
    srand(1);
    for (auto& v : h) v = (unsigned short)(0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15));
    hipMemcpy(y, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200;
    for (int s = 0; s < 4; ++s) {
        const long M = Ms[s];
        const int C = Cs[s];
        long nb = (M + 31) / 32;
        if (nb > 1024) nb = 1024;
        const long rpb = (M + nb - 1) / nb;
        const int nblk = (int)((M + rpb - 1) / rpb);
        const long nchunks = M * C / 8;
        long ab = (nchunks + 255) / 256;
        if (ab > 2048) ab = 2048;
        float ms[4];
        for (int chain = 0; chain < 4; ++chain) {
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(acc, 0, 4 * 512 * 8);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int it = 0; it < iters; ++it) {
                    if (chain == 0) {
                        reduce_kernel<0><<<nblk, 256>>>(y, M, C, rpb, partials, acc);
                        finalize_kernel<<<C / 16, 1024>>>(partials, nblk, C, M, mean, invstd);
                        apply_kernel<0><<<(int)ab, 256>>>(y, z, mean, invstd, acc, M, nchunks, C, mean, invstd);
                    } else if (chain == 1 || chain == 2) {
                        if (chain == 1) hipMemsetAsync(acc, 0, 4 * C * 8);
                        reduce_kernel<1><<<nblk, 256>>>(y, M, C, rpb, partials, acc);
                        apply_kernel<1><<<(int)ab, 256>>>(y, z, mean, invstd, acc, M, nchunks, C, mean, invstd);
                    } else {
                        atomics_only_kernel<<<nblk, 256>>>(acc, C);
                    }
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[chain], e0, e1);
            }
        }
        printf("M=%7ld C=%3d blocks=%4d | A reduce+finalize+apply %.2f us | B memset+reduce(atomics)+apply %.2f us | "
               "C without memset %.2f us | atomics-only launch %.2f us\n",
               M, C, nblk, ms[0] * 1000 / iters, ms[1] * 1000 / iters, ms[2] * 1000 / iters, ms[3] * 1000 / iters);
    }
    // conv-epilogue pattern: 256 persistent blocks, each adds 2 x 128 values per tile, T tiles spread over the launch
    return 0;
}
