// Microbenchmark of a 32x32x16 patch-wgrad stage body (no global traffic), random bf16 data.
//   MODE 0: 4 waves / block, ONE wave per SIMD, wave tile 64 k x 32 c x 9 taps = 18 accumulators f32x16 (288 regs);
//           per 16-pixel k-step: 2 dy + 9 x fragments via ds_read_b64_tr_b16 (22 reads) for 18 MFMAs.
//   MODE 1: 8 waves / block, two per SIMD, wave tile 32 k x 32 c x 9 taps (144 regs): 20 reads per 9 MFMAs.
// LDS image per 32-channel group: [pixel slot][32 ch] = 64 B rows: 4 consecutive slots = 256 B = all 64 banks.
//   hipcc --offload-arch=gfx950 -O3 mfma_stage32.hip -o mfma_stage32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) bf16x4* lds4_t;

constexpr int HS = 12, XSLOTS = 72;            // 6 x 12 halo slots (4 x 8 sub-patch + ring), as the real kernel
constexpr int XG = XSLOTS * 64;                // bytes of one 32-channel group of one sub-patch halo
constexpr int DYG = 32 * 64;                   // 32 pixels x 32 channels

__device__ __forceinline__ bf16x8 frag(const char* base, int off0, int off1) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(base + off0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(base + off1));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int KH>   // out-channel groups per wave: 2 (MODE 0) or 1 (MODE 1)
__global__ __launch_bounds__(KH == 2 ? 256 : 512) void k(const bf16x8* __restrict__ src, float* out, int iters,
                                                         int barrier) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = KH == 2 ? 256 : 512;
    // LDS: x [4 groups][2 sub-patches][72 slots][64 B] = 36 KB, dy [2 groups][64 px][64 B] = 8 KB
    for (int i = tid; i < 44 * 1024 / 16; i += nthreads) ((bf16x8*)smem)[i] = src[i];
    __syncthreads();
    const int cg = KH == 2 ? wave : (wave & 3);          // this wave's in-channel group
    const int kg0 = KH == 2 ? 0 : (wave >> 2);           // first out-channel group
    const char* xb = smem + cg * 2 * XG;
    const char* db = smem + 4 * 2 * XG;
    // lane geometry of the transposing read: 16-lane group g = lane >> 4: channels 16*(g&1).., pixels 8*(g>>1) + tp (+4)
    const int fr = lane & 15, g = lane >> 4;
    const int tp = fr >> 2, cb = (16 * (g & 1) + 4 * (fr & 3)) * 2;
    int offx[9][2], offa[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kk = 8 * (g >> 1) + tp + 4 * h;        // pixel of the 16-pixel k-step: row kk >> 3, col kk & 7
        offa[h] = kk * 64 + cb;
#pragma unroll
        for (int t = 0; t < 9; ++t) offx[t][h] = (((kk >> 3) + t / 3) * HS + (kk & 7) + t % 3) * 64 + cb;
    }
    f32x16 acc[9][KH];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < KH; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][i][j] = 0.f;
    bf16x8 a[2][KH], b[2][9];
    auto load = [&](int buf, int step) {       // step 0..3: 16-pixel k-step of the 64-pixel stage
        const int q = step >> 1, half = step & 1;
        const char* lx = xb + q * XG + half * 2 * HS * 64;
        const char* la = db + (q * 32 + half * 16) * 64;
#pragma unroll
        for (int i = 0; i < KH; ++i) a[buf][i] = frag(la + (kg0 + i) * 64 * 64, offa[0], offa[1]);
#pragma unroll
        for (int t = 0; t < 9; ++t) b[buf][t] = frag(lx, offx[t][0], offx[t][1]);
    };
    auto mfma = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < KH; ++i)
                acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[buf][i], b[buf][t], acc[t][i], 0, 0, 0);
    };
    if (barrier < 2) {
        load(0, 0);
        for (int it = 0; it < iters; ++it) {
            if (barrier) __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                load((step + 1) & 1, (step + 1) & 3);       // next k-step's fragments fly while this one multiplies
                mfma(step & 1);
            }
        }
    } else if (barrier == 2) {
        // no fragment read may cross the stage barrier (the real kernel's constraint); stage = 2 k-steps, all reads up front
        for (int it = 0; it < 2 * iters; ++it) {
            __builtin_amdgcn_s_barrier();
            load(0, (2 * it) & 3);
            load(1, (2 * it + 1) & 3);
            mfma(0);
            mfma(1);
        }
    } else {
        // stage = 2 k-steps; second k-step's reads issued after the first k-step's MFMAs started
        for (int it = 0; it < 2 * iters; ++it) {
            __builtin_amdgcn_s_barrier();
            load(0, (2 * it) & 3);
            __builtin_amdgcn_sched_barrier(0);
            load(1, (2 * it + 1) & 3);
            mfma(0);
            __builtin_amdgcn_sched_barrier(0);
            mfma(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < KH; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[t][i][j];
    out[blockIdx.x * nthreads + tid] = s;
}

template <int KH>
void run(const bf16x8* src, float* out, int barrier, int blocks_per_cu, const char* what) {
    const int iters = 2000, grid = 256 * blocks_per_cu, nthreads = KH == 2 ? 256 : 512;
    const int lds = 44 * 1024;
    hipFuncSetAttribute((const void*)k<KH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<KH><<<grid, nthreads, lds>>>(src, out, 50, barrier);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<KH><<<grid, nthreads, lds>>>(src, out, iters, barrier);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * (nthreads / 64) * iters * 4 * 9 * KH * 32.0 * 32 * 16 * 2;
    printf("%-60s %.0f TFLOP/s\n", what, flops / ms / 1e9);
}

int main() {
    std::vector<unsigned short> h(44 * 1024 / 2);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    bf16x8* src; float* out;
    hipMalloc(&src, 44 * 1024); hipMalloc(&out, 256 * 2 * 512 * 4);
    hipMemcpy(src, h.data(), 44 * 1024, hipMemcpyHostToDevice);
    run<2>(src, out, 0, 1, "4 waves, 1/SIMD, 64k x 32c x 9 per wave, tr reads");
    run<2>(src, out, 1, 1, "  + barrier per 64-pixel stage");
    run<1>(src, out, 0, 1, "8 waves, 2/SIMD, 32k x 32c x 9 per wave, tr reads");
    run<1>(src, out, 1, 1, "  + barrier per 64-pixel stage");
    run<1>(src, out, 2, 1, "  barrier per 32-pixel stage, no read crosses it");
    run<1>(src, out, 3, 1, "  same, reads pinned before the MFMAs");
    return 0;
}
