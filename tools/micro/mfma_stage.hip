// Microbenchmark of the patch-wgrad stage body in isolation (no global traffic): 8 waves per CU, per iteration
// 36 x v_mfma_f32_16x16x32_bf16 on 18 accumulators with DISTINCT operand registers (2x2 A, 2x9 B fragments),
// optionally re-read from LDS every iteration (44 ds_read_b64_tr_b16 pairs... here plain b128 reads of the same
// byte count) and optionally one s_barrier per iteration.  Random bf16 data (DVFS: zeros clock higher).
//   hipcc --offload-arch=gfx950 -O3 mfma_stage.hip -o mfma_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>   // bit0: barrier per iteration, bit1: LDS re-read of the fragments, bit2: ping-pong halves
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fill 26 KB of LDS with random data
    for (int i = tid; i < 26 * 1024 / 16; i += 512) ((bf16x8*)smem)[i] = src[i];
    __syncthreads();
    bf16x8 a[2][2], b[2][9];
    auto load = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[q][i] = *(const bf16x8*)(smem + ((q * 2 + i) * 1024 + lane * 16));
#pragma unroll
            for (int t = 0; t < 9; ++t) b[q][t] = *(const bf16x8*)(smem + (4096 + (q * 9 + t) * 1024 + lane * 16));
        }
    };
    load();
    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = f32x4{0, 0, 0, 0};
    auto mfma = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q][i], b[q][t], acc[t][i], 0, 0, 0);
    };
    if constexpr (MODE & 4) {
        if (wave < 4) {
            for (int it = 0; it < iters; ++it) {
                __builtin_amdgcn_s_barrier();
                load();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                mfma();
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
                __builtin_amdgcn_s_barrier();
                mfma();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                load();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (MODE & 1) __builtin_amdgcn_s_barrier();
            if (MODE & 2) {
                load();
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) s += acc[t][i][0] + acc[t][i][1] + acc[t][i][2] + acc[t][i][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const bf16x8* src, float* out, const char* what) {
    const int iters = 4000, grid = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    k<MODE><<<grid, 512, 80 * 1024>>>(src, out, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<grid, 512, 80 * 1024>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 36 * 16.0 * 16 * 32 * 2;
    printf("%-44s %.0f TFLOP/s  (%.2f us per 49 iterations)\n", what, flops / ms / 1e9, ms * 1e3 / iters * 49);
}

int main() {
    std::vector<unsigned short> h(26 * 1024 / 2);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    bf16x8* src; float* out;
    hipMalloc(&src, 26 * 1024); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), 26 * 1024, hipMemcpyHostToDevice);
    run<0>(src, out, "mfma only");
    run<1>(src, out, "mfma + barrier");
    run<2>(src, out, "lds re-read + mfma");
    run<3>(src, out, "lds re-read + mfma + barrier");
    run<4>(src, out, "ping-pong halves (2 barriers)");
    return 0;
}
