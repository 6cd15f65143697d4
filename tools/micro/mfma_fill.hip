// How much matrix throughput is left when a CU also STAGES data at a given rate?  The stage body of mfma_stage.hip
// (8 waves per CU, 36 v_mfma_f32_16x16x32_bf16 per wave and iteration, fragments re-read from LDS, one barrier) with K
// LDS-DMA pieces of 1 KiB per wave and iteration streamed from a working set of `set` bytes per CU (32 KiB per CU: L2 hits;
// 4 MiB per CU = 1 GiB in all: HBM), through a ring that keeps 2 iterations of pieces in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_fill.hip -o tools/micro/mfma_fill
// At full matrix speed an iteration is 1,152 cycles per SIMD (2 waves x 36 x 16), so K pieces per wave = 7.1 K B/clk/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

template <int K, bool PINGPONG>
__global__ __launch_bounds__(512) void k(const char* __restrict__ big, long set_bytes, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // fragments: 26 KiB of LDS filled once with this CU's first bytes (random data)
    const char* mine = big + (long)blockIdx.x * set_bytes;
    for (int i = tid; i < 26 * 1024 / 16; i += 512) ((bf16x8*)smem)[i] = ((const bf16x8*)mine)[i];
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
    // staging ring behind the fragment area: 3 slots of (8 waves x K KiB); piece p of iteration it comes from a streaming
    // position that wraps inside this CU's working set
    const long per_it = 8L * K * 1024;
    long pos = 0;
    auto stage = [&](int slot) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            const char* g = mine + ((pos + (long)(wave * K + p) * 1024) % set_bytes) + lane * 16;
            dma16(g, __builtin_amdgcn_readfirstlane(lds0 + 32 * 1024 + slot * (unsigned)per_it + (wave * K + p) * 1024));
        }
        pos += per_it;
    };
    auto wait_one_behind = [&]() {      // everything but the newest iteration's K pieces has landed
        if (K == 0) return;
        if (K == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (K == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (K == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (K == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (K == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    };
    if (K) { stage(0); stage(1); }
    int slot = 2;
    for (int it = 0; it < iters; ++it) {
        wait_one_behind();
        __builtin_amdgcn_s_barrier();
        if (PINGPONG && wave >= 4) {            // second half: multiply first, then issue (the kernels' ping-pong)
            load();
            __builtin_amdgcn_sched_barrier(0);
            mfma();
            __builtin_amdgcn_sched_barrier(0);
            if (K) stage(slot);
        } else {
            if (K) stage(slot);
            load();
            __builtin_amdgcn_sched_barrier(0);
            mfma();
            __builtin_amdgcn_sched_barrier(0);
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) s += acc[t][i][0] + acc[t][i][1] + acc[t][i][2] + acc[t][i][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int K, bool PP>
void run(const char* big, long set_bytes, float* out, const char* what) {
    const int iters = 3000, grid = 256;
    const int lds = 32 * 1024 + 3 * 8 * (K ? K : 1) * 1024;
    hipFuncSetAttribute((const void*)k<K, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<K, PP><<<grid, 512, lds>>>(big, set_bytes, out, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<K, PP><<<grid, 512, lds>>>(big, set_bytes, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 36 * 16.0 * 16 * 32 * 2;
    const double bytes = (double)grid * iters * 8 * K * 1024;
    printf("%-10s K=%d%s: %6.0f TFLOP/s   staged %5.2f TB/s = %5.1f B/clk/CU @2.4GHz\n", what, K, PP ? " ping-pong" : "          ",
           flops / ms / 1e9, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

// Same work, but a NINTH wave issues every DMA piece (8 K per iteration) and the eight working waves never touch vmcnt.
template <int K>
__global__ __launch_bounds__(576) void kl(const char* __restrict__ big, long set_bytes, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* mine = big + (long)blockIdx.x * set_bytes;
    for (int i = tid; i < 26 * 1024 / 16; i += 576) ((bf16x8*)smem)[i] = ((const bf16x8*)mine)[i];
    __syncthreads();
    const long per_it = 8L * K * 1024;
    if (wave == 8) {
        long pos = 0;
        auto stage = [&](int slot) {
#pragma unroll
            for (int p = 0; p < 8 * K; ++p) {
                const char* g = mine + ((pos + (long)p * 1024) % set_bytes) + lane * 16;
                dma16(g, __builtin_amdgcn_readfirstlane(lds0 + 32 * 1024 + slot * (unsigned)per_it + p * 1024));
            }
            pos += per_it;
        };
        stage(0); stage(1);
        int slot = 2;
        for (int it = 0; it < iters; ++it) {
            // everything but the newest iteration's 8 K pieces has landed
            if (8 * K == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (8 * K == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (8 * K == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage(slot);
            slot = slot == 2 ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
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
    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        load();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q][i], b[q][t], acc[t][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) s += acc[t][i][0] + acc[t][i][1] + acc[t][i][2] + acc[t][i][3];
    out[blockIdx.x * 512 + (tid & 511)] = s;
}

template <int K>
void runl(const char* big, long set_bytes, float* out, const char* what) {
    const int iters = 3000, grid = 256;
    const int lds = 32 * 1024 + 3 * 8 * K * 1024;
    hipFuncSetAttribute((const void*)kl<K>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    kl<K><<<grid, 576, lds>>>(big, set_bytes, out, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kl<K><<<grid, 576, lds>>>(big, set_bytes, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 36 * 16.0 * 16 * 32 * 2;
    const double bytes = (double)grid * iters * 8 * K * 1024;
    printf("%-10s K=%d loader wave: %6.0f TFLOP/s   staged %5.2f TB/s = %5.1f B/clk/CU @2.4GHz\n", what, K,
           flops / ms / 1e9, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    const long per_cu_big = 4L << 20, total = 256 * per_cu_big;
    std::vector<unsigned short> h(total / 2);
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) { float f = ((rand() & 0xffff) / 65536.f - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    char* big; float* out;
    hipMalloc(&big, total); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(big, h.data(), total, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        const long set = pass == 0 ? per_cu_big : 32 * 1024;
        const char* what = pass == 0 ? "HBM set" : "L2 set";
        run<0, false>(big, set, out, what);
        run<1, false>(big, set, out, what);
        run<2, false>(big, set, out, what);
        run<3, false>(big, set, out, what);
        run<4, false>(big, set, out, what);
        run<1, true>(big, set, out, what);
        run<2, true>(big, set, out, what);
        run<3, true>(big, set, out, what);
        run<4, true>(big, set, out, what);
        runl<1>(big, set, out, what);
        runl<2>(big, set, out, what);
        runl<3>(big, set, out, what);
        runl<4>(big, set, out, what);
    }
    return 0;
}
