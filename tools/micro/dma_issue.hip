// What a vector-memory instruction costs a wave to ISSUE on gfx950, by kind — the number behind the phase timings of
// profiles/r05_c64_phases.txt ("an LDS-DMA instruction costs ~250-300 cycles of issue whatever it moves").
// Each wave issues K instructions back to back (addresses ready in registers, source resident in L2), s_memtime around the
// ISSUE only, then drains; R rounds; W waves per CU (one block per CU).  Cycles per instruction, mean over waves:
//   0  buffer_load_dwordx4 ... lds, M0 written before each            (the kernels' form)
//   1  buffer_load_dwordx4 ... lds, M0 written once
//   2  global_load_lds_dwordx4, M0 written before each
//   3  buffer_load_dword ... lds, M0 written before each              (a quarter of the bytes)
//   4  buffer_load_dwordx4 into registers                             (no LDS)
//   5  buffer_store_dwordx4
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/dma_issue.hip -o dma_issue && ./dma_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V, int K>
__global__ __launch_bounds__(512) void issue_kernel(const char* src, char* dst, unsigned long long* out, int rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (K * 1024);
    i32x4 rs, rd;
    {
        const unsigned long long a = (unsigned long long)src, b = (unsigned long long)dst;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
        rs[2] = __builtin_amdgcn_readfirstlane((int)(1u << 26));
        rs[3] = __builtin_amdgcn_readfirstlane(0x00020000);
        rd = rs;
        rd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        rd[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
    }
    unsigned voff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) voff[k] = (unsigned)(((blockIdx.x * 8 + wave) * K + k) * 1024 + lane * 16) & ((1u << 22) - 1);   // 4 MB window
    u32x4 sink = {0u, 0u, 0u, 0u};
    unsigned long long issue = 0, total = 0;
    asm volatile("s_nop 4" ::: "memory");
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned m0v = lds0 + k * 1024;
            if constexpr (V == 0)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff[k]), "s"(rs), "s"(m0v) : "memory");
            else if constexpr (V == 1) {
                if (k == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");
                asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff[k]), "s"(rs) : "memory");
            } else if constexpr (V == 2) {
                const char* g = src + voff[k];
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(m0v) : "memory");
            } else if constexpr (V == 3)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds" ::"v"(voff[k]), "s"(rs), "s"(m0v) : "memory");
            else if constexpr (V == 4) {
                u32x4 d;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff[k]), "s"(rs) : "memory");
                asm volatile("" ::"v"(d));
            } else
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(sink), "v"(voff[k]), "s"(rd) : "memory");
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        issue += t1 - t0;
        total += t2 - t0;
    }
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 2] = issue;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = total;
    }
}

template <int V>
static void run(const char* name, const char* src, char* dst, unsigned long long* out) {
    constexpr int K = 4;
    const int rounds = 200;
    std::vector<unsigned long long> h(256 * 8 * 2);
    printf("%-52s", name);
    for (int W : {1, 2, 4, 8}) {
        hipMemset(out, 0, h.size() * 8);
        hipLaunchKernelGGL((issue_kernel<V, K>), dim3(256), dim3(64 * W), 8 * K * 1024, 0, src, dst, out, rounds);
        hipLaunchKernelGGL((issue_kernel<V, K>), dim3(256), dim3(64 * W), 8 * K * 1024, 0, src, dst, out, rounds);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double si = 0, st = 0;
        int n = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < W; ++w) {
                si += (double)h[(b * 8 + w) * 2];
                st += (double)h[(b * 8 + w) * 2 + 1];
                ++n;
            }
        printf("  W=%d: %6.0f issue %6.0f incl. drain", W, si / n / rounds / K, st / n / rounds / K);
    }
    printf("   (cycles per instruction)\n");
}

int main() {
    char *src, *dst;
    unsigned long long* out;
    hipMalloc(&src, 1 << 26);
    hipMalloc(&dst, 1 << 26);
    hipMalloc(&out, 256 * 8 * 2 * 8);
    hipMemset(src, 1, 1 << 26);
    hipFuncSetAttribute((const void*)issue_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    run<0>("buffer_load_dwordx4 lds, M0 per instruction", src, dst, out);
    run<1>("buffer_load_dwordx4 lds, M0 once", src, dst, out);
    run<2>("global_load_lds_dwordx4, M0 per instruction", src, dst, out);
    run<3>("buffer_load_dword lds, M0 per instruction", src, dst, out);
    run<4>("buffer_load_dwordx4 into registers", src, dst, out);
    run<5>("buffer_store_dwordx4", src, dst, out);
    return 0;
}
