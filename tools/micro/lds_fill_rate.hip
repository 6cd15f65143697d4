// How fast can ONE CU fill its LDS from memory?  The patch weight gradient and the linear-halo convolution stage their
// operands with LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per instruction); the kernels' staging-only
// runs move 14-17 B/clk/CU (profiles/r02_wgrad32_experiments.txt).  This program separates the candidates:
//   path  0: global_load_lds_dwordx4 (LDS-DMA), batches      1: global_load_dwordx4 -> VGPR -> ds_write_b128, batches
//         2: LDS-DMA with a rolling window (counted vmcnt): `inflight` pieces per wave always outstanding
//   src   working set per block (KiB): small = L2-resident after the first pass, large = streams from HBM / MALL
//   shape 0: every lane row is one contiguous 1-KiB piece
//         1: 8 segments of 128 B per instruction, segment stride = `stride` bytes (the kernels' "8 pixel rows of 128 B")
//   waves per block (1 block per CU), pieces in flight per wave
//   hipcc --offload-arch=gfx950 -O3 lds_fill_rate.hip -o lds_fill_rate && ./lds_fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

template <int PATH, int INFLIGHT>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, long block_bytes, int shape, long stride,
                                                   int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* base = src + (long)blockIdx.x * block_bytes;
    // per-lane offset inside one piece
    const long lane_off = shape == 0 ? lane * 16 : (long)(lane >> 3) * stride + (lane & 7) * 16;
    const long piece_span = shape == 0 ? 1024 : 8 * stride;
    const long npieces = block_bytes / piece_span;          // pieces available in the working set
    unsigned acc = 0;
    long pc = wave;                                           // this wave's piece cursor
    for (int it = 0; it < iters; ++it) {
        if (PATH == 2) {
            // rolling window: INFLIGHT pieces stay in flight per wave (counted vmcnt), as a deep staging ring would
            const char* g = base + (pc % npieces) * piece_span + lane_off;
            dma16(g, __builtin_amdgcn_readfirstlane(lds0 + ((wave * 8 + (it & 7)) & 63) * 1024));
            pc += nwaves;
            if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (INFLIGHT == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        } else if (PATH == 0) {
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) {
                const char* g = base + (pc % npieces) * piece_span + lane_off;
                dma16(g, __builtin_amdgcn_readfirstlane(lds0 + ((wave * INFLIGHT + k) & 63) * 1024));
                pc += nwaves;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[INFLIGHT];
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) {
                const char* g = base + (pc % npieces) * piece_span + lane_off;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(g) : "memory");
                pc += nwaves;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) {
                const unsigned a = lds0 + ((wave * INFLIGHT + k) & 63) * 1024 + lane * 16;
                asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v[k]) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    acc = ((unsigned*)smem)[tid];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PATH, int INFLIGHT>
static double run(const char* src, long block_bytes, int shape, long stride, int waves, int iters, unsigned* sink,
                  int nblk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto kern = fill_kernel<PATH, INFLIGHT>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    kern<<<nblk, waves * 64, 64 * 1024>>>(src, block_bytes, shape, stride, iters / 4 + 1, sink);
    hipEventRecord(e0);
    kern<<<nblk, waves * 64, 64 * 1024>>>(src, block_bytes, shape, stride, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nblk * waves * (PATH == 2 ? 1 : INFLIGHT) * 1024.0 * iters;
    return bytes / (ms * 1e-3);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("%s: %d CUs, %.0f MHz\n", prop.name, ncu, clk / 1e6);
    const long maxblock = 4L << 20;
    char* src;
    unsigned* sink;
    hipMalloc(&src, (size_t)ncu * maxblock);
    hipMemset(src, 1, (size_t)ncu * maxblock);
    hipMalloc(&sink, 4);
    printf("%-7s %-6s %-9s %-5s %-8s %10s %12s\n", "path", "shape", "set/CU", "waves", "inflight", "TB/s", "B/clk/CU");
    const long sets[3] = {32 << 10, 512 << 10, 4 << 20};
    for (int path = 0; path < 3; ++path)
        for (int shape = 0; shape < 2; ++shape)
            for (int si = 0; si < 3; ++si)
                for (int waves = 4; waves <= 8; waves += 4)
                    for (int infl = 2; infl <= (path == 2 ? 16 : 8); infl *= 2) {
                        const long bb = sets[si];
                        const long stride = 256;          // neighbouring 128-B rows of a 128-channel bf16 tensor
                        const int iters = path == 2 ? 16000 : 2000;
                        double r;
#define RUN(P, I) r = run<P, I>(src, bb, shape, stride, waves, iters, sink, ncu)
                        if (path == 0) {
                            if (infl == 2) RUN(0, 2); else if (infl == 4) RUN(0, 4); else RUN(0, 8);
                        } else if (path == 1) {
                            if (infl == 2) RUN(1, 2); else if (infl == 4) RUN(1, 4); else RUN(1, 8);
                        } else {
                            if (infl == 2) RUN(2, 2); else if (infl == 4) RUN(2, 4); else if (infl == 8) RUN(2, 8); else RUN(2, 16);
                        }
                        printf("%-7s %-6s %6ld KiB %-5d %-8d %10.2f %12.1f\n", path == 0 ? "lds-dma" : path == 1 ? "vgpr" : "dma-roll",
                               shape == 0 ? "1KiB" : "8x128", bb >> 10, waves, infl, r / 1e12, r / ncu / clk);
                    }
    return 0;
}
