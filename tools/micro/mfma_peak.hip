// Microbenchmark: sustained v_mfma_f32_16x16x32_bf16 rate with W waves per SIMD and C independent
// accumulator chains per wave, no memory traffic.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int CH>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
void run(int threads, int blocks_per_cu) {
    int iters = 20000;
    float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * blocks_per_cu;
    k<CH><<<grid, threads>>>(out, 100);
    hipEventRecord(e0);
    k<CH><<<grid, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * (threads / 64) * iters * CH * 16.0 * 16 * 32 * 2;
    printf("chains=%d waves/CU=%d: %.1f TFLOP/s (%.3f ms)\n", CH, blocks_per_cu * threads / 64, flops / ms / 1e9, ms);
    hipFree(out);
}
int main() {
    run<1>(256, 1); run<2>(256, 1); run<4>(256, 1); run<8>(256, 1);
    run<2>(512, 1); run<4>(512, 1); run<8>(512, 1);
    run<4>(512, 2); run<8>(256, 4);
    return 0;
}
