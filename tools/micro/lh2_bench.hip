// Stand-alone timing / phase attribution of conv3x3_lh2.hip on the three wide ResNet-18 stages (batch 256, bf16, random
// data).  The kernel source is compiled INTO this program, so experiment switches are compile-time:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLH2_DBG=n] [-DLH2_PROF] tools/micro/lh2_bench.hip -o lh2_bench
//   LH2_DBG bits: 1 no write-back, 2 no DMA after the prologue, 4 no MFMA, 8 no fragment reads
//   LH2_PROF    : per-wave cycles in load segments / matrix segments / barrier waits / write-back (s_memtime)
// Prints one line per shape and direction: us per launch, TFLOP/s; with LH2_PROF the four buckets per ping-pong half.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../primia_amd/csrc/conv3x3_lh2.hip"
#include "../../primia_amd/csrc/conv3x3_lh4.hip"   // (conv3x3_lh2_dispatch hands the 196-pixel tiles to it)
#include "../../primia_amd/csrc/options.hip"   // the option table the dispatch code reads

using namespace primia;

static uint16_t f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u; u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    struct Shape { const char* name; int H, C; } shapes[] = {{"l2.3x3", 28, 128}, {"l3.3x3", 14, 256}, {"l4.3x3", 7, 512}};
    for (auto& sh : shapes) {
        const int H = sh.H, C = sh.C, K = sh.C;
        const long M = (long)N * H * H;
        std::vector<uint16_t> hx(M * C), hw((long)K * 9 * C);
        srand(1);
        // operands like the network's: post-ReLU activations (half zeros); argv[2] = 1: dense uniform values — the matrix pipe
        // draws more power on those and the clock drops, which moved stand-alone A/B results by more than the effects under test
        const bool dense = argc > 2 && atoi(argv[2]) == 1;
        for (auto& v : hx) { const float u = (rand() / (float)RAND_MAX - 0.5f) * 2.f; v = f2bf(dense ? u : (u > 0 ? u : 0.f)); }
        for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
        bf16 *x, *w, *y;
        float* stats;
        unsigned long long* prof;
        hipMalloc(&x, M * C * 2); hipMalloc(&w, (long)K * 9 * C * 2); hipMalloc(&y, M * K * 2);
        hipMalloc(&stats, 4096 * 2 * K * 4); hipMalloc(&prof, 256 * 8 * 5 * 8);
        hipMemcpy(x, hx.data(), M * C * 2, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), (long)K * 9 * C * 2, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {     // 0 forward + statistics, 1 data gradient, 2 accumulating data gradient
            auto launch = [&]() {
                return conv3x3_lh2_dispatch(x, w, y, N, H, H, C, K, mode > 0, mode == 2, 0, mode == 0 ? stats : nullptr, nullptr);
            };
            if (launch() != 0) { printf("%s mode %d: not served\n", sh.name, mode); continue; }
            for (int i = 0; i < 3; ++i) launch();
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            const int reps = 20;
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps, tf = 2.0 * M * K * C * 9 / (us * 1e-6) / 1e12;
            printf("%s N=%d %s: %7.1f us  %6.0f TF/s  (DBG=%d)\n", sh.name, N, mode == 0 ? "fwd+stats" : mode == 1 ? "dgrad    " : "dgrad+=  ",
                   us, tf, (int)LH2_DBG);
#ifdef LH2_PROF
            hipMemset(prof, 0, 256 * 8 * 5 * 8);
            lh2_prof_buffer = prof;
            launch();
            hipDeviceSynchronize();
            lh2_prof_buffer = nullptr;
            std::vector<unsigned long long> hp(256 * 8 * 5);
            hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost);
            for (int half = 0; half < 2; ++half) {
                double b[5] = {0, 0, 0, 0, 0};
                int n = 0;
                for (int blk = 0; blk < 256; ++blk)
                    for (int wv = 4 * half; wv < 4 * half + 4; ++wv) {
                        const unsigned long long* q = &hp[((long)blk * 8 + wv) * 5];
                        if (q[0] + q[1] + q[2] + q[3] == 0) continue;
                        for (int k = 0; k < 5; ++k) b[k] += q[k];
                        ++n;
                    }
                if (n) printf("    %c waves (avg over %d): load %8.0f  matrix %8.0f  barrier-wait %8.0f  write-back+setup %8.0f  vmcnt-wait %8.0f cycles\n",
                              half ? 'B' : 'A', n, b[0] / n, b[1] / n, b[2] / n, b[3] / n, b[4] / n);
            }
            // wave (0, 0) of block 0 in detail: wm 0 carries the longest matrix segments
#endif
        }
        hipFree(x); hipFree(w); hipFree(y); hipFree(stats); hipFree(prof);
    }
    return 0;
}
