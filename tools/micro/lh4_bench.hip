// A/B of conv3x3_lh4 (csrc/conv3x3_lh4.hip: 8 matrix + 4 loader waves, 196-pixel tiles, one barrier per step) against
// conv3x3_lh2 (as shipped, and forced to 196-pixel tiles): us per launch on the three wide stages, outputs and BatchNorm
// partials of lh4 compared bit for bit with lh2's 196-pixel form.  argv[1] = batch (256), argv[2] = 1: dense uniform operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/lh4_bench.hip -o tools/micro/lh4_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../primia_amd/csrc/conv3x3_lh2.hip"
#include "../../primia_amd/csrc/options.hip"
#include "../../primia_amd/csrc/conv3x3_lh4.hip"

using namespace primia;

static uint16_t f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u; u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    const bool dense = argc > 2 && atoi(argv[2]) == 1;
    struct Shape { const char* name; int H, C; } shapes[] = {{"l2.3x3", 28, 128}, {"l3.3x3", 14, 256}, {"l4.3x3", 7, 512}};
    for (auto& sh : shapes) {
        const int H = sh.H, C = sh.C, K = sh.C;
        const long M = (long)N * H * H;
        std::vector<uint16_t> hx(M * C), hw((long)K * 9 * C);
        srand(1);
        for (auto& v : hx) { const float u = (rand() / (float)RAND_MAX - 0.5f) * 2.f; v = f2bf(dense ? u : (u > 0 ? u : 0.f)); }
        for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
        bf16 *x, *w, *y;
        float* stats;
        const size_t sb = 4096 * 2 * K * 4;
        hipMalloc(&x, M * C * 2); hipMalloc(&w, (long)K * 9 * C * 2); hipMalloc(&y, M * K * 2); hipMalloc(&stats, sb);
        hipMemcpy(x, hx.data(), M * C * 2, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), (long)K * 9 * C * 2, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {     // 0 forward + statistics, 1 data gradient, 2 accumulating data gradient
            double us[3];
            std::vector<uint16_t> out[3];
            std::vector<float> st[3];
            for (int v = 0; v < 3; ++v) {          // 0: lh2 as shipped, 1: lh2 with 196-pixel tiles, 2: lh4
                primia_set_option("lh2_bm", v == 1 ? 196 : 0);
                primia_set_option("lh4", 0);          // (the lh2 legs must not be re-routed to lh4)
                hipMemset(y, 0, M * K * 2);
                hipMemset(stats, 0, sb);
                auto launch = [&]() {
                    float* sp = mode == 0 ? stats : nullptr;
                    return v == 2 ? conv3x3_lh4_dispatch(x, w, y, N, H, H, C, K, mode > 0, mode == 2, 0, sp, nullptr)
                                  : conv3x3_lh2_dispatch(x, w, y, N, H, H, C, K, mode > 0, mode == 2, 0, sp, nullptr);
                };
                if (launch() != 0) { printf("%s mode %d: not served\n", sh.name, mode); return 1; }
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
                out[v].resize(M * K); st[v].resize(sb / 4);
                hipMemcpy(out[v].data(), y, M * K * 2, hipMemcpyDeviceToHost);
                hipMemcpy(st[v].data(), stats, sb, hipMemcpyDeviceToHost);
                for (int i = 0; i < 3; ++i) launch();
                hipDeviceSynchronize();
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                const int reps = 20;
                hipEventRecord(e0);
                for (int i = 0; i < reps; ++i) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                us[v] = ms * 1e3 / reps;
            }
            const bool same = mode == 2 || (memcmp(out[1].data(), out[2].data(), M * K * 2) == 0 &&
                                            memcmp(st[1].data(), st[2].data(), sb) == 0);
            long ndiff = 0;
            if (!same && mode != 2) for (long i = 0; i < M * K; ++i) ndiff += out[1][i] != out[2][i];
            const double fl = 2.0 * M * K * C * 9;
            printf("%s %s: lh2 %6.1f us (%4.0f TF)   lh2 196 px %6.1f us (%4.0f)   lh4 %6.1f us (%4.0f TF)   %s (%ld of %ld outputs differ)\n",
                   sh.name, mode == 0 ? "fwd+stats" : mode == 1 ? "dgrad    " : "dgrad+=  ", us[0], fl / us[0] / 1e6, us[1],
                   fl / us[1] / 1e6, us[2], fl / us[2] / 1e6, same ? "bit-identical" : "DIFFER", ndiff, M * K);
        }
        hipFree(x); hipFree(w); hipFree(y); hipFree(stats);
    }
    return 0;
}
