// A/B of conv3x3_lh3.hip (weight fragments straight to registers) against conv3x3_lh2.hip on the three wide ResNet-18
// stages (batch 256, bf16, random data): bit-identity of outputs and BatchNorm partials, then interleaved timing rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/lh3_bench.hip -o tools/micro/lh3_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../primia_amd/csrc/conv3x3_lh2.hip"
#include "experiments/conv3x3_lh3.hip"
#include "../../primia_amd/csrc/options.hip"

using namespace primia;

static uint16_t f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u; u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    struct Shape { const char* name; int H, C; } shapes[] = {{"l2.3x3", 28, 128}, {"l3.3x3", 14, 256}, {"l4.3x3", 7, 512}};
    int bad = 0;
    for (auto& sh : shapes) {
        const int H = sh.H, C = sh.C, K = sh.C;
        const long M = (long)N * H * H;
        std::vector<uint16_t> hx(M * C), hw((long)K * 9 * C), hy0(M * K);
        srand(1);
        for (auto& v : hx) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        for (auto& v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.1f);
        for (auto& v : hy0) v = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        // packed weight image of lh3: [n-tile][tap][chunk][wn][q = 2 i + h][lane = 16 fg + fr][8]
        std::vector<uint16_t> hwp((long)K * 9 * C);
        {
            const int nch = C / 64;
            for (int nt = 0; nt < K / 128; ++nt)
                for (int tap = 0; tap < 9; ++tap)
                    for (int c = 0; c < nch; ++c)
                        for (int wn = 0; wn < 2; ++wn)
                            for (int q = 0; q < 8; ++q)
                                for (int lane = 0; lane < 64; ++lane)
                                    for (int e = 0; e < 8; ++e) {
                                        const int i = q >> 1, h = q & 1, fr = lane & 15, fg = lane >> 4;
                                        const long row = nt * 128 + wn * 64 + 16 * i + fr;
                                        const long col = (long)tap * C + c * 64 + h * 32 + fg * 8 + e;
                                        const long dst = ((((((long)nt * 9 + tap) * nch + c) * 2 + wn) * 8 + q) * 64 + lane) * 8 + e;
                                        hwp[dst] = hw[row * 9 * C + col];
                                    }
        }
        bf16 *x, *w, *y, *y0, *wp;
        hipMalloc(&wp, (long)K * 9 * C * 2);
        hipMemcpy(wp, hwp.data(), (long)K * 9 * C * 2, hipMemcpyHostToDevice);
        float* stats;
        const size_t stat_bytes = 4096 * 2 * K * 4;
        hipMalloc(&x, M * C * 2); hipMalloc(&w, (long)K * 9 * C * 2); hipMalloc(&y, M * K * 2); hipMalloc(&y0, M * K * 2);
        hipMalloc(&stats, stat_bytes);
        hipMemcpy(x, hx.data(), M * C * 2, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), (long)K * 9 * C * 2, hipMemcpyHostToDevice);
        hipMemcpy(y0, hy0.data(), M * K * 2, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {     // 0 forward + statistics, 1 data gradient, 2 accumulating data gradient
            auto launch = [&](int gen) {
                return gen == 3 ? conv3x3_lh3_dispatch(x, wp, y, N, H, H, C, K, mode > 0, mode == 2, 0, mode == 0 ? stats : nullptr, nullptr)
                                : conv3x3_lh2_dispatch(x, w, y, N, H, H, C, K, mode > 0, mode == 2, 0, mode == 0 ? stats : nullptr, nullptr);
            };
            std::vector<uint16_t> out[2];
            std::vector<float> st[2];
            for (int g = 0; g < 2; ++g) {
                hipMemcpy(y, y0, M * K * 2, hipMemcpyDeviceToDevice);
                hipMemset(stats, 0, stat_bytes);
                if (launch(2 + g) != 0) { printf("%s mode %d gen %d: not served\n", sh.name, mode, 2 + g); continue; }
                hipDeviceSynchronize();
                out[g].resize(M * K); st[g].resize(stat_bytes / 4);
                hipMemcpy(out[g].data(), y, M * K * 2, hipMemcpyDeviceToHost);
                hipMemcpy(st[g].data(), stats, stat_bytes, hipMemcpyDeviceToHost);
            }
            const bool same = out[0].size() && out[0] == out[1] && !memcmp(st[0].data(), st[1].data(), stat_bytes);
            if (!same) {
                ++bad;
                long nd = 0, first = -1;
                for (long i = 0; i < (long)out[0].size() && i < (long)out[1].size(); ++i)
                    if (out[0][i] != out[1][i]) { if (first < 0) first = i; ++nd; }
                printf("%s mode %d: MISMATCH  %ld of %ld outputs differ (first at row %ld col %ld)\n", sh.name, mode, nd, M * K,
                       first / K, first % K);
            }
            double us[2] = {0, 0};
            const int rounds = 3, reps = 20;
            for (int r = 0; r < rounds; ++r)
                for (int g = 0; g < 2; ++g) {
                    for (int i = 0; i < 3; ++i) launch(2 + g);
                    hipDeviceSynchronize();
                    hipEvent_t e0, e1;
                    hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0);
                    for (int i = 0; i < reps; ++i) launch(2 + g);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    us[g] += ms * 1e3 / reps / rounds;
                    hipEventDestroy(e0); hipEventDestroy(e1);
                }
#ifdef LH2_PROF
            {
                unsigned long long* prof;
                hipMalloc(&prof, 256 * 8 * 5 * 8);
                hipMemset(prof, 0, 256 * 8 * 5 * 8);
                lh3_prof_buffer = prof;
                launch(3);
                hipDeviceSynchronize();
                lh3_prof_buffer = nullptr;
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
                    if (n) printf("    lh3 %c waves: load %8.0f  matrix %8.0f  barrier-wait %8.0f  write-back+setup %8.0f  weight-wait %8.0f cycles\n",
                                  half ? 'B' : 'A', b[0] / n, b[1] / n, b[2] / n, b[3] / n, b[4] / n);
                }
                hipFree(prof);
            }
#endif
            const double fl = 2.0 * M * K * C * 9;
            printf("%s N=%d %s: lh2 %6.1f us %5.0f TF/s | lh3 %6.1f us %5.0f TF/s  %s\n", sh.name, N,
                   mode == 0 ? "fwd+stats" : mode == 1 ? "dgrad    " : "dgrad+=  ", us[0], fl / us[0] / 1e6, us[1], fl / us[1] / 1e6,
                   same ? "bit-identical" : "DIFFERENT");
        }
        hipFree(x); hipFree(w); hipFree(y); hipFree(y0); hipFree(stats);
    }
    return bad ? 1 : 0;
}
