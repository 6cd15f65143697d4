// Stand-alone timing / phase attribution of conv_wgrad_patch33_kernel (csrc/conv_wgrad_patch.hip) on ResNet-18's
// 3x3 / stride-1 layers (batch 256, bf16).  The kernel source is compiled INTO this program, so the experiment switches
// are compile-time:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWGP33_DBG=n] [-DWGP33_PROF] tools/micro/wgp33_bench.hip -o wgp33_bench
//   WGP33_DBG bits: 1 no epilogue, 2 no DMA after the prologue, 4 no MFMA, 8 no fragment reads
//   WGP33_PROF    : per-wave cycles: DMA issue / vmcnt wait / barrier wait / compute / epilogue / prologue
// Data: x = relu of a normal variate (half zeros, like the network's activations), dy = small normal values; argv[2] = 1
// uses full-range random data on both sides (the matrix pipe then draws more power and the clock drops).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "../../primia_amd/csrc/conv_wgrad_patch.hip"
#include "../../primia_amd/csrc/options.hip"   // the option table the dispatch code reads

using namespace primia;

static uint16_t f2bf(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = v.u; u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16);
}
static float gauss() {
    float a = (rand() + 1.f) / (RAND_MAX + 2.f), b = rand() / (float)RAND_MAX;
    return sqrtf(-2.f * logf(a)) * cosf(6.2831853f * b);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    const int dense = argc > 2 ? atoi(argv[2]) : 0;
    const int group = argc > 3 ? atoi(argv[3]) : 1;      // > 1: that many layers of the shape in ONE launch
    struct Shape { const char* name; int H, C; } shapes[] = {{"l1.3x3", 56, 64}, {"l2.3x3", 28, 128}, {"l3.3x3", 14, 256}, {"l4.3x3", 7, 512}};
    for (auto& sh : shapes) {
        const int H = sh.H, C = sh.C, K = sh.C;
        const long M = (long)N * H * H;
        std::vector<uint16_t> hx(M * C), hy(M * K);
        srand(1);
        for (auto& v : hx) { float g = gauss(); v = f2bf(dense ? g : (g > 0 ? g : 0.f)); }
        for (auto& v : hy) v = f2bf(gauss() * (dense ? 1.f : 1e-3f));
        bf16 *x, *dy;
        float *dw, *ws;
        hipMalloc(&x, M * C * 2); hipMalloc(&dy, M * K * 2); hipMalloc(&dw, (long)K * 9 * C * 4);
        hipMemcpy(x, hx.data(), M * C * 2, hipMemcpyHostToDevice);
        hipMemcpy(dy, hy.data(), M * K * 2, hipMemcpyHostToDevice);
        WgradParams p{};
        p.x = x; p.dy = dy; p.dw = dw;
        p.N = N; p.H = H; p.W = H; p.C = C; p.K = K; p.R = 3; p.S = 3; p.stride = 1; p.pad = 1; p.Ho = H; p.Wo = H;
        p.klen = 9 * C; p.Md = M; p.ntaps = 9;
        int ng = group > 1 ? wgrad_patch_group_size(p, group) : 1;
        const size_t wsb = ng > 1 ? wgrad_patch_group_ws_bytes(p, ng) : wgrad_patch_ws_bytes(p);
        hipMalloc(&ws, wsb);
        p.ws = ws; p.ws_bytes = wsb;
        WgradParams pg[4] = {p, p, p, p};
        auto launch = [&]() { return ng > 1 ? wgrad_patch_group_dispatch(pg, ng, 0) : wgrad_patch_dispatch(p, 0); };
        if (launch() != 0) { printf("%s: not served\n", sh.name); continue; }
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 20;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, tf = 2.0 * ng * M * K * C * 9 / (us * 1e-6) / 1e12;
        printf("%s N=%d kernel+reduce: %7.1f us  %6.0f TF/s  (DBG=%d, %s data, %d layer(s) per launch: %.1f us per layer)\n",
               sh.name, N, us, tf, (int)WGP33_DBG, dense ? "dense" : "relu-like", ng, us / ng);
#ifdef WGP33_PROF
        {
            const PatchGeom g = patch_geom(p);
            const int nblk = g.combos * g.nsplit;
            unsigned long long* prof;
            hipMalloc(&prof, (size_t)nblk * 8 * 6 * 8);
            hipMemset(prof, 0, (size_t)nblk * 8 * 6 * 8);
            hipMemcpyToSymbol(HIP_SYMBOL(wgp33_prof_buffer_dev), &prof, sizeof(prof));
            launch();
            hipDeviceSynchronize();
            std::vector<unsigned long long> h((size_t)nblk * 8 * 6);
            hipMemcpy(h.data(), prof, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long* none = nullptr;
            hipMemcpyToSymbol(HIP_SYMBOL(wgp33_prof_buffer_dev), &none, sizeof(none));
            const char* names[6] = {"dma issue", "vmcnt wait", "barrier", "compute", "epilogue", "prologue"};
            for (int half = 0; half < 2; ++half) {
                double sum[6] = {0, 0, 0, 0, 0, 0};
                for (int b = 0; b < nblk; ++b)
                    for (int w = 4 * half; w < 4 * half + 4; ++w)
                        for (int k = 0; k < 6; ++k) sum[k] += (double)h[((size_t)b * 8 + w) * 6 + k];
                double tot = 0;
                for (int k = 0; k < 6; ++k) tot += sum[k];
                printf("   half %d (%s): total %8.0f clk/wave |", half, half ? "compute then issue" : "issue then compute", tot / (nblk * 4.0));
                for (int k = 0; k < 6; ++k) printf(" %s %5.1f%%", names[k], 100.0 * sum[k] / tot);
                printf("\n");
            }
            hipFree(prof);
        }
#endif
        hipFree(x); hipFree(dy); hipFree(dw); hipFree(ws);
    }
    return 0;
}
