// What bounds a tile write-back burst?  Every active CU's 8 waves store a [392 px x 128 ch] bf16 tile (100 KB) the way
// an MFMA epilogue can: 16 bytes per lane, and per store instruction either
//   P0  16 rows x 64 B   (conv3x3_lh2.hip after the v_permlane16_swap transpose)
//   P1   8 rows x 128 B  (full cache lines: what one more exchange between the two 8-lane halves of a row would give)
//   P2   4 rows x 256 B  (whole pixel rows: an LDS-staged write-back)
//   P3  16 rows x 32 B as 8-byte stores (the raw accumulator layout)
// with all 256 CUs active or only every 8th (one per XCD slot) — chip-wide HBM bound vs per-CU store-path bound.
// Prints cycles from the first store to vmcnt(0) of the slowest wave, and the implied bytes per cycle per CU.
//   hipcc --offload-arch=gfx950 -O3 store_burst.hip -o store_burst
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int P>
__global__ __launch_bounds__(512) void burst(char* out, int active_mod, int rounds, unsigned long long* cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x % active_mod) return;
    const long tile_bytes = 392L * 256;                       // [392][128] bf16
    unsigned long long t0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        char* tile = out + ((long)blockIdx.x * rounds + r) * tile_bytes;
        // each wave owns 49 pixel rows (256 B each) of the tile: 12.25 KB = 12-13 instructions of 1 KB
        const int row0 = wave * 49;
        const u32x4 v = {(unsigned)tid, (unsigned)r, 3u, 4u};
        if (P == 0) {          // 16 rows x 64 B per instruction, 4 instructions cover 16 rows x 256 B
            for (int g = 0; g < 3; ++g)
                for (int q = 0; q < 4; ++q) {
                    const int row = row0 + 16 * g + (lane & 15);
                    *(u32x4*)(tile + (long)row * 256 + q * 64 + (lane >> 4) * 16) = v;
                }
        } else if (P == 1) {   // 8 rows x 128 B
            for (int g = 0; g < 6; ++g)
                for (int q = 0; q < 2; ++q) {
                    const int row = row0 + 8 * g + (lane >> 3);
                    *(u32x4*)(tile + (long)row * 256 + q * 128 + (lane & 7) * 16) = v;
                }
        } else if (P == 2) {   // 4 rows x 256 B
            for (int g = 0; g < 12; ++g) {
                const int row = row0 + 4 * g + (lane >> 4);
                *(u32x4*)(tile + (long)row * 256 + (lane & 15) * 16) = v;
            }
        } else if (P >= 4) {   // P2's rows with cache-policy bits: 4 nt, 5 sc1, 6 sc0 sc1, 7 sc0 sc1 nt
            for (int g = 0; g < 12; ++g) {
                const int row = row0 + 4 * g + (lane >> 4);
                char* ptr = tile + (long)row * 256 + (lane & 15) * 16;
                if (P == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(v) : "memory");
                if (P == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
                if (P == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(v) : "memory");
                if (P == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(ptr), "v"(v) : "memory");
            }
        } else {               // 16 rows x 32 B, 8-byte stores
            const u32x2 w = {(unsigned)tid, (unsigned)r};
            for (int g = 0; g < 3; ++g)
                for (int q = 0; q < 8; ++q) {
                    const int row = row0 + 16 * g + (lane & 15);
                    *(u32x2*)(tile + (long)row * 256 + q * 32 + (lane >> 4) * 8) = w;
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    unsigned long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int rounds = 2;
    char* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256L * rounds * 392 * 256 + 4096);
    hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<unsigned long long> h(256 * 8);
    for (int mod : {1, 8}) {
        for (int P = 0; P < 8; ++P) {
            double best = 1e30, avg = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(cyc, 0, 256 * 8 * 8);
                if (P == 0) burst<0><<<256, 512>>>(out, mod, rounds, cyc);
                if (P == 1) burst<1><<<256, 512>>>(out, mod, rounds, cyc);
                if (P == 2) burst<2><<<256, 512>>>(out, mod, rounds, cyc);
                if (P == 3) burst<3><<<256, 512>>>(out, mod, rounds, cyc);
                if (P == 4) burst<4><<<256, 512>>>(out, mod, rounds, cyc);
                if (P == 5) burst<5><<<256, 512>>>(out, mod, rounds, cyc);
                if (P == 6) burst<6><<<256, 512>>>(out, mod, rounds, cyc);
                if (P == 7) burst<7><<<256, 512>>>(out, mod, rounds, cyc);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long mx = 0;
                for (auto v : h) mx = v > mx ? v : mx;
                const double c = (double)mx / rounds;
                if (rep) { best = c < best ? c : best; avg += c / 4; }
            }
            printf("CUs active 1/%d  pattern P%d: %8.0f cycles per 100 KB tile (best %8.0f)  = %5.1f B/clk/CU\n", mod, P, avg, best,
                   392.0 * 256 / avg);
        }
    }
    return 0;
}
