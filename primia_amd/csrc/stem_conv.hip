// The stem convolution conv1 = Conv2d(3, 64, 7, stride 2, pad 3) (torchlib/models.py:371-372) on a
// spatially PADDED channels-last input, bf16, gfx950.
//
// Input layout "NHWC4p": [N][H + 6][W + 8][4] — 3 zero rows above/below, 3 zero columns left and 5
// right, 4th channel zero.  With the padding physically present
//   * no tap ever needs a bounds check,
//   * the 8-pixel x 4-channel group that one kernel row contributes to output pixel (oy, ox) is the 64
//     contiguous bytes starting at padded pixel (2*oy + r, 2*ox): 16-byte aligned in memory and in LDS.
//
// Forward (implicit GEMM, M = 64 out-channels, N = pixels, reduction = 7 kernel rows x 32 elements in
// the stem weight layout of conv_common.h): an 8-wave block keeps the whole filter in registers
// (wave (kh, pq): out-channels 32*kh..+31, 56 VGPRs) and streams 8 x 16 output patches:
//     stage  : the patch's 21 x 40 input pixels (6.6 KiB) by LDS-DMA — every input pixel is fetched
//              1.5x instead of the 49/4 = 12x of a per-row im2col staging
//     compute: the B fragment of (output row, kernel row r) is read STRAIGHT from the staged input:
//              lane (ox, fg) reads the 16 bytes at ((2*oy + r) * 40 + 2*ox + 2*fg) * 8 — no im2col copy;
//              wave (kh, pq) owns output rows 2*pq, 2*pq+1: per kernel row 2 ds_read_b128 feed 4 MFMAs
//     store  : through LDS, whole 2-KiB output rows.
// The kernel is bound by the 411 MB it has to write at batch 256, not by the MFMAs.
#include <stdlib.h>

#include "conv_common.h"

namespace primia {

__device__ __attribute__((aligned(16))) const unsigned char kStemZeroPage[16] = {0};

__device__ __forceinline__ void stem_dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

struct StemFwdParams {
    const bf16* xp;   // [N][Hp][Wp][4]
    const bf16* wt;   // [64][256] stem forward layout
    bf16* y;          // [N][Ho][Wo][64]
    int N, Hp, Wp, Ho, Wo;
    int PH, PW, PPI;  // 8 x 16 patches per image column / row / image
    int total, per_block;
};

__global__ __launch_bounds__(512) void stem_conv_fwd_kernel(StemFwdParams p) {
    constexpr int STAGES = 3;
    constexpr int PATCH = 7 * 1024;       // 21 rows x 320 B = 6720 B staged by 7 DMA instructions
    constexpr int STAGE = 2 * PATCH;      // two patches per stage
    constexpr int OUTB = 2 * 128 * 128;   // output rows of one stage: 2 patches x 128 pixels x 128 B
    extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES * STAGE + 2 * OUTB
    char* const sout = smem + STAGES * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, pq = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    const int t0 = blockIdx.x * p.per_block;
    int t1 = t0 + p.per_block;
    if (t1 > p.total) t1 = p.total;
    const int nstages = (t1 - t0 + 1) >> 1;
    if (nstages <= 0) return;

    // ---- weights -> registers: A fragment (r, i): row 32*kh + 16*i + fr, elements r*32 + 8*fg .. +7 ----
    bf16x8_t wreg[7][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16* wrow = p.wt + (long)(32 * kh + 16 * i + fr) * 256 + 8 * fg;
#pragma unroll
        for (int r = 0; r < 7; ++r) wreg[r][i] = *(const bf16x8_t*)(wrow + r * 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    struct Cursor {
        int n, ph, pw, t;
    };
    auto make_cursor = [&](int t) {
        Cursor c;
        c.t = t;
        c.n = t / p.PPI;
        const int rem = t - c.n * p.PPI;
        c.ph = rem / p.PW;
        c.pw = rem - c.ph * p.PW;
        return c;
    };
    auto advance = [&](Cursor& c) {
        ++c.t;
        if (++c.pw == p.PW) {
            c.pw = 0;
            if (++c.ph == p.PH) {
                c.ph = 0;
                ++c.n;
            }
        }
    };
    Cursor cs = make_cursor(t0), cw = make_cursor(t0);  // staging / write-back

    // ---- staging: 14 DMA instructions per stage (2 patches x 7); wave w issues instruction w and, for
    // w < 6, instruction w + 8.  Lane G = 64*j + lane of a patch covers row G / 20, 16-byte column G % 20.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto stage = [&](int buf) {
        long base[2];
        bool live[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            live[q] = cs.t < t1;
            base[q] = ((long)(cs.n * p.Hp + cs.ph * 16) * p.Wp + cs.pw * 32) * 4;
            advance(cs);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = wave + 8 * it;  // wave-uniform
            if (idx >= 14) break;
            const int q = idx >= 7;
            const int G = (idx - 7 * q) * 64 + lane;
            const int row = G / 20, c16 = G - row * 20;
            const bf16* g = (live[q] && G < 420) ? p.xp + base[q] + ((long)row * p.Wp + 2 * c16) * 4
                                                 : (const bf16*)kStemZeroPage;
            stem_dma16(g, __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + idx * 1024));
        }
    };

    // B fragment of output row (2*pq + j), kernel row r: byte offset ((2*(2*pq + j) + r) * 20 + fr + fg) * 16
    const int offb0 = ((4 * pq) * 20 + fr + fg) * 16;  // + (2*j + r) * 320
    const int opix0 = (2 * pq) * 16 + fr;              // + 16*j : pixel of the patch this lane's results belong to

    auto compute = [&](int buf, int obuf) {
        const char* sb = smem + buf * STAGE;
        f32x4 acc[2][2][2];  // [patch][row j][K fragment i]
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[q][j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            bf16x8_t b[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) b[q][j] = *(const bf16x8_t*)(sb + q * PATCH + offb0 + (2 * j + r) * 320);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[q][j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[r][i], b[q][j], acc[q][j][i], 0, 0, 0);
        }
        // results -> LDS rows (bf16): pixel opix, 8-byte column 8*kh + 4*i + fg, chunk-swizzled like conv3x3_c64
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int opix = opix0 + 16 * j;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    u32x2 o;
                    o[0] = (uint32_t)f32_to_bf16(acc[q][j][i][0]) | ((uint32_t)f32_to_bf16(acc[q][j][i][1]) << 16);
                    o[1] = (uint32_t)f32_to_bf16(acc[q][j][i][2]) | ((uint32_t)f32_to_bf16(acc[q][j][i][3]) << 16);
                    const int col = 8 * kh + 4 * i + fg;
                    *(u32x2*)(sout + obuf * OUTB + q * 16384 + opix * 128 +
                              ((((col >> 1) ^ ((opix >> 1) & 7)) << 4) | ((col & 1) << 3))) = o;
                }
            }
    };

    // write-back of one stage: 2 patches x 8 output rows of 16 pixels x 128 B = 2 KiB contiguous each; wave w
    // stores row w of both patches (two 1-KiB instructions per row)
    auto writeback = [&](int obuf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool live = cw.t < t1;
            bf16* rowp = p.y + ((long)(cw.n * p.Ho + cw.ph * 8 + wave) * p.Wo + cw.pw * 16) * 64;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int px = hlf * 8 + (lane >> 3), c16 = lane & 7;
                const int opix = wave * 16 + px;
                const u32x4 v = *(const u32x4*)(sout + obuf * OUTB + q * 16384 + opix * 128 + ((c16 ^ ((opix >> 1) & 7)) << 4));
                if (live) *(u32x4*)(rowp + px * 64 + c16 * 8) = v;
            }
            advance(cw);
        }
    };

    // 3-deep LDS ring, one raw barrier per stage, counted vmcnt (stores count too).  Per iteration a wave issues,
    // in this order, 4 row stores (write-back of the previous stage) and d DMA instructions (d = 2 for waves
    // 0..5, 1 for waves 6, 7).  At the top of iteration s the operations newer than DMA(s) are the stores of
    // iteration s-1 (s >= 2) and DMA(s+1).  The last stage (a patch may be dead) uses vmcnt(0).
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nstages) stage(s);
    int cur = 0, nxt = STAGES - 1;
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) {
            if (wave < 6) {
                if (s >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                if (s >= 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s > 0) writeback((s - 1) & 1);
        if (s + STAGES - 1 < nstages) stage(nxt);
        compute(cur, s & 1);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }
    __syncthreads();
    writeback((nstages - 1) & 1);
}

}  // namespace primia

using namespace primia;

extern "C" {

int primia_stem_pad_dims(int H, int W, int* Hp, int* Wp) {
    PRIMIA_REQUIRE(H > 0 && W > 0 && Hp && Wp);
    *Hp = H + 6;
    *Wp = W + 8;
    return PRIMIA_OK;
}

int primia_stem_conv_fwd(const void* x_padded, const void* w_fwd, void* y, int N, int H, int W, int dtype,
                         primia_stream_t stream) {
    PRIMIA_REQUIRE(x_padded && w_fwd && y && N > 0 && H > 0 && W > 0);
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_UNSUPPORTED;
    // 8 x 16 output patches must tile the output exactly (every legal PriMIA input size is a multiple of 32)
    if (H % 32 != 0 || W % 32 != 0) return PRIMIA_ERR_UNSUPPORTED;
    if ((long)N * (H + 6) * (W + 8) * 4 >= (1L << 31) || (long)N * (H / 2) * (W / 2) * 64 >= (1L << 31))
        return PRIMIA_ERR_UNSUPPORTED;
    StemFwdParams p;
    p.xp = (const bf16*)x_padded; p.wt = (const bf16*)w_fwd; p.y = (bf16*)y;
    p.N = N; p.Hp = H + 6; p.Wp = W + 8; p.Ho = H / 2; p.Wo = W / 2;
    p.PH = p.Ho / 8; p.PW = p.Wo / 16; p.PPI = p.PH * p.PW;
    p.total = N * p.PPI;
    static const int target = getenv("PRIMIA_STEM_BLOCKS") ? atoi(getenv("PRIMIA_STEM_BLOCKS")) : 256;
    long per = (p.total + target - 1) / target;
    per = (per + 1) & ~1L;
    if (per < 2) per = 2;
    p.per_block = (int)per;
    const int grid = (int)((p.total + per - 1) / per);
    const size_t lds = (size_t)3 * 2 * 7 * 1024 + 2 * 2 * 128 * 128;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)stem_conv_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    stem_conv_fwd_kernel<<<grid, 512, lds, (hipStream_t)stream>>>(p);
    return launch_status();
}

}  // extern "C"
