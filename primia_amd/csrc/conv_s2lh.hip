// Transition blocks (torchlib/models.py:219-235, 268-284, 433-436): the 3x3 / stride-2 / pad-1 conv1 and the 1x1 / stride-2
// downsample of layer2.0 / 3.0 / 4.0 — forward (both in one launch, with BatchNorm partial sums) and data gradient (both in
// one launch) — as STRIDE-1 problems on the four parity planes of the full-resolution tensor, served by the linear-halo
// machinery of conv3x3_lh4.hip (8 matrix waves that never issue a DMA, 4 loader waves that do nothing else, one barrier per
// step).  Round 5; replaces conv_igemm_pair_kernel / conv_igemm_kernel<dgrad> on these six launches (0.17-0.22 of the MFMA
// peak, 13 vector + 15 scalar instructions per MFMA, 1.4-1.9x the algorithmic HBM traffic: profiles/r04_inst_mix_latest.txt).
//
// Parity planes WITHOUT a second layout in HBM.  Output pixel (ho, wo) of a 3x3 / 2 convolution reads input rows 2ho-1 .. 2ho+1:
// row 2ho-1 is row ho-1 of the odd-row plane, 2ho is row ho of the even plane, 2ho+1 is row ho of the odd plane (columns alike).
// So plane (pr, pc) = x[:, pr::2, pc::2, :] is a [N, H/2, W/2, C] image and serves
//     (even, even): tap (1,1)                      at shift  0          1 step
//     (even, odd ): taps (1,0) (1,2)               at shifts -1, 0      2 steps
//     (odd , even): taps (0,1) (2,1)               at shifts -Wo, 0     2 steps
//     (odd , odd ): taps (0,0) (0,2) (2,0) (2,2)   at shifts -Wo-1, -Wo, -1, 0    4 steps
// of a 2 x 2 / stride-1 window on the half-resolution grid.  In NHWC a plane pixel's 64-channel slice is one contiguous
// 128-byte line at a stride of two pixels: the loader waves' LDS-DMA GATHERS a plane's pixel run (+ Wo + 1 pixels of halo on
// ONE side) straight from the ordinary tensor — per-lane source offsets, linear LDS destination — at full line efficiency.
// Nothing upstream changes layout, every input line is fetched once per 128-channel output tile.  The downsample is the
// centre tap of plane (even, even) with its own filter and output: extra column tiles of the same launch.
// The data gradient is the transpose: dx's four parity classes are stride-1 problems on dy (a dense half-resolution tensor,
// staged as an ordinary linear halo with the extra pixels on the TRAILING side) with 1 / 2 / 2 / 4 taps; the downsample's
// gradient rides on class (even, even) as extra reduction chunks over dy_ds; a class's tile is scattered to its pixels of dx.
//
// One kernel executes all of it from a STEP PROGRAM (built on the host per launch, 1 word per step): a tile's column
// descriptor names its program; a step = (64-channel chunk of a source, tap, slot shift) x 192 pixels x 128 "virtual" output
// channels.  For dx with 64 channels (layer2.0) a tile carries two classes side by side (64 + 64 virtual channels; a tap only
// one of them has is a half-zero weight tile — 75 % useful MFMAs on that one launch).  Weights are read from the EXISTING
// kernel layouts ([K][R][S][C] forward, [C][R][S][K] data gradient): no new copies, the fused SGD tail is untouched.
// Tiles cost 1 .. 36 steps: blocks take contiguous tile ranges of equal COST, column tiles interleaved cheap / expensive.
#include <stdlib.h>

#include "conv3x3_lh.h"
#include "options.h"

// compile-time experiment switches (tools/s2lh_variants.sh builds and times the variants on one box)
#ifndef PRIMIA_PROBE
#define PRIMIA_PROBE 0
#endif
#ifndef S2_DBG
#define S2_DBG PRIMIA_PROBE   // the s2lh_dbg phase switches (wrong results when set) exist in probe builds only
#endif
#ifndef S2_PROG_KARG
#define S2_PROG_KARG 0    // 1: the loaders read the step words from the kernel arguments (scalar loads), not from LDS
#endif
#ifndef S2_WAIT_SIMPLE
#define S2_WAIT_SIMPLE 1  // 1: the loaders wait for everything issued before the current step (measured: 7 % faster than the exact count)
#endif
#ifndef S2_SCHED
#define S2_SCHED 0        // 1: scheduling barrier between a step's fragment reads and its MFMAs
#endif

namespace primia {

typedef int i32x4_t __attribute__((ext_vector_type(4)));

constexpr int kS2MaxCols = 12, kS2MaxTypes = 6, kS2MaxSteps = 40;
constexpr int kS2BM = 192;   // 12 pixel fragments: three per matrix wave

// step word
//   bits 0-1  shift: bit 0 = one column, bit 1 = one row (forward: towards smaller indices; data gradient: larger) — the same
//             bits name the validity the tap needs (column / row neighbour exists)
//   bit  2    last step of its chunk      bit 3  first step of its chunk      bits 4-6  steps in the chunk (valid at `first`)
//   bit  7    source 0 | 1                bits 8-9  parity plane pr * 2 + pc of the chunk (forward gather)
//   bits 10-13 64-channel slice of the source          bit 14  weight array 0 | 1
//   bits 15-20 / 21-26   weight rows 0-63 / 64-127 of the step tile: tap (4 bits), all-zero (1), row base +64 (1)
__host__ __device__ constexpr int s2_step_word(int sh, int last, int first, int nk, int src, int plane, int slice, int wsel,
                                               int tap0, int zero0, int rel0, int tap1, int zero1, int rel1) {
    return sh | (last << 2) | (first << 3) | (nk << 4) | (src << 7) | (plane << 8) | (slice << 10) | (wsel << 14) |
           (tap0 << 15) | (zero0 << 19) | (rel0 << 20) | (tap1 << 21) | (zero1 << 25) | (rel1 << 26);
}

struct S2Col {
    int type;      // step program
    int dsel;      // destination tensor 0 | 1 (forward: conv1 | downsample)
    int n0;        // first channel of the tile in a destination row
    int wrow0;     // first weight row
    int cls0, cls1;  // data gradient: parity class (pr * 2 + pc) of virtual channels 0-63 | 64-127
    int cost;      // steps
    int pad;
};

struct S2Params {
    const bf16* src[2];
    const bf16* wt[2];
    bf16* dst[2];
    float* stat[2];       // forward: BatchNorm partial sums [ntm][2][ld] of the values AS STORED (both or neither);
                          // data gradient with bnb_y (64-channel dx only): stat[0] = [ntm * ncol][2][64] backward sums, see below
    // data gradient, split tiles: dx of this launch is the gradient w.r.t. z = relu(bn(y) + identity) of the layer in FRONT of
    // the block (a residual layer whose forward pass left one ReLU-mask byte per 8 channels): with bnb_y set the write-back
    // also forms that BatchNorm's backward sums, sum g and sum g * xhat with g = dx AS STORED * mask bit — the reduction pass
    // over (y, dx, mask) is dropped (primia_conv2d_dgrad_pair_bnsums + primia_bn_bwd_mask_from_sums)
    const bf16* bnb_y;
    const uint8_t* bnb_mask;
    const float* bnb_mean;
    const float* bnb_invstd;
    int klen[2];          // elements per weight row
    int ld[2];            // elements per destination row
    int Cs;               // channels of the source(s)
    int mode;             // 0 forward, 1 data gradient
    int Ho, Wo, M2;       // half-resolution grid, N * Ho * Wo
    int split;            // data gradient, 64-channel dx: the two 64-channel halves of a tile are two classes
    unsigned magicWo, magicHo;   // ceil(2^32 / Wo), ceil(2^32 / Ho)
    int ntm, ncol, gcost, ntiles;
    int dbg;              // option s2lh_dbg (measurement only): 1 no halo DMA after the prologue, 2 no weight DMA, 4 no MFMA, 8 no stores,
                          // 16 no write-back at all, 32 no BatchNorm partials, 64 the forward halo addressed linearly (no gather)
    int nsteps[kS2MaxTypes];
    int prefix[kS2MaxCols + 1];   // cost at which column tile tc starts inside its row of tiles
    S2Col col[kS2MaxCols];
    int prog[kS2MaxTypes][kS2MaxSteps];
};

__device__ __forceinline__ void s2_dma(unsigned voff, i32x4_t rsrc, unsigned soff, unsigned lds_addr) {
    soff = __builtin_amdgcn_readfirstlane(soff);
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ i32x4_t s2_rsrc(const void* base, long bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4_t r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)(unsigned)(a >> 32) & 0xffff;
    r[2] = (int)(unsigned)(bytes > 0x7ffffff0L ? 0x7ffffff0L : bytes);
    r[3] = 0x00020000;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_readfirstlane(r[j]);
    return r;
}

constexpr int kS2Slots = 224;                       // 192 + Wo + 1 <= 221 live slots; the last slot is always zero
constexpr int kS2Rows = kS2Slots / 16;              // 14 DMA rows of 16 slots per 32-channel plane
constexpr int kS2Plane = kS2Slots * 64;             // one 32-channel half of every slot
constexpr int kS2Halo = 2 * kS2Plane;               // 28 KiB per halo buffer, ring of 3
constexpr int kS2Half = 128 * 64;
constexpr int kS2WTile = 2 * kS2Half;               // 16 KiB per step tile, ring of 4 (the loaders run three steps ahead)
constexpr int kS2WRing = 4;
constexpr int kS2OffW = 3 * kS2Halo;
constexpr int kS2OffScr = kS2OffW + kS2WRing * kS2WTile;
constexpr int kS2OffProg = kS2OffScr + 4 * 2 * 128 * 4;
constexpr int kS2OffCol = kS2OffProg + kS2MaxTypes * kS2MaxSteps * 4;      // S2Col[kS2MaxCols], then nsteps[kS2MaxTypes]
constexpr int kS2Lds = kS2OffCol + kS2MaxCols * 32 + kS2MaxTypes * 4;      // 157,016 B
constexpr int kS2ZeroSlot = kS2Slots - 1;
constexpr unsigned kS2Oob = 0xfffffff0u;
static_assert(kS2Lds <= 160 * 1024, "LDS");

// column descriptors / step counts out of LDS (wave-uniform index)
__device__ __forceinline__ int s2_col_field(const char* smem, int tc, int field) {
    return __builtin_amdgcn_readfirstlane(((const int*)(smem + kS2OffCol))[tc * 8 + field]);
}
__device__ __forceinline__ int s2_nsteps(const char* smem, int type) {
    return __builtin_amdgcn_readfirstlane(((const int*)(smem + kS2OffCol + kS2MaxCols * 32))[type]);
}

// first tile whose start cost is >= B (tiles of a row in column order, rows one after the other)
__device__ __forceinline__ int s2_first_tile(const S2Params& p, int B) {
    int tm = B / p.gcost;
    const int r = B - tm * p.gcost;
    int tc = 0;
    while (tc < p.ncol && p.prefix[tc] < r) ++tc;
    if (tc == p.ncol) {
        ++tm;
        tc = 0;
    }
    return tm * p.ncol + tc;
}

// flat pixel index of parity-grid position q (n * Ho + ho, wo) in class (pr, pc) of the full-resolution tensor
__device__ __forceinline__ unsigned s2_pix(unsigned q, unsigned n_ho, int Wo, int cls) {
    return 2u * q + 2u * n_ho * (unsigned)Wo + (unsigned)((cls >> 1) * 2 * Wo + (cls & 1));
}

// One MATRIX wave: JW = 3 pixel fragments starting at fragment F0 = 3 * (pixel group), channel half wn.
template <int JW>
__device__ __forceinline__ void s2_run(const S2Params& p, char* smem, int tile_first, int tile_count, const int F0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int Wo = p.Wo, Ho = p.Ho;
    const bool dg = p.mode == 1;
    const bool has_stat = p.stat[0] != nullptr && (!dg || p.bnb_y != nullptr);
    const int aoff = (wn * 64 + fr) * 64 + ((fg ^ lh_key(fr >> 2)) << 4);
    const int sgn = dg ? 1 : -1;
    int sj0 = 16 * F0 + fr + (dg ? 0 : Wo + 1);
    const int* prog = (const int*)(smem + kS2OffProg);

    f32x4 acc[4][JW];
    bf16x8_t a0[4], b[JW];
    int bad[JW];
    unsigned pm = 0;    // 3 bits per fragment: column neighbour exists | row neighbour exists | pixel is inside the tensor

    int m0 = 0, tm = 0, tc = 0;
    auto tile_setup = [&]() {
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            const int pl = 16 * (F0 + j) + fr;
            const unsigned q = (unsigned)(m0 + pl);
            const unsigned n_ho = __umulhi(q, p.magicWo);
            const int wo = (int)(q - n_ho * (unsigned)Wo);
            const int ho = (int)(n_ho - __umulhi(n_ho, p.magicHo) * (unsigned)Ho);
            unsigned v = 0;
            if (pl < kS2BM && m0 + pl < p.M2)
                v = 4u | (dg ? (wo < Wo - 1 ? 1u : 0u) | (ho < Ho - 1 ? 2u : 0u) : (wo > 0 ? 1u : 0u) | (ho > 0 ? 2u : 0u));
            m |= v << (3 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        pm = m;
    };

    // ---- write-back from the accumulators (conv3x3_lh4.hip's plain form: v_permlane16_swap transpose, whole 128-byte lines,
    //      stores first, BatchNorm partial sums while they drain); rows are dense (forward) or scattered to a class (dgrad) ----
    auto epilogue = [&]() {
        if ((S2_DBG ? p.dbg : 0) & 16) {
#pragma unroll
            for (int j = 0; j < JW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        const int hi8 = fr >> 3;
        const unsigned colin = (unsigned)(32 * hi8 + 16 * (fg & 1) + 8 * (fg >> 1));
        const int dsel = s2_col_field(smem, tc, 1);
        bf16* dst = dsel ? p.dst[1] : p.dst[0];
        const unsigned ld = (unsigned)(dsel ? p.ld[1] : p.ld[0]);
        const unsigned col0 = (unsigned)s2_col_field(smem, tc, 2) + ((dg && p.split) ? 0u : (unsigned)(wn * 64)) + colin;
        const int cls = s2_col_field(smem, tc, (dg && p.split && wn) ? 5 : 4);
        auto ror8 = [](uint32_t v) {
            return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);   // row_ror:8
        };
        float bs1[8], bs2[8], bmu[8], bis[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) bs1[k] = bs2[k] = bmu[k] = bis[k] = 0.f;
        if (dg && has_stat) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                bmu[k] = p.bnb_mean[col0 + k];
                bis[k] = p.bnb_invstd[col0 + k];
            }
        }
        auto row_elem = [&](int pl) -> unsigned {     // element offset of the row tile pixel pl is stored to
            const unsigned q = (unsigned)(m0 + pl);
            if (!dg) return q * ld;
            const unsigned n_ho = __umulhi(q, p.magicWo);
            return s2_pix(q, n_ho, Wo, cls) * ld;
        };
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            const int plA = 16 * (F0 + j) + (fr & 7), plB = plA + 8;
            const bool okA = plA < kS2BM && m0 + plA < p.M2, okB = plB < kS2BM && m0 + plB < p.M2;
            const unsigned eoA = row_elem(plA) + col0, eoB = row_elem(plB) + col0;
            u32x4 pc[2];
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                uint32_t x0 = (uint32_t)f32_to_bf16(acc[2 * bq][j][0]) | ((uint32_t)f32_to_bf16(acc[2 * bq][j][1]) << 16);
                uint32_t x1 = (uint32_t)f32_to_bf16(acc[2 * bq][j][2]) | ((uint32_t)f32_to_bf16(acc[2 * bq][j][3]) << 16);
                uint32_t y0 = (uint32_t)f32_to_bf16(acc[2 * bq + 1][j][0]) | ((uint32_t)f32_to_bf16(acc[2 * bq + 1][j][1]) << 16);
                uint32_t y1 = (uint32_t)f32_to_bf16(acc[2 * bq + 1][j][2]) | ((uint32_t)f32_to_bf16(acc[2 * bq + 1][j][3]) << 16);
                lh_swap16(x0, y0);
                lh_swap16(x1, y1);
                pc[bq] = u32x4{x0, x1, y0, y1};
            }
            u32x4 stA, stB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t give = hi8 ? pc[0][e] : pc[1][e];
                const uint32_t got = ror8(give);
                stA[e] = hi8 ? got : pc[0][e];
                stB[e] = hi8 ? pc[1][e] : got;
            }
            if (okA && !((S2_DBG ? p.dbg : 0) & 8)) __builtin_nontemporal_store(stA, (u32x4*)((char*)dst + (size_t)eoA * 2u));
            if (okB && !((S2_DBG ? p.dbg : 0) & 8)) __builtin_nontemporal_store(stB, (u32x4*)((char*)dst + (size_t)eoB * 2u));
            if (dg && has_stat) {
                // backward sums of the BatchNorm in front of the block, from the values just stored (store layout: this lane
                // holds channels col0 .. col0 + 7 of pixels A and B)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const bool ok = r ? okB : okA;
                    const unsigned eo = r ? eoB : eoA;
                    const u32x4 dzv = r ? stB : stA;
                    u32x4 yv = {0u, 0u, 0u, 0u};
                    unsigned mk = 0u;
                    if (ok) {
                        yv = *(const u32x4*)((const char*)p.bnb_y + (size_t)eo * 2u);
                        mk = p.bnb_mask[eo >> 3];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float yk = __uint_as_float((k & 1) ? (yv[k >> 1] & 0xffff0000u) : (yv[k >> 1] << 16));
                        const float dk = __uint_as_float((k & 1) ? (dzv[k >> 1] & 0xffff0000u) : (dzv[k >> 1] << 16));
                        const float g = ((mk >> k) & 1u) ? dk : 0.f;
                        bs1[k] += g;
                        bs2[k] += g * ((yk - bmu[k]) * bis[k]);
                    }
                }
            }
        }
        if (dg && has_stat) {
            // fold the 8 lanes that share these channels (fr & 7), park per pixel group; wn = 1 (the second class of a split tile:
            // the SAME 64 channels at other pixels) parks in the upper half of the 128-wide row, stat_combine adds the halves
            float* scr = (float*)(smem + kS2OffScr) + (F0 / 3) * 256;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = bs1[k], b = bs2[k];
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    const int ctrl = st == 0 ? 0xB1 : (st == 1 ? 0x4E : 0x141);      // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
                    if (st == 0) {
                        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xf, 0xf, true));
                        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0xB1, 0xf, 0xf, true));
                    } else if (st == 1) {
                        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xf, 0xf, true));
                        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x4E, 0xf, 0xf, true));
                    } else {
                        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x141, 0xf, 0xf, true));
                        b += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0x141, 0xf, 0xf, true));
                    }
                    (void)ctrl;
                }
                if ((fr & 7) == 0) {
                    const int ch = wn * 64 + (int)colin + k;
                    scr[ch] = a;
                    scr[128 + ch] = b;
                }
            }
        }
        if (!dg && has_stat && !((S2_DBG ? p.dbg : 0) & 32)) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 s1[4][2], s2[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) s1[i][h] = s2[i][h] = f32x2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const int pl = 16 * (F0 + j) + fr;
                const bool ok = pl < kS2BM && m0 + pl < p.M2;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 r = {bf16_to_f32(f32_to_bf16(acc[i][j][2 * h])), bf16_to_f32(f32_to_bf16(acc[i][j][2 * h + 1]))};
                        if (!ok) r = f32x2{0.f, 0.f};
                        s1[i][h] += r;
                        s2[i][h] += r * r;
                    }
            }
            float* scr = (float*)(smem + kS2OffScr) + (F0 / 3) * 256;   // pixel group 0..3
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = lh_row_sum(s1[i][e >> 1][e & 1]), t2 = lh_row_sum(s2[i][e >> 1][e & 1]);
                    if (fr == 0) {
                        const int ch = wn * 64 + 16 * i + 4 * fg + e;
                        scr[ch] = t1;
                        scr[128 + ch] = t2;
                    }
                }
        }
    };
    // waves 4-7, one barrier after every wave's write-back: thread -> (sum, channel); pixel groups added in the order 0,1,2,3
    auto stat_combine = [&](int tm_, int tc_) {
        if (!has_stat || wave < 4) return;
        const int dsel = s2_col_field(smem, tc_, 1), n0 = s2_col_field(smem, tc_, 2);
        const int t = tid & 255;
        const int q = t >> 7, ch = t & 127;
        const float* scr = (const float*)(smem + kS2OffScr);
        float s = scr[q * 128 + ch];
#pragma unroll
        for (int g = 1; g < 4; ++g) s += scr[g * 256 + q * 128 + ch];
        if (dg) {        // split tile: both 64-channel halves are the same channels; one partial row per (tm, column tile)
            if (ch < 64) {
                float s2v = scr[q * 128 + 64 + ch];
#pragma unroll
                for (int g = 1; g < 4; ++g) s2v += scr[g * 256 + q * 128 + 64 + ch];
                p.stat[0][((long)(tm_ * p.ncol + tc_) * 2 + q) * 64 + ch] = s + s2v;
            }
            return;
        }
        (dsel ? p.stat[1] : p.stat[0])[((long)tm_ * 2 + q) * (dsel ? p.ld[1] : p.ld[0]) + n0 + ch] = s;
    };

    int ring = 0, hbuf = 0;
    int prev_tm = 0, prev_tc = 0;
    bool pending_combine = false;
    __builtin_amdgcn_s_barrier();                  // the loaders' prologue has landed
    tm = __builtin_amdgcn_readfirstlane(tile_first / p.ncol);
    tc = __builtin_amdgcn_readfirstlane(tile_first - tm * p.ncol) - 1;
    for (int it = 0; it < tile_count; ++it) {
        if (++tc == p.ncol) {
            tc = 0;
            ++tm;
        }
        m0 = tm * kS2BM;
        const int type = s2_col_field(smem, tc, 0);
        const int ns = s2_nsteps(smem, type);
        tile_setup();
        int e_next = prog[type * kS2MaxSteps];
        for (int s = 0; s < ns; ++s) {
            asm volatile("" : "+v"(pm), "+v"(sj0));
            const int e = __builtin_amdgcn_readfirstlane(e_next);
            e_next = prog[type * kS2MaxSteps + (s + 1 < ns ? s + 1 : s)];      // (lands behind this step's MFMAs)
            const int sh = e & 3;
            const char* wp = smem + kS2OffW + ring * kS2WTile;
            bf16x8_t a1[4], b1[JW];
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[i] = *(const bf16x8_t*)(wp + (aoff + i * 1024));
            const int slot = sj0 + sgn * ((sh & 1) + (sh >> 1) * Wo);
            const int offt = slot * 64 + ((fg ^ lh_key(slot >> 2)) << 4) + hbuf * kS2Halo;
            const int zoff = kS2ZeroSlot * 64 + hbuf * kS2Halo;
            const unsigned need = (unsigned)sh | 4u;
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                bad[j] = (((pm >> (3 * j)) & need) == need) ? offt : zoff - j * 1024;
                b[j] = *(const bf16x8_t*)(smem + (bad[j] + j * 1024));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) a1[i] = *(const bf16x8_t*)(wp + (kS2Half + aoff + i * 1024));
#pragma unroll
            for (int j = 0; j < JW; ++j) b1[j] = *(const bf16x8_t*)(smem + (bad[j] + (j * 1024 + kS2Plane)));
            if (S2_SCHED) __builtin_amdgcn_sched_barrier(0);
            if (!((S2_DBG ? p.dbg : 0) & 4)) {
#pragma unroll
                for (int j = 0; j < JW; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < JW; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < JW; ++j) asm volatile("" ::"v"(b[j]), "v"(b1[j]));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a0[i]), "v"(a1[i]));
            }
            ring = (ring + 1) & (kS2WRing - 1);
            if (e & 4) hbuf = hbuf == 2 ? 0 : hbuf + 1;
            __builtin_amdgcn_s_barrier();
            if (s == 0 && pending_combine) {       // the previous tile's partials were parked before this barrier
                stat_combine(prev_tm, prev_tc);
                pending_combine = false;
            }
        }
        epilogue();
        prev_tm = tm;
        prev_tc = tc;
        pending_combine = has_stat;
    }
    if (has_stat) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        stat_combine(prev_tm, prev_tc);
    }
}

// Loader wave l (0..3).  Per step: its 4 pieces of the weight tile of step s + 2 (ring of 3) and its share of the halo of the
// chunk TWO chunks ahead (ring of 3 halo buffers; a chunk's 8 pieces per loader are spread over the steps of the chunk being
// computed, so that one-step chunks do not wait for a halo requested a step earlier); then everything issued before this step
// has landed (LDS-DMA pieces retire in order) and the step's barrier is met.
// Three cursors walk the block's tiles: the step being computed, the step two ahead (weights), the chunk two ahead (halo).
// Everything per tile / per chunk is decoded when a cursor enters it; a piece costs ~10 vector instructions.
struct S2TileCur {
    int it, tm, tc, type, ns, wrow0;
};

__device__ __forceinline__ void s2_cur_load(const char* smem, S2TileCur& t) {
    t.type = s2_col_field(smem, t.tc, 0);
    t.wrow0 = s2_col_field(smem, t.tc, 3);
    t.ns = s2_nsteps(smem, t.type);
}
__device__ __forceinline__ void s2_cur_init(const S2Params& p, const char* smem, S2TileCur& t, int tile_first) {
    t.it = 0;
    t.tm = __builtin_amdgcn_readfirstlane(tile_first / p.ncol);
    t.tc = __builtin_amdgcn_readfirstlane(tile_first - t.tm * p.ncol);
    s2_cur_load(smem, t);
}
__device__ __forceinline__ void s2_cur_next(const S2Params& p, const char* smem, S2TileCur& t, int tile_count) {
    ++t.it;
    if (++t.tc == p.ncol) {
        t.tc = 0;
        ++t.tm;
    }
    if (t.it < tile_count) s2_cur_load(smem, t);
}

__device__ __forceinline__ void s2_wait_vmcnt(int n) {   // wave-uniform; a smaller count than asked for only waits longer
    if (n >= 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if (n >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if (n >= 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Loader wave l (0..3).  Step t: its 4 pieces of the weight tile of step t + 3 (ring of 4), then — at the first step of a
// chunk — ALL its pieces of the halo of the chunk two chunks ahead (ring of 3: that buffer was released by the barrier that
// ended the previous chunk).  LDS-DMA pieces retire in issue order, so "step t + 1's weight tile and, if it opens a chunk,
// that chunk's halo have landed" is a count: the wave waits until no more pieces are outstanding than it issued AFTER the
// last one it needs — the halo requested this step (and usually the previous one) stays in flight across the barrier.
// Round 5 v1 waited for everything older than the current step: with operands streaming from HBM every step then paid the
// part of the memory latency that exceeds one step (2.3 us per step on layer4.0).
__device__ __forceinline__ void s2_loader(const S2Params& p, char* smem, int tile_first, int tile_count, int l) {
    const int lane = threadIdx.x & 63;
    const int Wo = p.Wo, Cs = p.Cs;
    const bool dg = p.mode == 1;
    const int nslots = kS2BM + Wo + 1;
    const int nrows = (nslots + 15) >> 4;
    const int lead = dg ? 0 : Wo + 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int* prog = (const int*)(smem + kS2OffProg);
    const long src_bytes = dg ? (long)p.M2 * Cs * 2 : (long)p.M2 * 4 * Cs * 2;
    const i32x4_t rs_src0 = s2_rsrc(p.src[0], src_bytes);
    const i32x4_t rs_src1 = s2_rsrc(p.src[1] ? p.src[1] : p.src[0], src_bytes);
    // (weight buffers: rows x klen; rows = the destination's channels)
    const i32x4_t rs_wt0 = s2_rsrc(p.wt[0], (long)p.ld[0] * p.klen[0] * 2);
    const i32x4_t rs_wt1 = s2_rsrc(p.wt[1] ? p.wt[1] : p.wt[0], (long)(dg ? p.ld[0] : p.ld[1]) * p.klen[1] * 2);
    const int swz = (((lane & 3) ^ lh_key(lane >> 4)) << 3);       // element offset of this lane's 16-byte chunk (source side)
    const unsigned wvoff0 = (unsigned)(((lane >> 2) * p.klen[0] + swz) * 2);
    const unsigned wvoff1 = (unsigned)(((lane >> 2) * p.klen[1] + swz) * 2);
    const int hplane = l & 1, hr0 = l >> 1;
    const int klen0 = p.klen[0], klen1 = p.klen[1];
    const int dbg = S2_DBG ? p.dbg : 0;

    auto word = [&](const S2TileCur& t, int s) -> int {
        if (S2_PROG_KARG) return __builtin_amdgcn_readfirstlane(p.prog[t.type][s]);
        return __builtin_amdgcn_readfirstlane(prog[t.type * kS2MaxSteps + s]);
    };
    int issued = 0;                               // pieces this wave has requested so far
    // this loader's four pieces (rows 16 l .. +15 of both 64-row halves, both 32-channel halves) of a step's weight tile
    auto stage_w = [&](const S2TileCur& t, int s, int ring) {
        if (t.it >= tile_count || (dbg & 2)) return;
        const int e = word(t, s);
        const int wsel = (e >> 14) & 1, c = (e >> 10) & 15;
        const int klen = wsel ? klen1 : klen0;
        const unsigned wv = wsel ? wvoff1 : wvoff0;
        const i32x4_t rs = wsel ? rs_wt1 : rs_wt0;
        const unsigned ldsw = lds0 + kS2OffW + ring * kS2WTile + l * 1024;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int f = e >> (15 + 6 * h);
            const int tap = f & 15, zero = (f >> 4) & 1, rel = (f >> 5) & 1;
            const unsigned soff = (unsigned)(((t.wrow0 + rel * 64 + 16 * l) * klen + tap * Cs + c * 64) * 2);
            const unsigned voff = zero ? kS2Oob : wv;
            s2_dma(voff, rs, soff, ldsw + h * 4096);
            s2_dma(voff, rs, soff + 64, ldsw + h * 4096 + kS2Half);
        }
        issued += 4;
    };
    // this loader's pieces k = 0..6 (32-channel plane hplane, slots 16 (hr0 + 2k) .. +15) of the halo of the chunk at (t, s)
    auto stage_h = [&](const S2TileCur& t, int s, int buf, bool prologue) {
        const bool live = t.it < tile_count && !(dbg & 1 && !prologue);
        if (!live && !prologue) return;
        int src = 0, soff = 0, q0 = 0, planeoff = 0;
        if (live) {
            const int e = word(t, s);
            src = (e >> 7) & 1;
            const int plane = (e >> 8) & 3, c = (e >> 10) & 15;
            planeoff = (plane >> 1) * 2 * Wo + (plane & 1);
            soff = c * 128 + hplane * 64;
            q0 = t.tm * kS2BM - lead + (lane >> 2);
        }
        const i32x4_t rs = src ? rs_src1 : rs_src0;
#pragma unroll
        for (int k = 0; k < kS2Rows / 2; ++k) {
            const int row = hr0 + 2 * k;
            if (row >= nrows && !prologue) continue;     // rows behind the last live slot: zeroed once, never written again
            unsigned voff = kS2Oob;
            const int i = (lane >> 2) + 16 * row;
            const int qq = q0 + 16 * row;
            if (live && i < nslots && qq >= 0 && qq < p.M2) {
                unsigned pix = (unsigned)qq;
                if (!dg && !(dbg & 64)) pix = 2u * pix + 2u * __umulhi(pix, p.magicWo) * (unsigned)Wo + (unsigned)planeoff;
                voff = (pix * (unsigned)Cs + (unsigned)swz) * 2u;
            }
            s2_dma(voff, rs, (unsigned)soff, lds0 + buf * kS2Halo + hplane * kS2Plane + row * 1024);
            ++issued;
        }
    };

    S2TileCur ht, wt_, ct;
    s2_cur_init(p, smem, ht, tile_first);
    wt_ = ht;
    ct = ht;
    int hs = 0, ws = 0, cs = 0;
    auto chunk_next = [&]() {                     // ht / hs at the first step of a chunk -> the next chunk
        if (ht.it >= tile_count) return;
        hs += (word(ht, hs) >> 4) & 7;
        if (hs >= ht.ns) {
            hs = 0;
            s2_cur_next(p, smem, ht, tile_count);
        }
    };
    auto wstep_next = [&]() {
        if (wt_.it >= tile_count) return;
        if (++ws == wt_.ns) {
            ws = 0;
            s2_cur_next(p, smem, wt_, tile_count);
        }
    };
    // ---- prologue: halos of chunks 0 and 1, the third buffer's zero rows, weight tiles of steps 0, 1, 2 ----
    stage_h(ht, hs, 0, true);
    chunk_next();
    stage_h(ht, hs, 1, true);
    chunk_next();
    {
        S2TileCur none = ht;
        none.it = tile_count;
        stage_h(none, 0, 2, true);
    }
    for (int r = 0; r < 3; ++r) {
        stage_w(wt_, ws, r);
        wstep_next();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // issue counts after the last piece of: the weight tile in ring slot r, the halo in buffer b
    int wq0 = issued, wq1 = issued, wq2 = issued, wq3 = issued;
    int hq0 = issued, hq1 = issued, hq2 = issued;
    int ring3 = 3, hb = 2, cbuf = 0, cring = 0;
    while (ct.it < tile_count) {
        const int e = word(ct, cs);
        const int issued_before = issued;
        stage_w(wt_, ws, ring3);
        wstep_next();
        if (ring3 == 0) wq0 = issued; else if (ring3 == 1) wq1 = issued; else if (ring3 == 2) wq2 = issued; else wq3 = issued;
        ring3 = (ring3 + 1) & 3;
        if (e & 8) {              // a chunk opens: the buffer of the chunk before it is free for the chunk two ahead
            stage_h(ht, hs, hb, false);
            chunk_next();
            if (hb == 0) hq0 = issued; else if (hb == 1) hq1 = issued; else hq2 = issued;
            hb = hb == 2 ? 0 : hb + 1;
        }
        // what step t + 1 reads: weight ring slot cring + 1 and, when this step closes its chunk, halo buffer cbuf + 1
        const int nr = (cring + 1) & 3;
        int need = nr == 0 ? wq0 : (nr == 1 ? wq1 : (nr == 2 ? wq2 : wq3));
        if (e & 4) {
            const int nb_ = cbuf == 2 ? 0 : cbuf + 1;
            const int hn = nb_ == 0 ? hq0 : (nb_ == 1 ? hq1 : hq2);
            need = hn > need ? hn : need;
            cbuf = nb_;
        }
        cring = nr;
        if (S2_WAIT_SIMPLE) need = issued_before;
        s2_wait_vmcnt(issued - need);
        __builtin_amdgcn_s_barrier();
        if (++cs == ct.ns) {
            cs = 0;
            s2_cur_next(p, smem, ct, tile_count);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.stat[0] != nullptr && (!dg || p.bnb_y != nullptr)) __builtin_amdgcn_s_barrier();
}

__global__ __launch_bounds__(768) void conv_s2lh_kernel(S2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, nb);
    const int total = p.ntm * p.gcost;
    const int first = s2_first_tile(p, (int)(((long)lb * total) / nb));
    int last = s2_first_tile(p, (int)(((long)(lb + 1) * total) / nb));
    if (lb == nb - 1 || last > p.ntiles) last = p.ntiles;
    const int count = last - first;
    if (count <= 0) return;
    {   // the step programs: kernel arguments -> LDS (every wave reads them every step)
        int* dstp = (int*)(smem + kS2OffProg);
        const int* srcp = &p.prog[0][0];
        for (int i = threadIdx.x; i < kS2MaxTypes * kS2MaxSteps; i += 768) dstp[i] = srcp[i];
        int* dcol = (int*)(smem + kS2OffCol);
        const int* scol = (const int*)&p.col[0];
        for (int i = threadIdx.x; i < kS2MaxCols * 8; i += 768) dcol[i] = scol[i];
        if (threadIdx.x < kS2MaxTypes) dcol[kS2MaxCols * 8 + threadIdx.x] = p.nsteps[threadIdx.x];
    }
    __syncthreads();
    if (wave >= 8) {
        s2_loader(p, smem, first, count, wave - 8);
        return;
    }
    const int wm = wave >> 1;
    s2_run<3>(p, smem, first, count, 3 * wm);     // (one copy of the matrix-wave code: the instruction cache is shared)
}

static int s2_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static int s2_launch(S2Params& p, hipStream_t st) {
    p.ntiles = p.ntm * p.ncol;
    p.dbg = PRIMIA_OPT(s2lh_dbg);
    p.gcost = 0;
    for (int c = 0; c < p.ncol; ++c) {
        p.prefix[c] = p.gcost;
        p.gcost += p.col[c].cost;
    }
    p.prefix[p.ncol] = p.gcost;
    p.magicWo = (unsigned)(((1ULL << 32) + p.Wo - 1) / p.Wo);
    p.magicHo = (unsigned)(((1ULL << 32) + p.Ho - 1) / p.Ho);
    const int ncu = s2_num_cus();
    // a block needs at least one tile; with fewer row groups than CUs the cost split still spreads the columns
    int grid = p.ntiles < ncu ? p.ntiles : ncu;
    const int opt = PRIMIA_OPT(s2lh_blocks);
    if (opt > 0 && opt < grid) grid = opt;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv_s2lh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kS2Lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    conv_s2lh_kernel<<<grid, 768, kS2Lds, st>>>(p);
    return launch_status();
}

// shapes: even H, W; W / 2 <= 28 (one halo buffer holds 192 + Wo + 1 slots); C a multiple of 64, K of 128; 32-bit offsets
bool conv_s2lh_ok(int N, int H, int W, int C, int K) {
    if (N <= 0 || H < 4 || W < 4 || (H & 1) || (W & 1) || W / 2 > 28) return false;   // (a 1-wide grid has no 32-bit magic divisor)
    if (C % 64 || K % 128 || C > 1024 || K > 1024) return false;
    const long M = (long)N * H * W;
    if (M * C >= (1L << 30) || (M / 4) * K >= (1L << 30)) return false;
    if (9 * (C / 64) > kS2MaxSteps || 4 * (K / 64) > kS2MaxSteps) return false;
    if (K / 128 * 2 > kS2MaxCols) return false;
    // data gradient column tiles: 2 (64-channel dx) or 4 * C / 128
    if (C != 64 && (C % 128 || 4 * (C / 128) > kS2MaxCols)) return false;
    return true;
}

int conv_s2lh_tiles_m(int N, int H, int W) { return (int)(((long)N * (H / 2) * (W / 2) + kS2BM - 1) / kS2BM); }

// Forward: y = conv3x3/2(x, w) (w may be null) and y_ds = conv1x1/2(x, w_ds) (may be null); x [N, H, W, C], outputs
// [N, H/2, W/2, K]; w [K][3][3][C], w_ds [K][C]; stat / stat_ds [tiles_m][2][K] or both null.
int conv_s2lh_fwd(const bf16* x, const bf16* w, bf16* y, float* stat, const bf16* w_ds, bf16* y_ds, float* stat_ds, int N,
                  int H, int W, int C, int K, hipStream_t st) {
    if (!conv_s2lh_ok(N, H, W, C, K)) return PRIMIA_ERR_UNSUPPORTED;
    if (!w && !w_ds) return PRIMIA_ERR_ARG;
    if (w && w_ds && ((stat == nullptr) != (stat_ds == nullptr))) return PRIMIA_ERR_ARG;
    S2Params p = {};
    p.mode = 0;
    p.src[0] = x; p.src[1] = nullptr;
    p.wt[0] = w ? w : w_ds; p.wt[1] = w_ds ? w_ds : w;
    p.klen[0] = w ? 9 * C : C; p.klen[1] = w_ds ? C : 9 * C;
    p.dst[0] = w ? y : y_ds; p.dst[1] = w_ds ? y_ds : y;
    p.stat[0] = w ? stat : stat_ds; p.stat[1] = w_ds ? stat_ds : stat;
    p.ld[0] = p.ld[1] = K;
    p.Cs = C;
    p.Ho = H / 2; p.Wo = W / 2;
    p.M2 = N * p.Ho * p.Wo;
    p.split = 0;
    p.ntm = conv_s2lh_tiles_m(N, H, W);
    const int nsl = C / 64;
    // type 0: conv1 — per 64-channel slice the planes (odd, odd), (even, even), (odd, even), (even, odd): the one-step chunk sits
    // between longer ones, so that the halo two chunks ahead always has steps to spread over
    int n = 0;
    for (int c = 0; c < nsl; ++c) {
        const struct { int plane, nk, sh[4], tap[4]; } ch[4] = {
            {3, 4, {3, 2, 1, 0}, {0, 2, 6, 8}}, {0, 1, {0}, {4}}, {2, 2, {2, 0}, {1, 7}}, {1, 2, {1, 0}, {3, 5}}};
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < ch[k].nk; ++i)
                p.prog[0][n++] = s2_step_word(ch[k].sh[i], i == ch[k].nk - 1, i == 0, ch[k].nk, 0, ch[k].plane, c, 0,
                                              ch[k].tap[i], 0, 0, ch[k].tap[i], 0, 1);
    }
    p.nsteps[0] = n;
    // type 1: downsample — plane (even, even), centre
    for (int c = 0; c < nsl; ++c) p.prog[1][c] = s2_step_word(0, 1, 1, 1, 0, 0, c, w ? 1 : 0, 0, 0, 0, 0, 0, 1);
    p.nsteps[1] = nsl;
    p.ncol = 0;
    for (int j = 0; j < K / 128; ++j) {
        if (w) p.col[p.ncol++] = S2Col{0, 0, 128 * j, 128 * j, 0, 0, 9 * nsl, 0};
        if (w_ds) p.col[p.ncol++] = S2Col{1, w ? 1 : 0, 128 * j, 128 * j, 0, 0, nsl, 0};
    }
    if (!w)   // downsample alone: its program reads weight array 0
        for (int c = 0; c < nsl; ++c) p.prog[1][c] = s2_step_word(0, 1, 1, 1, 0, 0, c, 0, 0, 0, 0, 0, 0, 1);
    return s2_launch(p, st);
}

// Data gradient: dx [N, H, W, C] = conv3x3/2^T(dy, wd) (+ conv1x1/2^T(dy_ds, wd_ds) when dy_ds is given); dy, dy_ds
// [N, H/2, W/2, K]; wd [C][3][3][K], wd_ds [C][K].  Every element of dx is written.
int conv_s2lh_dgrad(const bf16* dy, const bf16* wd, const bf16* dy_ds, const bf16* wd_ds, bf16* dx, int N, int H, int W, int C,
                    int K, hipStream_t st, const S2BnBwd* bnb) {
    if (!conv_s2lh_ok(N, H, W, C, K)) return PRIMIA_ERR_UNSUPPORTED;
    if ((dy_ds == nullptr) != (wd_ds == nullptr)) return PRIMIA_ERR_ARG;
    if (bnb && bnb->y && (C != 64 || !bnb->mask || !bnb->mean || !bnb->invstd || !bnb->sums)) return PRIMIA_ERR_UNSUPPORTED;
    S2Params p = {};
    p.mode = 1;
    p.src[0] = dy; p.src[1] = dy_ds;
    p.wt[0] = wd; p.wt[1] = wd_ds;
    p.klen[0] = 9 * K; p.klen[1] = K;
    p.dst[0] = p.dst[1] = dx;
    p.ld[0] = p.ld[1] = C;
    p.Cs = K;
    p.Ho = H / 2; p.Wo = W / 2;
    p.M2 = N * p.Ho * p.Wo;
    p.split = C == 64 ? 1 : 0;
    p.ntm = conv_s2lh_tiles_m(N, H, W);
    if (bnb && bnb->y) {
        p.bnb_y = bnb->y; p.bnb_mask = bnb->mask; p.bnb_mean = bnb->mean; p.bnb_invstd = bnb->invstd;
        p.stat[0] = bnb->sums;
    }
    const int nsl = K / 64;
    const bool ds = dy_ds != nullptr;
    auto tap = [](int r, int s) { return r * 3 + s; };
    if (p.split) {
        // type 0: classes (even, even) | (even, odd): taps (1,1) | (1,2) at shift 0, - | (1,0) one column on; + the downsample
        int n = 0;
        for (int c = 0; c < nsl; ++c) {
            p.prog[0][n++] = s2_step_word(0, 0, 1, 2, 0, 0, c, 0, tap(1, 1), 0, 0, tap(1, 2), 0, 0);
            p.prog[0][n++] = s2_step_word(1, 1, 0, 2, 0, 0, c, 0, 0, 1, 0, tap(1, 0), 0, 0);
        }
        if (ds)
            for (int c = 0; c < nsl; ++c) p.prog[0][n++] = s2_step_word(0, 1, 1, 1, 1, 0, c, 1, 0, 0, 0, 0, 1, 0);
        p.nsteps[0] = n;
        // type 1: classes (odd, even) | (odd, odd): (2,1) | (2,2) at 0, (0,1) | (0,2) one row on, - | (2,0), - | (0,0)
        int m = 0;
        for (int c = 0; c < nsl; ++c) {
            p.prog[1][m++] = s2_step_word(0, 0, 1, 4, 0, 0, c, 0, tap(2, 1), 0, 0, tap(2, 2), 0, 0);
            p.prog[1][m++] = s2_step_word(2, 0, 0, 4, 0, 0, c, 0, tap(0, 1), 0, 0, tap(0, 2), 0, 0);
            p.prog[1][m++] = s2_step_word(1, 0, 0, 4, 0, 0, c, 0, 0, 1, 0, tap(2, 0), 0, 0);
            p.prog[1][m++] = s2_step_word(3, 1, 0, 4, 0, 0, c, 0, 0, 1, 0, tap(0, 0), 0, 0);
        }
        p.nsteps[1] = m;
        p.ncol = 2;
        p.col[0] = S2Col{1, 0, 0, 0, 2, 3, m, 0};
        p.col[1] = S2Col{0, 0, 0, 0, 0, 1, n, 0};
    } else {
        // one type per class: 0 (even, even) [+ downsample], 1 (even, odd), 2 (odd, even), 3 (odd, odd)
        const struct { int nk, sh[4], tp[4]; } cl[4] = {{1, {0}, {tap(1, 1)}},
                                                        {2, {0, 1}, {tap(1, 2), tap(1, 0)}},
                                                        {2, {0, 2}, {tap(2, 1), tap(0, 1)}},
                                                        {4, {0, 1, 2, 3}, {tap(2, 2), tap(2, 0), tap(0, 2), tap(0, 0)}}};
        for (int t = 0; t < 4; ++t) {
            int n = 0;
            for (int c = 0; c < nsl; ++c)
                for (int i = 0; i < cl[t].nk; ++i)
                    p.prog[t][n++] = s2_step_word(cl[t].sh[i], i == cl[t].nk - 1, i == 0, cl[t].nk, 0, 0, c, 0, cl[t].tp[i], 0, 0,
                                                  cl[t].tp[i], 0, 1);
            if (t == 0 && ds)
                for (int c = 0; c < nsl; ++c) p.prog[0][n++] = s2_step_word(0, 1, 1, 1, 1, 0, c, 1, 0, 0, 0, 0, 0, 1);
            p.nsteps[t] = n;
        }
        // column order: the expensive class first, then the three cheap ones — per channel tile
        p.ncol = 0;
        const int order[4] = {3, 0, 1, 2};
        for (int j = 0; j < C / 128; ++j)
            for (int o = 0; o < 4; ++o) {
                const int t = order[o];
                p.col[p.ncol++] = S2Col{t, 0, 128 * j, 128 * j, t, t, p.nsteps[t], 0};
            }
    }
    return s2_launch(p, st);
}

}  // namespace primia
