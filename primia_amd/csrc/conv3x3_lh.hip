// 3x3 / stride-1 / pad-1 convolution, forward and data gradient, for the wide stages (ResNet-18 layer2-4:
// 128..512 channels on 28x28 .. 7x7 images), bf16, gfx950 — "linear halo" implicit GEMM.
//
// The generic implicit GEMM (conv_igemm.hip) stages a 128-pixel activation tile AND a weight tile for every one of
// the 9 taps of every 64-channel chunk: 32 KiB of L2 -> LDS traffic per 2 MFLOP-pairs, and that fill rate (not the
// MFMA or LDS pipes) is what bounds it.  Here a block owns BM = 224 (or 256) CONSECUTIVE output pixels (NHWC order, no
// spatial patch, so 28-, 14- and 7-wide images tile without waste) x 128 output channels, and per 64-channel chunk stages
//     the "linear halo":  pixels m0 - (W+1) .. m0 + BM - 1 + (W+1)  (<= 320 slots of 128 B)   ONCE for all 9 taps
//     one weight tile per tap: 128 rows x 128 B = 16 KiB
// Tap (r, s) of output pixel m reads source pixel m + (r-1)*W + (s-1): the same LDS buffer at a tap-uniform slot
// shift.  What the shifted slot holds when the tap falls outside the image (left / right border, first / last row,
// neighbouring image) is some other pixel, so validity is a per-pixel 9-bit mask and an invalid tap reads an all-zero
// slot instead (the 36 read offsets per lane are computed once).  L2 -> LDS bytes per chunk: 40 + 9 x 16 =
// 184 KiB for 4x the MFMA work of an implicit-GEMM tile step sequence that moves 9 x 32 = 288 KiB for 1x... i.e.
// 0.32x the bytes per flop.
//
// 8 waves = 4 (pixels: 64 each; 64, 64, 48, 48 in the 224-pixel tile) x 2 (channels: 64 each); per wave and 32-channel
// half 4 weight + 4 (3) pixel fragment reads (ds_read_b128) feed 16 (12) MFMAs.  Weights are the MFMA A operand (a lane ends up with 4 consecutive output
// channels of one pixel).  LDS: 2 halo buffers (2 x 40 KiB) + a 4-deep weight ring (4 x 16 KiB) = 144 KiB, one block
// per CU.  Ping-pong wave halves, two barriers per tap step; LDS-DMA issued as inline asm and ordered with counted vmcnt (see
// conv_wgrad_patch.hip).  Both LDS images carry the XOR chunk swizzle (chunk ^ ((row >> 1) & 7)) applied on the DMA
// source side; weight reads are conflict-free, shifted halo reads are conflict-free for half of the shifts and 2-way
// on part of the lanes otherwise (the LDS pipe is far from saturated here).
//
// Data gradient = the same kernel on (dy, w_dgrad [C][R][S][K]) with the tap direction flipped.
#include <stdlib.h>

#include "conv3x3_lh.h"

namespace primia {

__device__ __attribute__((aligned(16))) const unsigned char kLhZeroPage[16] = {0};

__device__ __forceinline__ void lh_dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ void lh_wait_vmcnt(int n) {   // wave-uniform n
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct LhParams {
    const bf16* src;   // [M][Cs]
    const bf16* wt;    // [Nd][9][Cs]
    bf16* dst;         // [M][Nd]
    int H, W, Cs, Nd;
    long M;            // N*H*W
    int flip;          // 0 forward, 1 data gradient
    int accumulate;    // dst += result
    const uint8_t* acc_mask;   // accumulate form: ReLU mask bits applied to the OLD values (one byte per 8 channels)
    int ntile_n;
    // data gradient only: backward sums of the BatchNorm that consumes dst (see primia_conv2d_dgrad_bnsums):
    // bwd_partials [tiles_m][2][Nd] = per-tile (sum g, sum g * (y - mean)), g = dst AS STORED where the ReLU passed
    const bf16* bn_y;
    const uint8_t* bn_mask;      // one byte per 8 channels (bit i = z_i > 0), or null: mask = fma(y-mean, invstd*gamma, beta) > 0
    const float* bn_gamma;
    const float* bn_beta;
    const float* bn_mean;
    const float* bn_invstd;
    float* bwd_partials;
    float* stat_partials;  // forward only: BatchNorm partial sums [tiles_m][2][Nd] of the values AS STORED (or null)
    int debug;         // timing experiments only (PRIMIA_LH_DEBUG): 1 no stores, 2 no staging, 4 no MFMA
};

constexpr int kLhBN = 128;
template <int J>
struct LhJ {
    static constexpr int value = J;
};
// pixels per tile: 224 (default) or 256 (PRIMIA_LH_BM=256)
static int lh_bm() {
    static const int bm = getenv("PRIMIA_LH_BM") && atoi(getenv("PRIMIA_LH_BM")) == 256 ? 256 : 224;
    return bm;
}
constexpr int kLhSlots = 320;                  // BM + 2*(W+1) <= 256 + 62, W <= 30
constexpr int kLhHalo = kLhSlots * 128;        // bytes per halo buffer
constexpr int kLhWt = kLhBN * 128;             // bytes per weight tile
constexpr int kLhWR = 4;                       // weight ring depth
constexpr int kLhLds = 2 * kLhHalo + kLhWR * kLhWt;
constexpr int kLhZeroSlot = kLhSlots - 1;      // never a live slot (W <= 30): staged as zeros in both buffers

// JB = pixel fragments of the "B" waves (4-7).  JB = 4: 256-pixel tiles.  JB = 3: 224-pixel tiles (waves 0-3 own
// 2 x 64 pixels, waves 4-7 2 x 48): 28x28, 14x14 and 7x7 images at batch 256 then give 896 / 448 / 224 tiles = 3.5 / 1.75 /
// 0.875 rounds of 256 CUs where 256-pixel tiles give 784 / 392 / 196 = 4 / 2 / 1 rounds of which the last is 77 % empty;
// on every SIMD an A segment (32 MFMAs) alternates with a B segment (24), so a step costs 7/8 of the 256-pixel step.
template <bool ACC, int JB>
__global__ __launch_bounds__(512) void conv3x3_lh_kernel(LhParams p) {
    constexpr int BM = 128 + 32 * JB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = tile % p.ntile_n, tm = tile / p.ntile_n;
    const long m0 = (long)tm * BM;
    const int n0 = tn * kLhBN;
    const int W = p.W, H = p.H, Cs = p.Cs;
    const long hm0 = m0 - (W + 1);             // source pixel of halo slot 0
    const int nslots = BM + 2 * W + 2;         // live slots
    const int nchunks = Cs >> 6;
    const int klen = 9 * Cs;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- staging ----------------------------------------------------------------------------------
    // halo instruction g = 8*q + wave (q = 0..4) of chunk c: 8 slots x 128 B; this lane: slot 64q + tid/8, LDS piece
    // lane%8.  The swizzled source chunk does not depend on q (64q/2 is a multiple of 8); validity is 5 bits.
    const int hslot0 = tid >> 3;
    const bf16* hptr = p.src + (hm0 + hslot0) * Cs + (((lane & 7) ^ ((hslot0 >> 1) & 7)) * 8);
    unsigned hok = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int slot = 64 * q + hslot0;
        const long pix = hm0 + slot;
        if (slot < nslots && pix >= 0 && pix < p.M) hok |= 1u << q;
    }
    auto halo_dma = [&](int c, int q) {
        const bf16* gp = ((hok >> q) & 1u) ? hptr + ((long)q * 64 * Cs + c * 64) : (const bf16*)kLhZeroPage;
        lh_dma16(gp, __builtin_amdgcn_readfirstlane(lds0 + (c & 1) * kLhHalo + (8 * q + wave) * 1024));
    };
    // weight instructions 2*wave, 2*wave+1 of step t (tap, chunk): 8 rows x 128 B each
    const bf16* wrow[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = (2 * wave + k) * 8 + (lane >> 3);
        wrow[k] = p.wt + (long)(n0 + row) * klen + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
    }
    auto wt_dma = [&](int tap, int c, int buf) {
        const int off = tap * Cs + c * 64;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            lh_dma16(wrow[k] + off, __builtin_amdgcn_readfirstlane(lds0 + 2 * kLhHalo + buf * kLhWt + (2 * wave + k) * 1024));
    };

    // ---- per-lane constants --------------------------------------------------------------------------
    // pixel fragments j: tile pixel wm*64 + 16j + fr; its halo slot at shift 0 is that + (W+1); 9-bit tap validity
    int sj[4];
    unsigned pmask[4];
    const int wpix0 = wave < 4 ? wm * 64 : 128 + (wm - 2) * 16 * JB;   // first tile pixel of this wave
    const int nj = wave < 4 ? 4 : JB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pl = wpix0 + 16 * j + fr;
        sj[j] = pl + W + 1;
        const long m = m0 + pl;
        unsigned mask = 0;
        if (m < p.M && j < nj) {
            const int w = (int)(m % W);
            const int h = (int)((m / W) % H);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int r = t / 3, s = t - 3 * r;
                const int dr = p.flip ? 1 - r : r - 1, ds = p.flip ? 1 - s : s - 1;
                if ((unsigned)(h + dr) < (unsigned)H && (unsigned)(w + ds) < (unsigned)W) mask |= 1u << t;
            }
        }
        pmask[j] = mask;
    }
    // halo read offsets of the 9 taps (32-channel half 1: ^ 64); a tap outside the image reads the all-zero slot
    int boffT[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int r = t / 3, s = t - 3 * r;
        const int shift = p.flip ? (1 - r) * W + (1 - s) : (r - 1) * W + (s - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = ((pmask[j] >> t) & 1u) ? sj[j] + shift : kLhZeroSlot;
            boffT[t][j] = slot * 128 + ((fg ^ ((slot >> 1) & 7)) << 4);
        }
    }
    // weight fragments i: row wn*64 + 16i + fr, 16-byte chunk fg of the 32-channel half (half 1: offset ^ 64)
    int aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wn * 64 + 16 * i + fr;
        aoff[i] = row * 128 + ((fg ^ ((row >> 1) & 7)) << 4);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: ping-pong halves ----------------------------------------------------------------------
    // Waves w and w + 4 share a SIMD.  Waves 0-3 ("A") read the fragments of step t (and issue their DMA pieces) in
    // segment 2t and run its 32 MFMAs in segment 2t+1; waves 4-7 ("B") do the same one segment later.  Barrier B_g
    // stands in front of segment g, so on every SIMD a matrix segment always runs beside a memory segment
    // (tools/micro/mfma_stage.hip: 1.48 -> 1.80 PFLOP/s for such a body).  A load segment ends with lgkmcnt(0): the
    // buffers a step leaves are free for DMA two barriers later.  Per step a wave issues, in this order, at most one
    // halo piece of the next chunk and the two weight pieces of step t+3; before every even barrier it waits until
    // only the pieces of the last two steps may still be in flight (its pieces of step t were issued three steps ago).
    const bool staging = !(p.debug & 2);
    if (staging) {
#pragma unroll
        for (int q = 0; q < 5; ++q) halo_dma(0, q);
        wt_dma(0, 0, 0);
        wt_dma(1, 0, 1);
        wt_dma(2, 0, 2);
    }
    int wb = 0, wb3 = 3;                      // weight ring slots of step t and step t + 3
    int prev1 = 2, prev2 = 2;

    bf16x8_t a[2][4], b[2][4];
    auto top = [&]() {
        lh_wait_vmcnt(staging ? prev1 + prev2 : 0);
        __builtin_amdgcn_s_barrier();
    };
    // `tap` is a compile-time constant after unrolling, `chunk` is not
    auto load_segment = [&](int chunk, int tap, auto jtag) {
        constexpr int J = decltype(jtag)::value;
        int issued = 0;
        if (staging) {
            if (tap < 5 && chunk + 1 < nchunks) {
                halo_dma(chunk + 1, tap);
                ++issued;
            }
            const int tap3 = tap + 3 >= 9 ? tap + 3 - 9 : tap + 3, chunk3 = tap + 3 >= 9 ? chunk + 1 : chunk;
            if (chunk3 < nchunks) {
                wt_dma(tap3, chunk3, wb3);
                issued += 2;
            }
        }
        prev2 = prev1;
        prev1 = issued;
        if (!(p.debug & 4)) {
            const char* hb = smem + (chunk & 1) * kLhHalo;
            const char* wbp = smem + 2 * kLhHalo + wb * kLhWt;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[kk][i] = *(const bf16x8_t*)(wbp + (aoff[i] ^ (kk << 6)));
#pragma unroll
                for (int j = 0; j < J; ++j) b[kk][j] = *(const bf16x8_t*)(hb + (boffT[tap][j] ^ (kk << 6)));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        wb = wb == kLhWR - 1 ? 0 : wb + 1;
        wb3 = wb3 == kLhWR - 1 ? 0 : wb3 + 1;
    };
    auto mfma_segment = [&](auto jtag) {
        constexpr int J = decltype(jtag)::value;
        if (p.debug & 4) return;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (wave < 4) {
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                top();                                  // B_2t
                load_segment(c, tap, LhJ<4>{});
                __builtin_amdgcn_s_barrier();           // B_2t+1
                mfma_segment(LhJ<4>{});
            }
        }
        __builtin_amdgcn_s_barrier();                   // B_2n
    } else {
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                top();                                  // B_2t
                if (tap != 0 || c != 0) mfma_segment(LhJ<JB>{}); // step t - 1
                __builtin_amdgcn_s_barrier();           // B_2t+1
                load_segment(c, tap, LhJ<JB>{});
            }
        }
        __builtin_amdgcn_s_barrier();                   // B_2n
        mfma_segment(LhJ<JB>{});
    }

    // ---- epilogue: results leave through LDS as whole 256-byte pixel rows ------------------------------------
    // (after B_2n nobody reads the staging buffers any more).  A lane holds 4 consecutive channels (16i + 4 fg ..) of
    // pixel 16j + fr per fragment; rows are staged as bf16 (256 B per pixel), or as fp32 (512 B) in the accumulate
    // form, which adds the old values at write-back with ONE rounding.  16-byte chunks are XOR-swizzled with the
    // pixel so that both the fragment writes and the row reads spread over the banks.
    if ((p.debug & 1) && acc[0][0][0] != 12345.f) return;
    constexpr int OPIX = ACC ? 512 : 256;
    // backward-sum form: the BatchNorm input (and mask bytes) of this thread's 8 write-back chunks are requested
    // NOW, so that their latency runs beside the LDS staging of the accumulators
    u32x4 yraw[8];
    unsigned mkb[8];
    if (p.bwd_partials) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = tid + 512 * k;
            const long m = m0 + (q >> 4);
            yraw[k] = u32x4{0, 0, 0, 0};
            mkb[k] = 0;
            if (m < p.M && (q >> 4) < BM) {
                const long eo = m * p.Nd + n0 + (q & 15) * 8;
                yraw[k] = *(const u32x4*)(p.bn_y + eo);
                if (p.bn_mask) mkb[k] = p.bn_mask[eo >> 3];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j >= nj) break;                      // wave-uniform: the B waves of a 224-pixel tile own three fragments
        const int px = wpix0 + 16 * j + fr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = wn * 64 + 16 * i + fg * 4;
            if constexpr (ACC) {
                *(f32x4*)(smem + px * OPIX + (((ch >> 2) ^ (px & 31)) << 4)) = acc[i][j];
            } else {
                u32x2 o;
                o[0] = (uint32_t)f32_to_bf16(acc[i][j][0]) | ((uint32_t)f32_to_bf16(acc[i][j][1]) << 16);
                o[1] = (uint32_t)f32_to_bf16(acc[i][j][2]) | ((uint32_t)f32_to_bf16(acc[i][j][3]) << 16);
                *(u32x2*)(smem + px * OPIX + (((ch >> 3) ^ (px & 15)) << 4) + ((ch & 4) << 1)) = o;
            }
        }
    }
    __syncthreads();
    float st1[8], st2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) st1[e] = st2[e] = 0.f;
    float bmean[8], bscale[8], bbeta[8];     // this thread's 8 channels are the same in every trip
    if (p.bwd_partials) {
        const int cb = n0 + (tid & 15) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bmean[e] = p.bn_mean[cb + e];
            bscale[e] = p.bn_invstd[cb + e] * p.bn_gamma[cb + e];   // the forward pass's operand order
            bbeta[e] = p.bn_beta ? p.bn_beta[cb + e] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int q = tid + 512 * k;            // 16-byte output chunk: pixel q / 16, channels 8 * (q % 16) ..
        const int px = q >> 4, c8 = q & 15;
        const long m = m0 + px;
        if (m >= p.M || px >= BM) continue;
        bf16* gq = p.dst + m * p.Nd + n0 + c8 * 8;
        u32x4 v;
        if constexpr (ACC) {
            u32x4 old = *(const u32x4*)gq;
            if (p.acc_mask) {
                const unsigned mk = p.acc_mask[(gq - p.dst) >> 3];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    old[e] &= ((mk >> (2 * e)) & 1u ? 0x0000ffffu : 0u) | ((mk >> (2 * e + 1)) & 1u ? 0xffff0000u : 0u);
            }
            const f32x4 lo = *(const f32x4*)(smem + px * OPIX + (((2 * c8) ^ (px & 31)) << 4));
            const f32x4 hi = *(const f32x4*)(smem + px * OPIX + (((2 * c8 + 1) ^ (px & 31)) << 4));
            float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] += __uint_as_float(old[e] << 16);
                f[2 * e + 1] += __uint_as_float(old[e] & 0xffff0000u);
                v[e] = (uint32_t)f32_to_bf16(f[2 * e]) | ((uint32_t)f32_to_bf16(f[2 * e + 1]) << 16);
            }
        } else {
            v = *(const u32x4*)(smem + px * OPIX + ((c8 ^ (px & 15)) << 4));
            if (p.stat_partials) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                    st1[2 * e] += lo;
                    st2[2 * e] += lo * lo;
                    st1[2 * e + 1] += hi;
                    st2[2 * e + 1] += hi * hi;
                }
            }
        }
        *(u32x4*)gq = v;
        if (p.bwd_partials) {
            float yv[8];
            Chunk<bf16>::unpack(yraw[k], yv);
            unsigned mk;
            if (p.bn_mask) {
                mk = mkb[k];
            } else {
                mk = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    mk |= (__builtin_fmaf(yv[e] - bmean[e], bscale[e], bbeta[e]) > 0.f ? 1u : 0u) << e;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float raw = (e & 1) ? __uint_as_float(v[e >> 1] & 0xffff0000u) : __uint_as_float(v[e >> 1] << 16);
                const float gv = (mk >> e) & 1u ? raw : 0.f;
                st1[e] += gv;
                st2[e] += gv * (yv[e] - bmean[e]);
            }
        }
    }
    if ((!ACC && p.stat_partials) || p.bwd_partials) {
        // BatchNorm batch statistics of the NEXT layer for free (as conv3x3_c64.hip): a write-back thread holds the
        // same 8 channels (tid % 16) of 8 pixels; the 32 threads of a channel group are folded through LDS in a
        // fixed order -> one deterministic partial per block, consumed by primia_bn_fwd_train_from_sums.
        __syncthreads();                       // the staged rows are dead
        float* red = (float*)smem;             // [32][2][128]
        const int grp = tid >> 4, c0 = (tid & 15) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(grp * 2 + 0) * 128 + c0 + e] = st1[e];
            red[(grp * 2 + 1) * 128 + c0 + e] = st2[e];
        }
        __syncthreads();
        if (tid < 256) {
            const int q = tid >> 7, c = tid & 127;
            float a = 0.f;
#pragma unroll 8
            for (int g = 0; g < 32; ++g) a += red[(g * 2 + q) * 128 + c];
            (p.bwd_partials ? p.bwd_partials : p.stat_partials)[((long)tm * 2 + q) * p.Nd + n0 + c] = a;
        }
    }
}

// returns PRIMIA_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the implicit GEMM)
// pixel tiles (= BatchNorm partial slots of the forward form) if the shape is served by this kernel, else
// PRIMIA_ERR_UNSUPPORTED
int conv3x3_lh_tiles_m(int N, int H, int W, int Cs, int Nd) {
    static const bool off = getenv("PRIMIA_LH") && getenv("PRIMIA_LH")[0] == '0';
    if (off || W > 30 || W < 2 || Cs % 64 || Nd % kLhBN) return PRIMIA_ERR_UNSUPPORTED;
    const long M = (long)N * H * W;
    if (M * (Cs > Nd ? Cs : Nd) >= (1L << 31)) return PRIMIA_ERR_UNSUPPORTED;
    return (int)((M + lh_bm() - 1) / lh_bm());
}

int conv3x3_lh_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                        int accumulate, hipStream_t st, float* stat_partials, const LhBnArgs* bn,
                        const uint8_t* acc_mask) {
    if (conv3x3_lh_tiles_m(N, H, W, Cs, Nd) < 0) return PRIMIA_ERR_UNSUPPORTED;
    if (stat_partials && (flip || accumulate)) return PRIMIA_ERR_ARG;
    if (bn && (!flip || stat_partials)) return PRIMIA_ERR_ARG;
    const long M = (long)N * H * W;
    LhParams p;
    p.src = src; p.wt = wt; p.dst = dst;
    p.H = H; p.W = W; p.Cs = Cs; p.Nd = Nd; p.M = M;
    p.flip = flip; p.accumulate = accumulate;
    p.acc_mask = accumulate ? acc_mask : nullptr;
    p.ntile_n = Nd / kLhBN;
    p.stat_partials = stat_partials;
    p.bwd_partials = nullptr;
    p.bn_y = nullptr; p.bn_mask = nullptr; p.bn_gamma = p.bn_beta = p.bn_mean = p.bn_invstd = nullptr;
    if (bn) {
        p.bn_y = (const bf16*)bn->y; p.bn_mask = bn->mask; p.bn_gamma = bn->gamma; p.bn_beta = bn->beta;
        p.bn_mean = bn->mean; p.bn_invstd = bn->invstd; p.bwd_partials = bn->partials;
    }
    static const int dbg = getenv("PRIMIA_LH_DEBUG") ? atoi(getenv("PRIMIA_LH_DEBUG")) : 0;
    p.debug = dbg;
    const int bm = lh_bm();
    const int grid = (int)((M + bm - 1) / bm) * p.ntile_n;
    void (*kern)(LhParams) = bm == 256 ? (accumulate ? conv3x3_lh_kernel<true, 4> : conv3x3_lh_kernel<false, 4>)
                                       : (accumulate ? conv3x3_lh_kernel<true, 3> : conv3x3_lh_kernel<false, 3>);
    static bool attr_set[2] = {false, false};
    if (!attr_set[accumulate ? 1 : 0]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLhLds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set[accumulate ? 1 : 0] = true;
    }
    kern<<<grid, 512, kLhLds, st>>>(p);
    return launch_status();
}

}  // namespace primia
