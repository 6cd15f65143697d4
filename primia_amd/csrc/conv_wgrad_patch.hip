// Weight gradient of the 3x3 / stride-1 convolutions as a halo-patch kernel (bf16, gfx950).
//
//   dw[k, (r,s,c)] += sum_m dy[m, k] * x[pix(m, r, s), c]
//
// The per-tap kernels of conv_wgrad.hip stage a dy tile and an x tile per (tap, pixel step): every activation and every
// output gradient travels L2 -> LDS nine times.  Here one 8-wave block owns a [64 out-chan] x [9 taps x 64 in-chan] slab
// of dw and walks 2-D sub-patches of the image; per sub-patch dy and x are staged ONCE (with a row / column halo) and
// the nine taps read the same LDS images at shifted pixel slots.
//
// This file holds the third generation of the scheme (conv_wgrad_patch33_kernel, "3 + 3 fragments").  Rounds 1 and 2
// built it on v_mfma_f32_16x16x32_bf16 (2 + 9 fragment reads for 18 MFMAs, 72 accumulators per wave) and then on
// v_mfma_f32_32x32x16_bf16 with the two sub-patches of a stage going to the two HALVES of the block (1 + 9 fragments per
// 9 MFMAs, 81 -> 71 us per layer); both were superseded and removed (profiles/r02_wgrad32_experiments.txt,
// profiles/r03_wgp33_phase_profile.txt keep their measurements).  What v3 inherits from them:
//   * LDS image: a pixel slot is 128 B = two 64-B channel groups; group g of slot s sits in half g ^ bit1(s), so the
//     four consecutive slots one 32-lane phase of ds_read_b64_tr_b16 touches always cover all 64 banks, for any slot
//     alignment (tap shifts!);
//   * staging = buffer-addressed LDS-DMA (out-of-image / dead lanes get an offset beyond num_records: hardware zeros);
//   * wave (half, kg, cg) owns out-channels 32*kg.., in-channels 32*cg.. of all nine taps for ITS half's sub-patch =
//     nine f32x16 accumulators; half 0 issues its pieces then multiplies, half 1 multiplies then issues; the halves
//     meet in LDS once after the loop; one partial slab per block -> ordered reduce (deterministic).
#include <stdlib.h>

#include "conv_wgrad.h"

namespace primia {

__device__ __attribute__((aligned(16))) const unsigned char kWpZeroPage[16] = {0};

// LDS-DMA issued as inline asm: the compiler's waitcnt pass does not see an LDS write, so it does not
// drain vmcnt before every LDS read of the compute phase (it does for the builtin); the kernel orders
// DMA and reads itself with counted s_waitcnt vmcnt + s_barrier.  M0 carries the wave-uniform LDS
// base of the 1-KiB destination; nothing else in this kernel uses M0.
__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ int key_lin(int slot) { return ((slot >> 1) & 1) | (((slot >> 3) & 1) << 1); }

template <int SW>
__device__ __forceinline__ int key_halo(int slot) {
    return SW == 8 ? ((slot >> 1) & 3) : key_lin(slot);
}

struct PatchParams {
    const bf16* x;
    const bf16* dy;
    float* dw;
    int H, W, C, K, klen;     // stride 1, pad 1: Ho == H, Wo == W
    int nct, nkt;
    int PH, PW, PPI;          // sub-patches per image column / row / image
    int total;                // sub-patches overall
    int per_block;            // sub-patches per block (even)
    long split_stride;
    double* sqnorm;           // per-sample norm pass (see WgradParams::sqnorm)
    int nsplit, split_fastest;
    float* ws;                // if set: every block stores its partial slab here (no atomics), see below
    int pairimg;              // v3, DP-SGD norm pass: a half owns whole images (units of t0 / per_block / total: images)
    int nimg;                 // batch size
    // v3, grouped launch: `ngroups` layers of ONE shape share the launch — blocks [g * group_blocks, (g + 1) * group_blocks)
    // work on layer g (x / dy of layers 1.. in xg / dyg; slabs of layer g behind those of layer g - 1)
    int ngroups, group_blocks;
    const bf16* xg[3];
    const bf16* dyg[3];
};

constexpr int kSlab = 64 * 9 * 64;   // accumulator values of one block

// compile-time experiment switches (tools/micro/wgp33_bench.hip): 1 no epilogue, 2 no DMA after the prologue, 4 no MFMA,
// 8 no fragment reads.  WGP33_PROF: per-wave cycles in DMA issue / vmcnt wait / barrier wait / compute / epilogue.
#ifndef WGP33_DBG
#define WGP33_DBG 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool V>
struct OrderTag {
    static constexpr bool value = V;
};

__device__ __forceinline__ int key32(int slot) { return (slot >> 1) & 1; }

// Epilogue shared by the stride-1 and stride-2 kernels: the two halves of the block meet in LDS (waves 4-7 park their
// accumulators, waves 0-3 add them), then one of the three outputs: per-sample squared norm, partial slab, atomics.
template <bool MERGE = true>
__device__ __forceinline__ void wgrad32_epilogue(f32x16 (&acc)[9], const PatchParams& p, char* smem, int wave, int lane,
                                                 int half, int kg, int cg, int kt, int ct, int split) {
    // ---- the two halves meet in LDS: waves 4-7 park their accumulators, waves 0-3 add them ---------------------------
    // (MERGE = false, the loader-wave form: four matrix waves own the whole slab, nothing to meet — and no block-wide
    // barrier here: the loader waves have left)
    if ((WGP33_DBG & 1) && acc[0][0] != 12345.f) return;     // compile-time experiment switch (tools/micro/wgp33_bench.hip)
    if constexpr (MERGE) {
    __syncthreads();                                   // every wave is done reading the stage ring
    {
        f32x4* park = (f32x4*)smem + ((wave & 3) * 36) * 64 + lane;     // [wave & 3][t * 4 + m][lane] chunks of 16 B
        if (half) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    park[(t * 4 + m) * 64] = f32x4{acc[t][4 * m], acc[t][4 * m + 1], acc[t][4 * m + 2], acc[t][4 * m + 3]};
        }
        __syncthreads();
        if (half) return;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 v = park[(t * 4 + m) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][4 * m + j] += v[j];
            }
    }
    }
    // lane holds out-chan rows 32*kg + 8*m + 4*(lane>>5) + j (register 4*m + j), in-chan column 32*cg + (lane & 31)
    if (p.sqnorm) {
        double sq = 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) sq += (double)acc[t][j] * (double)acc[t][j];
        wave_sqnorm_add(sq, p.sqnorm + split);
        if (!p.ws) return;      // (ws set: the sample's tile is KEPT for the clipped sum — one split per image)
    }
    if (p.ws) {
        // slab chunk ((t*4 + m)*4 + wave)*64 + lane = registers 4m..4m+3 of accumulator t (wgrad_patch32_reduce_kernel)
        float* o = p.ws + ((long)(kt * p.nct + ct) * p.nsplit + split) * kSlab + (wave * 64 + lane) * 4;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                *(f32x4*)(o + (t * 4 + m) * 1024) = f32x4{acc[t][4 * m], acc[t][4 * m + 1], acc[t][4 * m + 2], acc[t][4 * m + 3]};
        return;
    }
    float* out = p.dw + (long)split * p.split_stride;
    auto flush = [&](int t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kt * 64 + 32 * kg + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            const int e = t * p.C + ct * 64 + 32 * cg + (lane & 31);
            unsafeAtomicAdd(out + (long)k * p.klen + e, acc[t][r]);
        }
    };
    // blocks of one slab start at different taps, so that at any instant they hit different cache lines
#define PRIMIA_FLUSH32_FROM(R)                     \
    case R:                                        \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) flush((t + R) % 9); \
        break;
    switch (split % 9) {
        PRIMIA_FLUSH32_FROM(0)
        PRIMIA_FLUSH32_FROM(1)
        PRIMIA_FLUSH32_FROM(2)
        PRIMIA_FLUSH32_FROM(3)
        PRIMIA_FLUSH32_FROM(4)
        PRIMIA_FLUSH32_FROM(5)
        PRIMIA_FLUSH32_FROM(6)
        PRIMIA_FLUSH32_FROM(7)
        PRIMIA_FLUSH32_FROM(8)
    }
#undef PRIMIA_FLUSH32_FROM
}

// Ordered reduction of the v2 slabs (chunk q = ((t*4 + m)*4 + wave)*64 + lane, see the kernel's store).
struct ReduceGroup {        // grouped launch: dw of layers 1..3 and the combos of one layer (0: single layer)
    float* dwg[3];
    int combos;
    const float* wgt;       // DP-SGD clipped sum: split s is sample s, weighted by wgt[s] (null: plain sum)
};

template <int SL>
__device__ __forceinline__ void patch32_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int nct,
                                                    int C, int klen, ReduceGroup rg, int blk) {
    constexpr int CL = 256 / SL;
    constexpr int CPB = kSlab / 4 / CL;
    __shared__ f32x4 red[SL][CL];
    int combo = blk / CPB;
    if (rg.combos > 0) {        // slabs: [layer][combo][split]
        const int gi = combo / rg.combos;
        combo -= gi * rg.combos;
        ws += (long)gi * rg.combos * nsplit * kSlab;
        if (gi > 0) dw = rg.dwg[gi - 1];
    }
    const int q = (blk % CPB) * CL + (threadIdx.x % CL);
    const int sl = threadIdx.x / CL;
    const float* src = ws + (long)combo * nsplit * kSlab + q * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int s = sl;
    if (rg.wgt) {
        for (; s < nsplit; s += SL) a += *(const f32x4*)(src + (long)s * kSlab) * rg.wgt[s];
    }
#pragma unroll 1
    for (; s + 3 * SL < nsplit; s += 4 * SL) {
        const f32x4 v0 = *(const f32x4*)(src + (long)s * kSlab);
        const f32x4 v1 = *(const f32x4*)(src + (long)(s + SL) * kSlab);
        const f32x4 v2 = *(const f32x4*)(src + (long)(s + 2 * SL) * kSlab);
        const f32x4 v3 = *(const f32x4*)(src + (long)(s + 3 * SL) * kSlab);
        a += v0; a += v1; a += v2; a += v3;
    }
    for (; s < nsplit; s += SL) a += *(const f32x4*)(src + (long)s * kSlab);
    if (SL > 1) {
        red[sl][threadIdx.x % CL] = a;
        __syncthreads();
        if (sl != 0) return;
#pragma unroll
        for (int k = 1; k < SL; ++k) a += red[k][threadIdx.x % CL];
    }
    const int lane = q & 63, wave = (q >> 6) & 3, m = (q >> 8) & 3, t = q >> 10;
    const int kt = combo / nct, ct = combo - kt * nct;
    const int k0 = kt * 64 + 32 * (wave >> 1) + 8 * m + 4 * (lane >> 5);
    const int e = t * C + ct * 64 + 32 * (wave & 1) + (lane & 31);
#pragma unroll
    for (int j = 0; j < 4; ++j) dw[(long)(k0 + j) * klen + e] = a[j];
}

template <int SL>
__global__ __launch_bounds__(256) void wgrad_patch32_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                   int nsplit, int nct, int C, int klen, ReduceGroup rg) {
    patch32_reduce_body<SL>(ws, dw, nsplit, nct, C, klen, rg, blockIdx.x);
}

// =====================================================================================================================
// v3 ("3 + 3"): the nine taps of a k-step from THREE row-shifted dy fragments and THREE column-shifted x fragments.
//
//   dw[k, (r, s), c] = sum over x pixels (h', w')  dy[h' - r + 1, w', k] * x[h', w' + s - 1, c]
//
// i.e. tap (r, s) = MFMA(Y_r, X_s) with Y_r = the k-step's pixels moved r - 1 rows UP in dy and X_s = the same pixels
// moved s - 1 columns in x: the row shift sits on one operand and the column shift on the other, so a k-step needs
// 3 + 3 fragments instead of v2's 1 + 9 — and for two-row fragments Y_2 of k-step j + 1 IS Y_0 of k-step j (kept in
// registers).  What that buys (the stage body of v2 was bound by its 40 transposing reads and by staging):
//   * fragment reads per 36 MFMAs: 80 (v2) -> 42 (8-row sub-patches of two-row fragments) / 48-56 (others);
//   * staging: x needs only its column halo and dy only its row halo, and a sub-patch may be TALL: 8 x 8 pixels stage
//     80 + 80 slots for 64 pixels (20 DMA pieces) where two 4 x 8 sub-patches staged 2 x 72 + 64 (26);
//   * one barrier per 64 pixels and half where H allows 8-row sub-patches;
//   * 24 fragment registers instead of 80.
// Sub-patch shapes <SW, SH>: 8 x 8 (H % 8 == 0, and 7 x 7 images: one sub-patch each), 8 x 4, 16 x 2.  The k-step
// fragment is 2 rows x 8 columns (SW = 8) or 1 row x 16 columns (SW = 16), as in v2; LDS slots, the g ^ bit1(slot)
// channel-group swizzle, buffer-addressed LDS-DMA with hardware zero fill, the staggered halves, the slab epilogue
// and the ordered reduce are v2's.  Images: x [SH rows][HSX slots] (column halo; HSX = 10 | 20, so that a k-step is a
// multiple of 4 slots = an immediate offset), dy [SH + 2 rows][SW slots] (row halo).
#ifndef PRIMIA_WGP33_PIN
#define PRIMIA_WGP33_PIN 1
#endif
#ifdef WGP33_PROF
__device__ unsigned long long* wgp33_prof_buffer_dev;
#define WGP33_MARK(slot)                               \
    {                                                  \
        const unsigned long long t_now = clock64();    \
        prof_t[slot] += t_now - prof_prev;             \
        prof_prev = t_now;                             \
    }
#else
#define WGP33_MARK(slot)
#endif
// LW ("loader waves", round 5): waves 0-3 are the ONLY matrix waves — one per SIMD, wave (kg, cg) owns out-channels 32 kg ..,
// in-channels 32 cg .. of all nine taps for EVERY sub-patch (both sub-patches of a stage, one after the other: no halves, no
// meeting in LDS) and never issues a DMA instruction or waits on vmcnt; waves 4-7 issue every LDS-DMA piece, wait for their
// own and meet the stage barrier.  r03's phase profile: a wave of the two-halves form spends as long blocked in its five DMA
// issues per stage (~2,000 cycles: the memory pipe's queue is full) as it spends multiplying, and a stage is issue + compute,
// not the maximum of the two, because a blocked issue also holds the wave's own MFMAs behind it.  A third wave per SIMD does
// not fit beside 144 accumulator registers — but ONE matrix wave per SIMD with nine independent accumulators keeps the matrix
// pipe fed by itself, so the second wave of each SIMD can be a pure loader.  (DP-SGD's per-sample forms stay on the two-halves kernel.)
template <int SW, int SH, int STAGES, bool LW = false>
__device__ __forceinline__ void patch33_body(const PatchParams& p, int bid_in) {
    constexpr int FR = SW == 8 ? 2 : 1;         // rows of a k-step fragment
    constexpr int NK = SH / FR;                 // k-steps per sub-patch
    constexpr int HSX = SW == 8 ? 10 : 20;      // x row stride in slots
    constexpr int XSLOTS = (SH * HSX + 7) / 8 * 8;
    constexpr int DSLOTS = (SH + 2) * SW;
    static_assert(DSLOTS % 8 == 0 && SH % FR == 0, "whole DMA pieces");
    constexpr int XP = XSLOTS / 8, DP = DSLOTS / 8;
    constexpr int HALF = (XSLOTS + DSLOTS) * 128;   // one sub-patch: [x image][dy image]
    constexpr int STAGE = 2 * HALF;
    constexpr int NPIECE = 2 * (XP + DP);
    constexpr int NISSUE = LW ? 4 : 8;                 // waves that issue DMA pieces
    constexpr int MAXIT = (NPIECE + NISSUE - 1) / NISSUE;
    static_assert(MAXIT <= 12, "piece bookkeeping");
    typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = LW ? 0 : wave >> 2, kg = (wave >> 1) & 1, cg = wave & 1;
    const bool loader = LW && wave >= 4;
    const int iw = LW ? (wave & 3) : wave;             // index among the issuing waves
#ifdef WGP33_PROF
    unsigned long long prof_t[6] = {0, 0, 0, 0, 0, 0}, prof_prev = clock64();   // [5] = everything before the main loop ends... see marks
#endif

    int bid = bid_in;
    int gi = 0;                                   // layer of a grouped launch (block-uniform)
    if (p.ngroups > 1) {
        gi = bid / p.group_blocks;
        bid -= gi * p.group_blocks;
    }
    int split;
    if (p.split_fastest) {
        split = bid % p.nsplit;
        bid /= p.nsplit;
    }
    const int ct = bid % p.nct; bid /= p.nct;
    const int kt = bid % p.nkt;
    if (!p.split_fastest) split = bid / p.nkt;
    const int t0 = split * p.per_block;
    int t1 = t0 + p.per_block;
    if (t1 > p.total) t1 = p.total;
    // (pairimg: t0 / t1 count IMAGES; the two halves of a stage work on the same sub-patch position of two images)
    const int nstages = p.pairimg ? ((t1 - t0 + 1) >> 1) * p.PPI : (t1 - t0 + 1) >> 1;

    const bf16* __restrict__ x = (gi == 0 ? p.x : p.xg[gi - 1]) + ct * 64;
    const bf16* __restrict__ dy = (gi == 0 ? p.dy : p.dyg[gi - 1]) + kt * 64;

    // ---- staging constants (per DMA piece `it` of this wave): piece idx = wave + 8 it; pieces [q][x: XP | dy: DP] ----
    // row bit i = image row (sub-patch origin) + i - 1, column bit 12 + i = image column origin + i - 1
    int rel[MAXIT];
    unsigned lbits[MAXIT];
    int pq[MAXIT], pisx[MAXIT], pdst[MAXIT];
    const int npc = (LW && !loader) ? 0 : (NPIECE - iw + NISSUE - 1) / NISSUE;            // pieces of this wave (wave-uniform)
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        int idx = iw + NISSUE * it;
        if (idx >= NPIECE) idx = NPIECE - 1;            // (never issued)
        const int q = idx >= XP + DP;
        const int j = idx - q * (XP + DP);
        const bool isx = j < XP;
        const int c16 = lane & 7;
        int ri, ci;
        if (isx) {
            const int slot = j * 8 + (lane >> 3);
            const int hy = slot / HSX, hx = slot - hy * HSX;
            const int grp = (c16 >> 2) ^ key32(slot);
            ri = hy + 1;                                 // rows 0 .. SH-1 of the sub-patch
            ci = hx < SW + 2 && hy < SH ? hx : -1;       // columns -1 .. SW; pad slots: never valid
            rel[it] = ((hy * p.W + (hx - 1)) * p.C + (grp * 4 + (c16 & 3)) * 8) * 2;
        } else {
            const int slot = (j - XP) * 8 + (lane >> 3);
            const int py = slot / SW, px = slot - py * SW;
            const int grp = (c16 >> 2) ^ key32(slot);
            ri = py;                                     // rows -1 .. SH
            ci = px + 1;
            rel[it] = (((py - 1) * p.W + px) * p.K + (grp * 4 + (c16 & 3)) * 8) * 2;
        }
        lbits[it] = ci < 0 ? 0x80000000u : (1u << ri) | (1u << (12 + ci));
        pisx[it] = __builtin_amdgcn_readfirstlane((int)isx);
        pq[it] = __builtin_amdgcn_readfirstlane(q);
        pdst[it] = __builtin_amdgcn_readfirstlane(q * HALF + (isx ? j * 1024 : XSLOTS * 128 + (j - XP) * 1024));
    }
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const long xbytes = ((long)p.nimg * p.H * p.W * p.C - ct * 64) * 2;
    const long dybytes = ((long)p.nimg * p.H * p.W * p.K - kt * 64) * 2;
    auto make_rsrc = [](const void* base, long bytes) {
        const unsigned long long a = (unsigned long long)base;
        i32x4 r;
        r[0] = (int)(unsigned)a;
        r[1] = (int)(unsigned)(a >> 32) & 0xffff;       // stride 0: raw buffer
        r[2] = (int)(unsigned)(bytes > 0xfffffff0L ? 0xfffffff0L : bytes);
        r[3] = 0x00020000;
        return r;
    };
    i32x4 rsrc_x = make_rsrc(x, xbytes), rsrc_dy = make_rsrc(dy, dybytes);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rsrc_x[j] = __builtin_amdgcn_readfirstlane(rsrc_x[j]);
        rsrc_dy[j] = __builtin_amdgcn_readfirstlane(rsrc_dy[j]);
    }
    constexpr unsigned kOob = 0xfffffff0u;
    int sn, sph, spw;
    if (p.pairimg) {
        sn = t0; sph = 0; spw = 0;
    } else {
        sn = t0 / p.PPI;
        const int rem = t0 - sn * p.PPI;
        sph = rem / p.PW;
        spw = rem - sph * p.PW;
    }
    int st = t0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned voff[MAXIT], gdst[MAXIT];
    auto prep = [&](int buf) {
        unsigned smask[2];
        int xo[2], dyo[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool live = p.pairimg ? sn + q < t1 : st < t1;
            const int rb = sph * SH, cb = spw * SW;
            const int pixbase = ((p.pairimg ? sn + q : sn) * p.H + rb) * p.W + cb;
            // valid row bits: image row rb + i - 1 in [0, H); columns likewise from bit 12
            int rhi = p.H - rb + 1, chi = p.W - cb + 1;
            rhi = rhi > SH + 2 ? SH + 2 : rhi;
            chi = chi > SW + 2 ? SW + 2 : chi;
            const unsigned rowm = ((1u << rhi) - 1u) & ~(rb == 0 ? 1u : 0u);
            const unsigned colm = ((1u << chi) - 1u) & ~(cb == 0 ? 1u : 0u);
            smask[q] = live ? rowm | (colm << 12) : 0u;
            xo[q] = pixbase * p.C * 2;
            dyo[q] = pixbase * p.K * 2;
            if (p.pairimg && q == 0) continue;     // both halves: the same sub-patch position, images sn and sn + 1
            ++st;
            if (++spw == p.PW) {
                spw = 0;
                if (++sph == p.PH) {
                    sph = 0;
                    sn += p.pairimg ? 2 : 1;
                }
            }
        }
        const unsigned base = lds0 + buf * STAGE;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if (it >= npc) break;
            const unsigned m = pq[it] ? smask[1] : smask[0];
            const int org = pisx[it] ? (pq[it] ? xo[1] : xo[0]) : (pq[it] ? dyo[1] : dyo[0]);
            voff[it] = (lbits[it] & m) == lbits[it] ? (unsigned)(rel[it] + org) : kOob;
            gdst[it] = base + pdst[it];
        }
    };
    bool do_issue = true;
    auto issue = [&](int it) {
        if (do_issue && it < npc) {
            const unsigned m0v = __builtin_amdgcn_readfirstlane(gdst[it]);
            if (pisx[it])
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(voff[it]), "s"(rsrc_x), "s"(m0v) : "memory");
            else
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(voff[it]), "s"(rsrc_dy), "s"(m0v) : "memory");
        }
    };
    auto stage = [&](int buf) {
        prep(buf);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) issue(it);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

    // ---- fragment addresses: lane (16-lane group g16, fr) reads pixel kk = 8*(g16>>1) + (fr>>2) (+4) of the k-step,
    //      channels 16*(g16&1) + 4*(fr&3) .. +3 of the wave's group; k-steps and row shifts are immediate offsets --------
    const int fr = lane & 15, g16 = lane >> 4;
    const int cbyte = (16 * (g16 & 1) + 4 * (fr & 3)) * 2;
    int offy[2], offx[3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kk = 8 * (g16 >> 1) + (fr >> 2) + 4 * h;
        const int py = SW == 8 ? (kk >> 3) : 0, px = SW == 8 ? (kk & 7) : kk;
        const int sy = py * SW + px;                       // dy image, row offset y = 0
        offy[h] = half * HALF + XSLOTS * 128 + sy * 128 + ((kg ^ key32(sy)) << 6) + cbyte;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int sx = py * HSX + px + s;              // x image, k-step 0
            offx[s][h] = half * HALF + sx * 128 + ((cg ^ key32(sx)) << 6) + cbyte;
        }
    }
    constexpr int KSTEP_X = FR * HSX * 128;    // next k-step: FR rows further in both images
    constexpr int YROW = SW * 128;             // one dy row

    // One stage of this wave's sub-patch: NK k-steps x 9 MFMAs.  Y[y] = dy fragment whose first row is image row y - 1
    // of the sub-patch; tap row r of k-step j needs Y[FR j + 2 - r].
    constexpr bool prio = false;       // (s_setprio around the MFMAs: measured, no gain — profiles/r03_negative_results.txt)
    auto compute = [&](int buf, int q_half = 0) {
        typedef __attribute__((address_space(3))) char* ldsp_t;
        const ldsp_t sb = (ldsp_t)(size_t)(lds0 + buf * STAGE + q_half * HALF);
        ldsp_t py_[2], px_[3][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            py_[h] = sb + offy[h];
#pragma unroll
            for (int s = 0; s < 3; ++s) px_[s][h] = sb + offx[s][h];
        }
        constexpr int NY = FR * (NK - 1) + 3;
        bf16x8_t Y[NY], X[2][3];
        auto read_y = [&](int y) {
            bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(py_[0] + y * YROW));
            bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(py_[1] + y * YROW));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto read_x = [&](int j, int s) {
            bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(px_[s][0] + j * KSTEP_X));
            bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(px_[s][1] + j * KSTEP_X));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        if (WGP33_DBG & 8) {
#pragma unroll
            for (int y = 0; y < NY; ++y) asm volatile("" : "=v"(Y[y]));
#pragma unroll
            for (int s = 0; s < 3; ++s) { asm volatile("" : "=v"(X[0][s])); asm volatile("" : "=v"(X[1][s])); }
        } else {
#pragma unroll
            for (int y = 0; y < 3; ++y) Y[y] = read_y(y);
#pragma unroll
            for (int s = 0; s < 3; ++s) X[0][s] = read_x(0, s);
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            // the fragments of k-step j + 1 are requested BEFORE the MFMAs of k-step j (left to itself the scheduler
            // issues them right in front of their first use: ~100-200 cycles of LDS latency per k-step in the open)
            if (j + 1 < NK && !(WGP33_DBG & 8)) {
#pragma unroll
                for (int s = 0; s < 3; ++s) X[(j + 1) & 1][s] = read_x(j + 1, s);
#pragma unroll
                for (int y = FR * j + 3; y < FR * (j + 1) + 3; ++y) Y[y] = read_y(y);
            }
            if (PRIMIA_WGP33_PIN) __builtin_amdgcn_sched_barrier(0);
            if (prio) __builtin_amdgcn_s_setprio(1);   // the partner wave's reads / DMA issue must not take slots from the MFMAs
            if (WGP33_DBG & 4) {
#pragma unroll
                for (int y = 0; y < NY; ++y) asm volatile("" ::"v"(Y[y]));
#pragma unroll
                for (int s = 0; s < 3; ++s) asm volatile("" ::"v"(X[j & 1][s]));
            } else {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    acc[3 * r + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Y[FR * j + 2 - r], X[j & 1][s], acc[3 * r + s], 0, 0, 0);
            }
            if (prio) __builtin_amdgcn_s_setprio(0);
            if (PRIMIA_WGP33_PIN) __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto wait_inflight = [&](int stages_ahead) {   // this wave's pieces of the newest `stages_ahead` stages may stay in flight
        const int n = stages_ahead * npc;
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (more pieces than the cases above: wait for all)
        }
    };

    if (!LW || loader) {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nstages) stage(s);
    }
    if constexpr (LW) {
        int cur = 0, nxt = STAGES - 1;
        if (loader) {
            for (int s = 0; s < nstages; ++s) {
                int ahead = nstages - 1 - s;
                if (ahead > STAGES - 2) ahead = STAGES - 2;
                wait_inflight(ahead);                  // this wave's pieces of stage s have landed
                WGP33_MARK(1)
                __builtin_amdgcn_s_barrier();          // ... everybody's have, and stage s - 1 has been multiplied
                WGP33_MARK(2)
                do_issue = s + STAGES - 1 < nstages && !(WGP33_DBG & 2);
                if (do_issue) stage(nxt);              // into the buffer stage s - 1 occupied
                WGP33_MARK(0)
                nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
            }
#ifdef WGP33_PROF
            if (lane == 0 && wgp33_prof_buffer_dev)
                for (int k = 0; k < 6; ++k) wgp33_prof_buffer_dev[((long)blockIdx.x * 8 + wave) * 6 + k] = prof_t[k];
#endif
            return;
        }
        for (int s = 0; s < nstages; ++s) {
            __builtin_amdgcn_s_barrier();
            WGP33_MARK(2)
            compute(cur, 0);
            compute(cur, 1);
            WGP33_MARK(3)
            cur = cur + 1 == STAGES ? 0 : cur + 1;
        }
#ifdef WGP33_PROF
        if (lane == 0 && wgp33_prof_buffer_dev)
            for (int k = 0; k < 6; ++k) wgp33_prof_buffer_dev[((long)blockIdx.x * 8 + wave) * 6 + k] = prof_t[k];
#endif
        if (!(WGP33_DBG & 1)) wgrad32_epilogue<false>(acc, p, smem, wave, lane, 0, kg, cg, kt + gi * p.nkt, ct, split);
        else {
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(acc[t]));
        }
        return;
    }
    // the two waves of a SIMD belong to different halves: half 0 issues its DMA pieces and THEN multiplies, half 1
    // multiplies and THEN issues (see v2); two copies of the loop, one order each
    int sp_done = 0, img_pair = 0;     // pairimg: stages done of the current image pair; pairs done
    WGP33_MARK(5)
    auto main_loop = [&](auto stage_first) {
        int cur = 0, nxt = STAGES - 1;
        for (int s = 0; s < nstages; ++s) {
            int ahead = nstages - 1 - s;
            if (ahead > STAGES - 2) ahead = STAGES - 2;
            wait_inflight(ahead);
            WGP33_MARK(1)
            __builtin_amdgcn_s_barrier();
            WGP33_MARK(2)
            do_issue = s + STAGES - 1 < nstages && !(WGP33_DBG & 2);
            if constexpr (decltype(stage_first)::value) {
                if (do_issue) stage(nxt);
                WGP33_MARK(0)
                compute(cur);
                WGP33_MARK(3)
            } else {
                compute(cur);
                WGP33_MARK(3)
                if (do_issue) stage(nxt);
                WGP33_MARK(0)
            }
            if (p.pairimg && ++sp_done == p.PPI) {
                // DP-SGD norm pass: each half has just finished a whole image of its own, and the wave's accumulators are
                // its complete share of that image's gradient tile — add its squared norm and start over.  (One block per
                // (image, slab) — 16,384 blocks of a single stage for layer4 at batch 256, each with the prologue and the
                // 144-KiB meeting of the halves — took 530 us per layer; this form ~40.)
                sp_done = 0;
                const int img = t0 + 2 * img_pair + half;
                ++img_pair;
                double sq = 0.0;
                if (p.ws && img < t1) {     // (stores first: they drain while the squares are summed)
                    float* o = p.ws + ((long)(kt * p.nct + ct) * p.nimg + img) * kSlab + ((wave & 3) * 64 + lane) * 4;
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            *(f32x4*)(o + (t * 4 + m) * 1024) =
                                f32x4{acc[t][4 * m], acc[t][4 * m + 1], acc[t][4 * m + 2], acc[t][4 * m + 3]};
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        sq += (double)acc[t][j] * (double)acc[t][j];
                        acc[t][j] = 0.f;
                    }
                if (img < t1) wave_sqnorm_add(sq, p.sqnorm + img);
            }
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        }
    };
    if (half == 0)
        main_loop(OrderTag<true>{});
    else
        main_loop(OrderTag<false>{});

    if (p.pairimg) return;
    WGP33_MARK(5)
    // (grouped launch: the slab index of layer gi = that of layer 0 + gi * combos * nsplit, i.e. kt + gi * nkt)
    if (!(WGP33_DBG & 1)) wgrad32_epilogue(acc, p, smem, wave, lane, half, kg, cg, kt + gi * p.nkt, ct, split);
    else {
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(acc[t]));
    }
    WGP33_MARK(4)
#ifdef WGP33_PROF
    if (lane == 0 && wgp33_prof_buffer_dev) {
#pragma unroll
        for (int k = 0; k < 6; ++k) wgp33_prof_buffer_dev[((long)blockIdx.x * 8 + wave) * 6 + k] = prof_t[k];
    }
#endif
}

template <int SW, int SH, int STAGES>
__global__ __launch_bounds__(512) void conv_wgrad_patch33_kernel(PatchParams p) {
    patch33_body<SW, SH, STAGES>(p, xcd_remap(blockIdx.x, gridDim.x));
}

// the loader-wave form (option wgp_lw; the batched gradient only)
template <int SW, int SH, int STAGES>
__global__ __launch_bounds__(512) void conv_wgrad_patch33lw_kernel(PatchParams p) {
    patch33_body<SW, SH, STAGES, true>(p, xcd_remap(blockIdx.x, gridDim.x));
}

struct PatchGeom {
    bool ok, wide;
    int SW, SH;        // v3: sub-patch shape
    int PH, PW, PPI, total, per_block, nsplit, combos;
};

// DP-SGD norm pass with whole images per half: needs at least two images per block for every slab
static bool pairimg_mode(const WgradParams& w, const PatchGeom& g) {
    return w.persample && w.sqnorm && PRIMIA_OPT(wgp_pairimg) && w.N >= 2 && (long)g.combos * ((w.N + 1) / 2) >= 256;   // (else: one block per image, as before)
}
// ngroup > 1: geometry of ONE layer of a grouped launch (the layers share the 256 CUs)
static PatchGeom patch_geom(const WgradParams& w, int ngroup = 1) {
    PatchGeom g{};
    // x and dy are addressed with 32-bit BYTE offsets through buffer resources (signed arithmetic, range check against
    // num_records): elements < 2^30
    g.ok = !(w.R != 3 || w.S != 3 || w.stride != 1 || w.pad != 1 || w.C % 64 || w.K % 64) &&
           (long)w.N * w.H * w.W * (w.C > w.K ? w.C : w.K) < (1L << 30);
    if (!g.ok) return g;
    int SW, SH;
    {
        // 8 x 8 unless it pads the image more than 35 % beyond the best shape: measured at batch 256, us per call incl.
        // the reduce, 8 x 8 | best-fitting shape: 28 x 28 images (pads to 32 x 32, +14 %) 77 | 81 (8 x 4), 14 x 14 (16 x 16,
        // +14 %) 75 | 83 (16 x 2) — a stage of 64 pixels per half with one barrier beats 32 pixels without padding
        // (7-row bands of 16-column strips for H = 28 / 14 were measured: no gain, 9 spilled registers — not kept)
        static const int cand[3][2] = {{8, 8}, {8, 4}, {16, 2}};
        const int force = PRIMIA_OPT(wgp_shape);       // 0 .. 2, -1: by image size
        long slots[3], best = -1;
        for (int i = 0; i < 3; ++i) {
            slots[i] = (long)((w.W + cand[i][0] - 1) / cand[i][0] * cand[i][0]) * ((w.H + cand[i][1] - 1) / cand[i][1] * cand[i][1]);
            if (best < 0 || slots[i] < best) best = slots[i];
        }
        int pick = 0;
        if (force >= 0 && force < 3) {
            pick = force;
        } else if (slots[0] * 100 > best * 135) {
            pick = slots[1] <= slots[2] ? 1 : 2;
        }
        SW = cand[pick][0]; SH = cand[pick][1];
        g.wide = SW == 16;
    }
    g.SW = SW; g.SH = SH;
    g.PH = (w.H + SH - 1) / SH; g.PW = (w.W + SW - 1) / SW; g.PPI = g.PH * g.PW;
    g.total = w.N * g.PPI;
    g.combos = (w.C / 64) * (w.K / 64);
    // one 8-wave block per CU
    const int target_blocks = PRIMIA_OPT(wgp_blocks);
    const int target = target_blocks ? target_blocks : 256;
    long want = (target + g.combos - 1) / g.combos;
    if (ngroup > 1) want = target / ((long)ngroup * g.combos);      // all layers' blocks in ONE round
    if (want < 1) want = 1;
    long per = (g.total + want - 1) / want;
    per = (per + 1) & ~1L;
    if (per < 2) per = 2;
    if (w.persample) per = g.PPI;  // one split per image
    if (pairimg_mode(w, g)) {
        // norm pass: a block walks many images, each half whole images of its own (see the kernel); units: images
        g.total = w.N;
        per = (w.N + want - 1) / want;
        per = (per + 1) & ~1L;
        if (per < 2) per = 2;
    }
    g.per_block = (int)per;
    g.nsplit = (int)((g.total + per - 1) / per);
    return g;
}

// bytes of workspace the store-and-reduce path needs for this layer (0: layer not served by this kernel)
size_t wgrad_patch_ws_bytes(const WgradParams& w) {
    const PatchGeom g = patch_geom(w);
    if (!g.ok || w.persample) return 0;
    return (size_t)g.combos * g.nsplit * kSlab * sizeof(float);
}

static void fill_patch_params(PatchParams& p, const WgradParams& w, const PatchGeom& g) {
    p.x = (const bf16*)w.x; p.dy = (const bf16*)w.dy; p.dw = w.dw;
    p.H = w.H; p.W = w.W; p.C = w.C; p.K = w.K; p.klen = w.klen;
    p.nct = w.C / 64; p.nkt = w.K / 64;
    p.PH = g.PH; p.PW = g.PW; p.PPI = g.PPI;
    p.total = g.total;
    p.per_block = g.per_block;
    p.nsplit = g.nsplit;
    p.split_fastest = PRIMIA_OPT(wgp_order);
    p.split_stride = w.persample ? (long)w.K * w.klen : 0;
    p.sqnorm = w.persample ? w.sqnorm : nullptr;
    const bool store = !w.persample && w.ws && w.ws_bytes >= (size_t)g.combos * g.nsplit * kSlab * sizeof(float);
    p.ws = store ? w.ws : nullptr;
    p.pairimg = pairimg_mode(w, g) ? 1 : 0;
    p.nimg = w.N;
    p.ngroups = 1;
    p.group_blocks = 0;
}

template <int SW, int SH, int STAGES = 3>
static int launch_patch33(const WgradParams& w, const PatchGeom& g, hipStream_t st) {
    PatchParams p;
    fill_patch_params(p, w, g);
    // the stage ring (at most 144 KiB; 160 KiB for four 8 x 8 stages), or the 144 KiB the two halves need to meet in
    // after the main loop
    size_t lds = (size_t)kSlab * 4;
    if (SW == 8 && SH == 8 && STAGES == 4) lds = 163840;
    const bool lw = PRIMIA_OPT(wgp_lw) && !p.pairimg && !p.sqnorm;
    void (*kern)(PatchParams) = lw ? conv_wgrad_patch33lw_kernel<SW, SH, STAGES> : conv_wgrad_patch33_kernel<SW, SH, STAGES>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[lw]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set[lw] = true;
    }
    kern<<<(unsigned)(g.combos * g.nsplit), 512, lds, st>>>(p);
    if (p.ws) {
        const int ns = g.nsplit;
        if (ns >= 64)
            wgrad_patch32_reduce_kernel<16><<<g.combos * (kSlab / 4 / 16), 256, 0, st>>>(w.ws, w.dw, ns, p.nct, w.C, w.klen, ReduceGroup{});
        else if (ns >= 8)
            wgrad_patch32_reduce_kernel<4><<<g.combos * (kSlab / 4 / 64), 256, 0, st>>>(w.ws, w.dw, ns, p.nct, w.C, w.klen, ReduceGroup{});
        else
            wgrad_patch32_reduce_kernel<1><<<g.combos * (kSlab / 4 / 256), 256, 0, st>>>(w.ws, w.dw, ns, p.nct, w.C, w.klen, ReduceGroup{});
    }
    return launch_status();
}

// ---- grouped launch: up to four layers of ONE shape in one launch of conv_wgrad_patch33_kernel<8, 8> -------------------
// A call of this kernel carries ~25-29 us that do not shrink with the work (launch, lane constants, first DMA round trip,
// the halves' meeting + slab store, the reduce launch: profiles/r03_wgp33_phase_profile.txt) next to 45-55 us of main
// loop; with n layers the blocks split n ways — each block walks n times as many sub-patches behind ONE such cost, and
// each layer writes 1/n of the slabs.  Weight gradients are leaves of the backward graph and the layers of a ResNet
// stage share a shape, so the caller may hold a layer's (x, dy) back until its siblings' are ready.
// Preferred group size for `count` layers of this shape: the largest n <= min(count, 4) whose blocks fill >= 90 % of the
// CUs in one round (0: shape not served).
int wgrad_patch_group_size(const WgradParams& w, int count) {
    if (!PRIMIA_OPT(wgp_group) || w.persample) return 0;
    const PatchGeom g1 = patch_geom(w);
    if (!g1.ok || g1.SW != 8 || g1.SH != 8) return 0;
    for (int n = count < 4 ? count : 4; n >= 2; --n) {
        const PatchGeom g = patch_geom(w, n);
        const long blocks = (long)n * g.combos * g.nsplit;
        const int minfill = PRIMIA_OPT(wgp_group_minfill);
        if (blocks <= 256 && blocks * 100 >= 256 * minfill) return n;
    }
    return 1;
}

size_t wgrad_patch_group_ws_bytes(const WgradParams& w, int n) {
    if (n < 2 || n > 4 || wgrad_patch_group_size(w, n) < n) return 0;
    const PatchGeom g = patch_geom(w, n);
    return (size_t)n * g.combos * g.nsplit * kSlab * sizeof(float);
}

// ws[0 .. n): the layers (same N, H, W, C, K; 3 x 3 / stride 1); the workspace of layer 0 is the group's
int wgrad_patch_group_dispatch(const WgradParams* ws_, int n, hipStream_t st) {
    const WgradParams& w = ws_[0];
    const size_t need = wgrad_patch_group_ws_bytes(w, n);
    if (!need || !w.ws || w.ws_bytes < need) return PRIMIA_ERR_UNSUPPORTED;
    for (int i = 0; i < n; ++i)
        if (!ws_[i].x || !ws_[i].dy || !ws_[i].dw || ws_[i].N != w.N || ws_[i].H != w.H || ws_[i].W != w.W ||
            ws_[i].C != w.C || ws_[i].K != w.K || ws_[i].R != 3 || ws_[i].S != 3 || ws_[i].stride != 1 || ws_[i].pad != 1)
            return PRIMIA_ERR_ARG;
    const PatchGeom g = patch_geom(w, n);
    PatchParams p;
    WgradParams w0 = w;
    w0.ws_bytes = (size_t)g.combos * g.nsplit * kSlab * sizeof(float);   // (fill_patch_params checks one layer's share)
    fill_patch_params(p, w0, g);
    p.ws = w.ws;
    p.pairimg = 0;
    p.ngroups = n;
    p.group_blocks = g.combos * g.nsplit;
    ReduceGroup rg{};
    rg.combos = g.combos;
    for (int i = 1; i < n; ++i) {
        p.xg[i - 1] = (const bf16*)ws_[i].x;
        p.dyg[i - 1] = (const bf16*)ws_[i].dy;
        rg.dwg[i - 1] = ws_[i].dw;
    }
    const size_t lds = (size_t)kSlab * 4;
    const bool lw = PRIMIA_OPT(wgp_lw) != 0;
    void (*kern)(PatchParams) = lw ? conv_wgrad_patch33lw_kernel<8, 8, 3> : conv_wgrad_patch33_kernel<8, 8, 3>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[lw]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set[lw] = true;
    }
    kern<<<(unsigned)(n * p.group_blocks), 512, lds, st>>>(p);
    const int ns = g.nsplit, all = n * g.combos;
    if (ns >= 64)
        wgrad_patch32_reduce_kernel<16><<<all * (kSlab / 4 / 16), 256, 0, st>>>(w.ws, w.dw, ns, p.nct, w.C, w.klen, rg);
    else if (ns >= 8)
        wgrad_patch32_reduce_kernel<4><<<all * (kSlab / 4 / 64), 256, 0, st>>>(w.ws, w.dw, ns, p.nct, w.C, w.klen, rg);
    else
        wgrad_patch32_reduce_kernel<1><<<all * (kSlab / 4 / 256), 256, 0, st>>>(w.ws, w.dw, ns, p.nct, w.C, w.klen, rg);
    return launch_status();
}

// ---- DP-SGD: norm pass that KEEPS every sample's tiles, clipped sum as a weighted reduce (see conv_wgrad.hip) ----------
size_t wgrad_patch_keep_bytes(const WgradParams& w) {
    const long budget = (long)PRIMIA_OPT(dp_keep_mb) << 20;   // (layer1: 38 MB per layer, layer2: 151; layer3 would be 604 MB written and read back: a loss)
    WgradParams q = w;
    q.persample = 1;
    double dummy;
    q.sqnorm = &dummy;
    const PatchGeom g = patch_geom(q);
    if (!g.ok) return 0;       // (whole images per half, or one block per (image, slab): both keep)
    const size_t n = (size_t)g.combos * w.N * kSlab * sizeof(float);
    return (long)n <= budget ? n : 0;
}

int wgrad_patch_keep_dispatch(const WgradParams& w, hipStream_t st) {
    const size_t need = wgrad_patch_keep_bytes(w);
    if (!need || !w.sqnorm || !w.ws) return PRIMIA_ERR_UNSUPPORTED;
    if (w.ws_bytes < need) return PRIMIA_ERR_WORKSPACE;
    const PatchGeom g = patch_geom(w);
    PatchParams p;
    fill_patch_params(p, w, g);      // (persample: p.ws = null)
    if (!p.pairimg && g.nsplit != w.N) return PRIMIA_ERR_UNSUPPORTED;
    p.ws = w.ws;
    const size_t lds = (size_t)kSlab * 4;
    void (*kern)(PatchParams);
    if (g.SW == 16) kern = g.SH == 7 ? conv_wgrad_patch33_kernel<16, 7, 2> : conv_wgrad_patch33_kernel<16, 2, 3>;
    else kern = g.SH == 8 ? conv_wgrad_patch33_kernel<8, 8, 3> : conv_wgrad_patch33_kernel<8, 4, 3>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PRIMIA_ERR_LAUNCH;
    kern<<<(unsigned)(g.combos * g.nsplit), 512, lds, st>>>(p);
    return launch_status();
}

int wgrad_patch_clipped_sum(const WgradParams& w, const float* slabs, const float* clip, hipStream_t st) {
    if (!wgrad_patch_keep_bytes(w) || !w.dw) return PRIMIA_ERR_UNSUPPORTED;
    const int combos = (w.C / 64) * (w.K / 64);
    ReduceGroup rg{};
    rg.wgt = clip;
    wgrad_patch32_reduce_kernel<16><<<combos * (kSlab / 4 / 16), 256, 0, st>>>(slabs, w.dw, w.N, w.C / 64, w.C, w.klen, rg);
    return launch_status();
}

// 16 = conv_wgrad_patch33_kernel, 18 = conv_wgrad_patch33lw_kernel, 11 = conv_wgrad_patch32_kernel, 12 = conv_wgrad_patch_kernel (round 1), 0 = shape not served
int wgrad_patch_kernel_id(const WgradParams& w) {
    if (!patch_geom(w).ok) return 0;
    return PRIMIA_OPT(wgp_lw) ? 18 : 16;       // 18 = conv_wgrad_patch33lw_kernel (loader waves; the batched gradient)
}

// DP-SGD norm pass on this kernel: 0 shape not served, 24 one block per (image, slab), 25 whole images per half-block
int wgrad_patch_persample_kernel_id(const WgradParams& w) {
    WgradParams q = w;
    static double dummy;
    q.persample = 1;
    q.sqnorm = &dummy;       // (never dereferenced: the mode test only asks whether a norm pass was requested)
    const PatchGeom g = patch_geom(q);
    if (!g.ok) return 0;
    return pairimg_mode(q, g) ? 25 : 24;
}

int wgrad_patch_dispatch(const WgradParams& w, hipStream_t st) {
    const PatchGeom g = patch_geom(w);
    if (!g.ok) return PRIMIA_ERR_UNSUPPORTED;
    if (g.SW == 16) return launch_patch33<16, 2>(w, g, st);
    // (8 x 8: four stages of 40 KiB are exactly the CU's 160 KiB of LDS)
    if (g.SH == 8 && PRIMIA_OPT(wgp_stages88) == 4) return launch_patch33<8, 8, 4>(w, g, st);
    return g.SH == 8 ? launch_patch33<8, 8>(w, g, st) : launch_patch33<8, 4>(w, g, st);
}

}  // namespace primia
