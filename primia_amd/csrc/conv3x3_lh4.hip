// 3x3 / stride-1 / pad-1 convolution, forward and data gradient, 196-pixel x 128-channel tiles (ResNet-18 layer4 at batch 256, and
// every wide shape too small for conv3x3_lh2's 392-pixel tiles) — the linear-halo scheme of conv3x3_lh2.hip rebuilt around what
// tools/micro/mfma_fill.hip measured in round 4: what staging costs the matrix pipe is the ISSUE of the LDS-DMA instructions by
// waves that also multiply (a blocked issue holds the wave's MFMAs behind it), not the data movement.
//   * 12 waves: 8 MATRIX waves that never issue a DMA instruction and never wait on vmcnt (their own stores share that counter),
//     4 LOADER waves that do nothing else.  Three waves per SIMD = at most 168 VGPRs: the 196-pixel tile (4 | 3 | 3 | 3 fragments
//     of 16 pixels: 64 accumulator + 64 fragment registers) fits, a 392-pixel tile (7 fragments) does not.
//   * no ping-pong halves: every matrix wave reads its fragments and multiplies, step after step, with ONE barrier per step (the
//     waves of a SIMD drift apart on their own, as in the synthetic stage body).
//   * weights: whole 128 x 64 step tiles through a ring of 3 — the loaders run two steps ahead; halo: double buffer per 64-channel
//     chunk, the next chunk's 32 pieces spread over the first seven taps of the current one.
//   * tiles, order of every sum, write-back and BatchNorm partial sums are conv3x3_lh2's for its 196-pixel tiles: outputs and
//     partials are BIT-IDENTICAL to that kernel (tools/micro/lh4_bench.hip checks all nine cases), so every parity test of the 3x3
//     layers holds unchanged.  Stand-alone (post-ReLU operands, batch 256): layer4 forward 55.2 -> 51.0 us, data gradient 53.7 ->
//     48.2, accumulating data gradient 56.2 -> 51.3; in the training step (same-box A/B, 3 x 100 steps): 5.03 -> 5.00 ms.
//     Forms that did NOT pay (profiles/r04_mfma_ceilings.txt): loader waves bolted onto lh2's ping-pong (+5 % stand-alone, nothing
//     in the step), split weight rings + streamed fragments so that 392-pixel tiles fit 168 registers (slower everywhere).
// The per-tile write-back (plain / accumulate / masked forms, BatchNorm partial sums) is ONE function shared with conv3x3_lh2.hip
// (conv3x3_lh.h: lh_tile_writeback): the two kernels produce the same bits on the same tiles by construction, and
// tests/test_gpu_ops.py::test_loader_wave_kernel_is_bit_identical_to_the_linear_halo_kernel still checks it.
#include <stdlib.h>

#include "conv3x3_lh.h"

namespace primia {

typedef int i32x4_t __attribute__((ext_vector_type(4)));

struct Lh4Params {
    const bf16* src;   // [M][Cs]
    const bf16* wt;    // [Nd][9][Cs]
    bf16* dst;         // [M][Nd]
    int H, W, Cs, Nd;
    int M;             // N*H*W
    const uint8_t* acc_mask;   // accumulate form: ReLU mask bits applied to the OLD values (one byte per 8 channels)
    float* stat_partials;      // BatchNorm partial sums [tiles_m][2][Nd] (or null): forward: of the values AS STORED; plain data gradient
                               // with bnb.y: of the BatchNorm backward of the layer before (conv3x3_lh.h: LhBnBwd)
    LhBnBwd bnb;
    int ntile_n, ntiles;
    unsigned magicW, magicH;   // ceil(2^16 / W), ceil(2^16 / H)
    unsigned long long* prof;  // LH4_PROF builds: [block][wave][4] cycles in load / matrix / barrier-wait / write-back
};

// compile-time experiment switches (tools/micro/lh2_bench.hip): 1 no write-back (accumulators kept alive), 2 no DMA
// after the prologue, 4 no MFMA, 8 no fragment reads
#ifndef LH4_DBG
#define LH4_DBG 0
#endif
#ifndef LH4_PRIO
#define LH4_PRIO 1
#endif
#ifndef LH4_SWAP
#define LH4_SWAP 0
#endif
#ifndef LH4_NT
#define LH4_NT 1
#endif
#ifndef LH4_RF
#define LH4_RF 0
#endif
#ifdef LH4_PROF
#define LH4_MARK(slot)                                 \
    {                                                  \
        const unsigned long long t_now = clock64();    \
        prof_t[slot] += t_now - prof_prev;             \
        prof_prev = t_now;                             \
    }
#else
#define LH4_MARK(slot)
#endif

// Chunk swizzle of every 64-byte row (halo slots, weight rows): 16-byte chunk c of row r sits at c ^ l4_key(r >> 2).
// A ds_read_b128 is served in four groups of 16 lanes that are NOT consecutive lanes — {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) — so a group mixes eight lanes of one channel chunk (rows 0-3 and
// 12-15 of a fragment) with eight of the next (rows 4-11).  Rows that share a bank quarter are 4 apart; the key must make
// key[g], key[g + 3], key[g + 1] ^ 1, key[g + 2] ^ 1 pairwise distinct for every g.  Round 3's table {0, 2, 3, 1} did that
// only for g = 0 and 2, i.e. for fragments starting at a slot that is a multiple of 8: every tap shift (+-1, +-W) paid a
// 2-way conflict on every pixel fragment read (SQ_LDS_BANK_CONFLICT 3.75 M of 10 M LDS cycles per launch,
// profiles/r04_stall_counters.txt).  {0, 2, 0, 2} is conflict-free at EVERY alignment.
__device__ __forceinline__ int l4_key(int quad) { return (quad & 1) << 1; }

__device__ __forceinline__ void l4_dma(unsigned voff, i32x4_t rsrc, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ i32x4_t l4_rsrc(const void* base, long bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4_t r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)(unsigned)(a >> 32) & 0xffff;       // stride 0: raw buffer
    r[2] = (int)(unsigned)(bytes > 0x7ffffff0L ? 0x7ffffff0L : bytes);
    r[3] = 0x00020000;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_readfirstlane(r[j]);
    return r;
}

__device__ __forceinline__ void l4_wait_vmcnt(int n) {   // wave-uniform n
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

template <int J>
struct LhJ4 {
    static constexpr int value = J;
};

constexpr int kL4Slots = 256;                       // halo slots per buffer (196 + 2 * 28 + 2 = 254 live at most)
constexpr int kL4Plane = kL4Slots * 64;             // 16 KiB: one 32-channel half of every slot (16 DMA pieces)
constexpr int kL4Halo = 2 * kL4Plane;               // 32 KiB
constexpr int kL4Half = 128 * 64;                   // 8 KiB: one 32-channel half of a step's weight tile
constexpr int kL4WTile = 2 * kL4Half;               // [half 0][half 1]
constexpr int kL4OffW = 2 * kL4Halo;                // ring of 3 step tiles
constexpr int kL4OffScr = kL4OffW + 3 * kL4WTile;   // BatchNorm partials of the four pixel groups
constexpr int kL4Lds = kL4OffScr + 4 * 2 * 128 * 4; // 118,784 B
constexpr int kL4ZeroSlot = kL4Slots - 1;
constexpr unsigned kL4Oob = 0xfffffff0u;


// One MATRIX wave's whole life.  JW = pixel fragments of this wave, F0 = its first fragment.
template <int BM, int JW, int F0, bool FLIP, bool ACC>
__device__ __forceinline__ void lh4_run(const Lh4Params& p, char* smem, int tile_first, int tile_count) {
    constexpr bool ISA = false;        // (the statistics of a tile are combined by waves 4-7, as lh2's second half does)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int W = p.W, H = p.H, Cs = p.Cs, Nd = p.Nd;
    const int nchunks = Cs >> 6;
    // ---- fragment read addresses ---------------------------------------------------------------------------
    // weights: row wn*64 + 16i + fr of a half tile, 16-byte chunk fg; + i * 1024
    const int aoff = (wn * 64 + fr) * 64 + ((fg ^ l4_key(fr >> 2)) << 4);   // + kL4Half for the second 32 channels
    // pixels: fragment j covers tile pixels 16 (F0 + j) + fr, halo slot at shift 0 = that + W + 1; + j * 1024
    int sj0 = 16 * F0 + fr + W + 1;

    f32x4 acc[4][JW];
    bf16x8_t a0[4], a1[4], b[JW];      // 32-channel half 0 of the step; half 1 is read INSIDE the matrix segment, the
    int bad[JW];                       // pixel fragments into the registers half 0 has just released
    unsigned pmask[3] = {0u, 0u, 0u};  // bit 9 (j % 3) + t of word j / 3: tap t of fragment j stays inside the image

    int m0 = 0, n0 = 0, tm = 0;
    auto tile_coords = [&](int tile, int& tm_, int& m0_, int& n0_) {
        tm_ = tile / p.ntile_n;
        n0_ = (tile - tm_ * p.ntile_n) * 128;
        m0_ = tm_ * BM;
    };
    // per-tile lane constants: 9-bit tap validity of every pixel fragment (forward tap numbering; FLIP mirrors it)
    auto tile_setup = [&]() {
        const int w0 = m0 % W, h0 = (m0 / W) % H;          // wave-uniform
        unsigned pm[3] = {0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            const int pl = 16 * (F0 + j) + fr;
            const unsigned t = (unsigned)(w0 + pl);
            const unsigned q = (t * p.magicW) >> 16;
            const int w = (int)(t - q * W);
            const unsigned hr = (unsigned)h0 + q;
            const int h = (int)(hr - ((hr * p.magicH) >> 16) * H);
            unsigned mask = 0;
            if (pl < BM && m0 + pl < p.M) {
                const unsigned cm = (w > 0 ? 1u : 0u) | 2u | (w < W - 1 ? 4u : 0u);
                mask = (h > 0 ? cm : 0u) | (cm << 3) | (h < H - 1 ? cm << 6 : 0u);
            }
            if (FLIP) {   // tap t of the flipped direction = tap 8 - t of the forward one
                unsigned rv = 0;
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) rv |= ((mask >> t9) & 1u) << (8 - t9);
                mask = rv;
            }
            pm[j / 3] |= mask << (9 * (j % 3));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        pmask[0] = pm[0]; pmask[1] = pm[1]; pmask[2] = pm[2];
    };

    // ---- write-back from the accumulator registers -----------------------------------------------------------
    // A lane (fr, fg) holds channels 16i + 4fg .. +3 (i = 0..3) of pixel 16 (F0 + j) + fr.
    //  (1) per pair of fragments (i = 2b, 2b+1) a v_permlane16_swap between the lane rows fg = 2a and 2a+1 leaves an
    //      even row with channels 16 (2b) + 8a .. +7 and an odd row with channels 16 (2b+1) + 8a .. +7: 16 contiguous
    //      bytes per lane, i.e. pieces P0 (channels 0-31 of the wave's 64) and P1 (32-63) of the lane's pixel;
    //  (2) the two 8-lane halves of every row trade pieces (DPP row_ror:8), so that ONE store instruction carries
    //      complete 128-byte lines: pixels fr & 7 in the first, 8 + (fr & 7) in the second (lanes fr < 8 hold P0, lanes
    //      fr >= 8 hold P1).  The CU's store path takes ~16 B/clk whatever the pattern (tools/micro/store_burst.hip:
    //      16.1 B/clk for whole lines, 13.8 for half lines, 7.9 for 8-byte stores): a 100-KB tile is ~6,500 cycles of
    //      store issue, which is why the stores go FIRST and the BatchNorm sums are formed while they drain.
    auto epilogue = [&]() {
        lh_tile_writeback<BM, JW, F0, ACC, LH4_NT != 0, (LH4_DBG & 1) != 0, FLIP && !ACC>(acc, p.dst, p.acc_mask, p.stat_partials != nullptr,
                                                                            (float*)(smem + kL4OffScr), p.M, Nd, m0, n0, wn, fr, fg, p.bnb);
    };
    // B half, one segment after both halves' write-back: thread -> (q, channel); groups added in the order 0,1,2,3
    auto stat_combine = [&](int tm_, int n0_) {
        if (ACC || !p.stat_partials || wave < 4) return;
        const int t = tid & 255;
        const int q = t >> 7, ch = t & 127;
        const float* scr = (const float*)(smem + kL4OffScr);
        float s = scr[q * 128 + ch];
#pragma unroll
        for (int g = 1; g < 4; ++g) s += scr[g * 256 + q * 128 + ch];
        p.stat_partials[((long)tm_ * 2 + q) * Nd + n0_ + ch] = s;
    };

    // ---- main loop: one step = (chunk, tap); fragments of both channel halves, then 2 x JW x 4 MFMAs; one barrier per step ----
    int ring = 0, hbuf = 0;
    int prev_tm = 0, prev_n0 = 0;
    bool pending_combine = false;
    __builtin_amdgcn_s_barrier();                  // the loaders' prologue has landed
    for (int it = 0; it < tile_count; ++it) {
        tile_coords(tile_first + it, tm, m0, n0);
        tile_setup();
        for (int c = 0; c < nchunks; ++c) {
            asm volatile("" : "+v"(pmask[0]), "+v"(pmask[1]), "+v"(pmask[2]), "+v"(sj0));
            auto step = [&](auto tap_tag) {
                constexpr int tap = decltype(tap_tag)::value;
                const char* wp = smem + kL4OffW + ring * kL4WTile;
                bf16x8_t a1[4], b1[JW];
#pragma unroll
                for (int i = 0; i < 4; ++i) a0[i] = *(const bf16x8_t*)(wp + (aoff + i * 1024));
                constexpr int tr = tap / 3, ts = tap - 3 * tr;
                const int slot = sj0 + (FLIP ? (1 - tr) * W + (1 - ts) : (tr - 1) * W + (ts - 1));
                const int offt = slot * 64 + ((fg ^ l4_key(slot >> 2)) << 4) + hbuf * kL4Halo;
                const int zoff = kL4ZeroSlot * 64 + hbuf * kL4Halo;
#pragma unroll
                for (int j = 0; j < JW; ++j) {
                    bad[j] = ((pmask[j / 3] >> (9 * (j % 3) + tap)) & 1u) ? offt : zoff - j * 1024;
                    b[j] = *(const bf16x8_t*)(smem + (bad[j] + j * 1024));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = *(const bf16x8_t*)(wp + (kL4Half + aoff + i * 1024));
#pragma unroll
                for (int j = 0; j < JW; ++j) b1[j] = *(const bf16x8_t*)(smem + (bad[j] + (j * 1024 + kL4Plane)));
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
                ring = ring == 2 ? 0 : ring + 1;
                __builtin_amdgcn_s_barrier();      // everybody has read this step's weight tile (and, after tap 8, the halo)
                if (tap == 0 && c == 0 && pending_combine) {       // the previous tile's partials: parked before this barrier
                    stat_combine(prev_tm, prev_n0);
                    pending_combine = false;
                }
            };
            step(LhJ4<0>{}); step(LhJ4<1>{}); step(LhJ4<2>{}); step(LhJ4<3>{}); step(LhJ4<4>{});
            step(LhJ4<5>{}); step(LhJ4<6>{}); step(LhJ4<7>{}); step(LhJ4<8>{});
            hbuf ^= 1;
        }
        epilogue();
        prev_tm = tm; prev_n0 = n0;
        pending_combine = true;
    }
    if (!ACC && p.stat_partials) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        stat_combine(prev_tm, prev_n0);
    }
}

// Loader wave l (0..3): per step the weight tile of step s + 2 (4 of its 16 pieces) and, over the first seven taps of a chunk, its
// 8 pieces of the next chunk's halo; waits for everything older than this step's pieces, then meets the step's barrier.
template <int BM, bool FLIP, bool ACC>
__device__ __forceinline__ void lh4_loader(const Lh4Params& p, char* smem, int tile_first, int tile_count, int l) {
    const int lane = threadIdx.x & 63;
    const int W = p.W, Cs = p.Cs, Nd = p.Nd;
    const int nchunks = Cs >> 6;
    const int klen = 9 * Cs;
    const int nslots = BM + 2 * W + 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4_t rs_src = l4_rsrc(p.src, (long)p.M * Cs * 2);
    const i32x4_t rs_wt = l4_rsrc(p.wt, (long)Nd * klen * 2);
    const int hplane = l & 1, hr0 = l >> 1;
    // halo piece k (0..7) of this loader: plane hplane, slots 16 (hr0 + 2k) .. +15
    auto halo_piece = [&](int k, int hm0, int c, int buf) {
        const int hslot = (lane >> 2) + 16 * (hr0 + 2 * k);
        unsigned voff = (unsigned)(((hm0 + hslot) * Cs + (((lane & 3) ^ l4_key(lane >> 4)) << 3)) * 2);
        if (hslot >= nslots || hm0 + hslot < 0) voff = kL4Oob;
        l4_dma(voff, rs_src, __builtin_amdgcn_readfirstlane((unsigned)(c * 128 + hplane * 64)),
               __builtin_amdgcn_readfirstlane(lds0 + buf * kL4Halo + hplane * kL4Plane + (hr0 + 2 * k) * 1024));
    };
    const unsigned wvoff = (unsigned)(((lane >> 2) * klen + (((lane & 3) ^ l4_key(lane >> 4)) << 3)) * 2);
    auto wt_piece = [&](int pc, int n0_, int tap, int c, int half, int ring) {
        const unsigned soff = (unsigned)((((long)(n0_ + 16 * pc)) * klen + tap * Cs + c * 64 + half * 32) * 2);
        l4_dma(wvoff, rs_wt, __builtin_amdgcn_readfirstlane(soff),
               __builtin_amdgcn_readfirstlane(lds0 + kL4OffW + ring * kL4WTile + half * kL4Half + pc * 1024));
    };
    auto tile_coords = [&](int tile, int& m0_, int& n0_) {
        const int tm_ = tile / p.ntile_n;
        n0_ = (tile - tm_ * p.ntile_n) * 128;
        m0_ = tm_ * BM;
    };
    // cursor over the steps: (tile index in this block's range, chunk, tap)
    struct Cur { int it, c, tap; };
    auto advance = [&](Cur& q) {
        if (++q.tap == 9) { q.tap = 0; if (++q.c == nchunks) { q.c = 0; ++q.it; } }
    };
    auto stage_w = [&](const Cur& q, int ring) {      // this loader's four pieces of step q's weight tile
        if (q.it >= tile_count) return 0;
        int m0_, n0_;
        tile_coords(tile_first + q.it, m0_, n0_);
        wt_piece(l, n0_, q.tap, q.c, 0, ring);
        wt_piece(l + 4, n0_, q.tap, q.c, 0, ring);
        wt_piece(l, n0_, q.tap, q.c, 1, ring);
        wt_piece(l + 4, n0_, q.tap, q.c, 1, ring);
        return 4;
    };
    // ---- prologue: halo of the first chunk, weight tiles of steps 0 and 1 ----
    {
        int m0_, n0_;
        tile_coords(tile_first, m0_, n0_);
#pragma unroll
        for (int k = 0; k < 8; ++k) halo_piece(k, m0_ - (W + 1), 0, 0);
    }
    Cur wq{0, 0, 0};
    stage_w(wq, 0); advance(wq);
    stage_w(wq, 1); advance(wq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int ring2 = 2, hbuf = 0;                           // ring slot of step s + 2
    for (int it = 0; it < tile_count; ++it) {
        int m0, n0, nm0 = 0, nn0 = 0;
        tile_coords(tile_first + it, m0, n0);
        const bool more_tiles = it + 1 < tile_count;
        if (more_tiles) tile_coords(tile_first + it + 1, nm0, nn0);
        for (int c = 0; c < nchunks; ++c) {
            const bool last_chunk = c + 1 == nchunks;
            const bool has_next = !last_chunk || more_tiles;
            const int nx_hm0 = (last_chunk ? nm0 : m0) - (W + 1), nx_c = last_chunk ? 0 : c + 1;
            auto step = [&](auto tap_tag) {
                constexpr int tap = decltype(tap_tag)::value;
                int n = stage_w(wq, ring2);
                advance(wq);
                ring2 = ring2 == 2 ? 0 : ring2 + 1;
                if (tap <= 6 && has_next) {
                    constexpr int k0 = tap == 0 ? 0 : tap + 1, k1 = tap + 2;
#pragma unroll
                    for (int k = k0; k < k1; ++k) {
                        halo_piece(k, nx_hm0, nx_c, hbuf ^ 1);
                        ++n;
                    }
                }
                // everything older than this step's pieces has landed: the weight tile of step s + 1, earlier halo pieces
                switch (n) {
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;   // (1 or 2 halo pieces without a weight tile)
                }
                __builtin_amdgcn_s_barrier();
            };
            step(LhJ4<0>{}); step(LhJ4<1>{}); step(LhJ4<2>{}); step(LhJ4<3>{}); step(LhJ4<4>{});
            step(LhJ4<5>{}); step(LhJ4<6>{}); step(LhJ4<7>{}); step(LhJ4<8>{});
            hbuf ^= 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!ACC && p.stat_partials) __builtin_amdgcn_s_barrier();
}

template <int BM, bool FLIP, bool ACC>
__global__ __launch_bounds__(768) void conv3x3_lh4_kernel(Lh4Params p) {
    static_assert(BM == 196, "196-pixel tiles");
    constexpr int J0 = 4, J = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, nb);
    const int first = (int)(((long)lb * p.ntiles) / nb), last = (int)(((long)(lb + 1) * p.ntiles) / nb);
    const int count = last - first;
    if (count <= 0) return;
    if (wave >= 8) {
        lh4_loader<BM, FLIP, ACC>(p, smem, first, count, wave - 8);
        return;
    }
    const int wm = wave >> 1;
    if (wm == 0) lh4_run<BM, J0, 0, FLIP, ACC>(p, smem, first, count);
    else if (wm == 1) lh4_run<BM, J, J0, FLIP, ACC>(p, smem, first, count);
    else if (wm == 2) lh4_run<BM, J, J0 + J, FLIP, ACC>(p, smem, first, count);
    else lh4_run<BM, J, J0 + 2 * J, FLIP, ACC>(p, smem, first, count);
}

static int lh4_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Called by conv3x3_lh2_dispatch for the shapes that take 196-pixel tiles (option lh4); same contract as that function.
int conv3x3_lh4_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                         int accumulate, hipStream_t st, float* stat_partials, const uint8_t* acc_mask, const LhBnBwd* bnb) {
    const int bm = 196;
    if (W > 28 || W < 2 || H < 2 || Cs % 64 || Nd % 128) return PRIMIA_ERR_UNSUPPORTED;
    if (stat_partials && (accumulate || (flip != 0) != (bnb != nullptr && bnb->y != nullptr))) return PRIMIA_ERR_ARG;
    if (accumulate && !flip) return PRIMIA_ERR_UNSUPPORTED;
    const long M = (long)N * H * W;
    if (M * (Cs > Nd ? Cs : Nd) >= (1L << 30)) return PRIMIA_ERR_UNSUPPORTED;     // byte offsets stay below 2^31
    Lh4Params p;
    p.src = src; p.wt = wt; p.dst = dst;
    p.H = H; p.W = W; p.Cs = Cs; p.Nd = Nd; p.M = (int)M;
    p.acc_mask = accumulate ? acc_mask : nullptr;
    p.stat_partials = stat_partials;
    p.bnb = (bnb && bnb->y) ? *bnb : LhBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr};
    p.ntile_n = Nd / 128;
    p.ntiles = (int)((M + bm - 1) / bm) * p.ntile_n;
    p.magicW = (65536u + W - 1) / W;
    p.magicH = (65536u + H - 1) / H;
    p.prof = nullptr;
    const int ncu = lh4_num_cus();
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    void (*kern)(Lh4Params) = !flip ? conv3x3_lh4_kernel<196, false, false>
                                    : (accumulate ? conv3x3_lh4_kernel<196, true, true> : conv3x3_lh4_kernel<196, true, false>);
    const int slot = !flip ? 0 : (accumulate ? 2 : 1);
    static bool attr_set[3] = {false, false, false};
    if (!attr_set[slot]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kL4Lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set[slot] = true;
    }
    kern<<<grid, 768, kL4Lds, st>>>(p);
    return launch_status();
}

}  // namespace primia
