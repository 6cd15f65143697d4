// 3x3 / stride-1 / pad-1 convolution, forward and data gradient, wide stages (ResNet-18 layer2-4), bf16, gfx950:
// second generation of the "linear halo" implicit GEMM (conv3x3_lh.hip), rebuilt around what round 2 measured about it —
// of a 67-84 us launch only 38-40 us was the MFMA phase; the rest was per-tile skeleton (launch, first-halo latency,
// write-back) that nothing overlapped, a 77 %-full last round of tiles, and load segments longer than the MFMA
// segments they alternate with.
//
//   * PERSISTENT: one 8-wave block per CU walks a contiguous range of tiles; the LDS rings (2 halo buffers, the weight
//     rings) run straight through tile boundaries, so the next tile's first halo chunk and weights are already
//     resident when the current tile's last step ends.  Results leave from the accumulator registers (no LDS
//     staging: the rings never stop), as 16-byte stores after a v_permlane16_swap transpose.
//   * BALANCED tiles: BM = 392 pixels (two per CU on 28x28, one per CU and channel half on 14x14 at batch 256) or
//     196 pixels (7x7 images; small problems).  25 (13) pixel fragments of 16 are split 7 / 6 / 6 / 6 (4 / 3 / 3 / 3)
//     over the four pixel groups; the ping-pong halves (waves 0-3 | 4-7) then carry 13 | 12 (7 | 6) fragments, and on
//     every SIMD a matrix segment of one half runs beside the load segment of the other.
//   * the tile is 1.75x the old one at the same weight traffic: 287 flop per L2 -> LDS byte instead of 176 (the
//     matrix pipe needs 174 at its peak), and 22 fragment reads feed 56 MFMAs per wave and step (was 16 for 32).
//   * registers: only the first 32-channel half of a step's fragments is read in the load segment; the second half
//     is read INSIDE the matrix segment, each pixel fragment into the registers its first half has just released
//     (112 accumulator + 60 fragment registers for 7 fragments instead of 112 + 88).
//   * matrix segments hold nothing but MFMAs and those reads: every LDS-DMA piece is issued in a LOAD segment (its
//     issue costs the wave 100-200 cycles, which must run beside the partner's MFMAs, not in front of its own).
//
// A block owns BM CONSECUTIVE NHWC pixels x 128 output channels.  Per 64-channel chunk the pixel run plus W + 1
// pixels either side ("linear halo", <= 450 slots of 128 B) is staged ONCE for all 9 taps: tap (r, s) of output
// pixel m reads source pixel m + (r-1) W + (s-1), i.e. the same buffer at a tap-uniform slot shift; a tap that leaves
// the image reads an all-zero slot instead (9-bit validity per pixel fragment and lane, one select per fragment and
// step).  Staging is buffer-addressed LDS-DMA, so out-of-range pixels and dead slots are zero-filled by the hardware
// range check; the halo image carries the XOR chunk swizzle (chunk ^ ((slot >> 1) & 7)) on the DMA source side.
//
// Weights of a step (tap, chunk) = 128 rows x 64 channels, staged as two half tiles of 128 rows x 64 B (channels
// 0-31 | 32-63), because the halves live differently: the first is read in the load segments of step t (ring of 2:
// requested by the A waves in load(t-1)), the second in the matrix segments of step t, which end one segment later
// (ring of 3: requested by the B waves in load(t-2)).  64-byte rows: four rows span the 64 banks; chunk c of row r
// sits at c ^ l2_key(r >> 2), which keeps every lane group of a ds_read_b128 conflict-free at any row alignment.
//
// Data gradient = the same kernel on (dy, w_dgrad [C][R][S][K]) with the tap direction flipped (FLIP).
#include <stdlib.h>

#include "conv3x3_lh.h"

namespace primia {

typedef int i32x4_t __attribute__((ext_vector_type(4)));

struct Lh2Params {
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
    unsigned long long* prof;  // LH2_PROF builds: [block][wave][4] cycles in load / matrix / barrier-wait / write-back
};

// compile-time experiment switches (tools/micro/lh2_bench.hip): 1 no write-back (accumulators kept alive), 2 no DMA
// after the prologue, 4 no MFMA, 8 no fragment reads
#ifndef LH2_NT
#define LH2_NT 1
#endif
#ifndef LH2_DBG
#define LH2_DBG 0
#endif
#ifndef LH2_PRIO
#define LH2_PRIO 1
#endif
#ifndef LH2_PRIO_LOAD
#define LH2_PRIO_LOAD 0      // priority outside the matrix segments (experiment: 3 with LH2_PRIO 0 = the reverse of the default)
#endif
#ifndef LH2_SWAP
#define LH2_SWAP 0
#endif
#ifndef LH2_RF
#define LH2_RF 0
#endif
#ifdef LH2_PROF
#define LH2_MARK(slot)                                 \
    {                                                  \
        const unsigned long long t_now = clock64();    \
        prof_t[slot] += t_now - prof_prev;             \
        prof_prev = t_now;                             \
    }
#else
#define LH2_MARK(slot)
#endif

constexpr int kL2Slots = 464;                       // halo slots per buffer
constexpr int kL2Plane = kL2Slots * 64;             // 29,696 B: one 32-channel half of every slot (29 DMA pieces of 16 slots)
constexpr int kL2Halo = 2 * kL2Plane;               // 59,392 B
constexpr int kL2Half = 128 * 64;                   // 8 KiB: one 32-channel half of a step's weight tile
constexpr int kL2OffA0 = 2 * kL2Halo;               // ring of 2: first halves
constexpr int kL2OffA1 = kL2OffA0 + 2 * kL2Half;    // ring of 3: second halves
constexpr int kL2OffScr = kL2OffA1 + 3 * kL2Half;   // BatchNorm partials of the four pixel groups [4][2][128] fp32
constexpr int kL2Lds = kL2OffScr + 4 * 2 * 128 * 4; // 163,840 B = all of a CU's LDS
constexpr int kL2ZeroSlot = kL2Slots - 1;           // never live (live slots <= 450): zero-filled with every chunk
static_assert(kL2Lds <= 163840, "LDS budget");
constexpr unsigned kL2Oob = 0xfffffff0u;

// Chunk swizzle of every 64-byte row (halo slots, weight rows): 16-byte chunk c of row r sits at c ^ l2_key(r >> 2).
// A ds_read_b128 is served in four groups of 16 lanes that are NOT consecutive lanes — {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) — so a group mixes eight lanes of one channel chunk (rows 0-3 and
// 12-15 of a fragment) with eight of the next (rows 4-11).  Rows that share a bank quarter are 4 apart; the key must make
// key[g], key[g + 3], key[g + 1] ^ 1, key[g + 2] ^ 1 pairwise distinct for every g.  Round 3's table {0, 2, 3, 1} did that
// only for g = 0 and 2, i.e. for fragments starting at a slot that is a multiple of 8: every tap shift (+-1, +-W) paid a
// 2-way conflict on every pixel fragment read (SQ_LDS_BANK_CONFLICT 3.75 M of 10 M LDS cycles per launch,
// profiles/r04_stall_counters.txt).  {0, 2, 0, 2} is conflict-free at EVERY alignment.
__device__ __forceinline__ int l2_key(int quad) { return (quad & 1) << 1; }

__device__ __forceinline__ void l2_dma(unsigned voff, i32x4_t rsrc, unsigned soff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ i32x4_t l2_rsrc(const void* base, long bytes) {
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

__device__ __forceinline__ void l2_wait_vmcnt(int n) {   // wave-uniform n
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    }
}

template <int J>
struct LhJ2 {
    static constexpr int value = J;
};

// One wave's whole life.  JW = pixel fragments of this wave, F0 = its first fragment, ISA = first ping-pong half.
template <int BM, int JW, int F0, bool ISA, bool FLIP, bool ACC>
__device__ __forceinline__ void lh2_run(const Lh2Params& p, char* smem, int tile_first, int tile_count) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wh = wave & 3;           // channel half; index inside the ping-pong half
    const int fr = lane & 15, fg = lane >> 4;
    const int W = p.W, H = p.H, Cs = p.Cs, Nd = p.Nd;
    const int nchunks = Cs >> 6;
    const int klen = 9 * Cs;
    const int nslots = BM + 2 * W + 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    constexpr bool staging = !(LH2_DBG & 2);
#ifdef LH2_PROF
    unsigned long long prof_t[5] = {0, 0, 0, 0, 0}, prof_prev = clock64();
#endif

    const i32x4_t rs_src = l2_rsrc(p.src, (long)p.M * Cs * 2);
    const i32x4_t rs_wt = l2_rsrc(p.wt, (long)Nd * klen * 2);

    // ---- staging addresses ------------------------------------------------------------------------------
    // Halo image of a chunk = two planes (channels 0-31 | 32-63 of every slot, 64 B each): the second half of a pixel
    // fragment then sits at a CONSTANT distance from the first (an immediate offset of its ds_read, no address
    // arithmetic inside the matrix segment).  Piece q (0..57): plane q / 29, slots 16 (q % 29) .. +15; this lane: slot
    // + lane / 4, LDS chunk lane % 4 <- source chunk (lane % 4) ^ key(slot >> 2), and (slot >> 2) & 3 = lane >> 4 for
    // every piece, so ONE per-lane byte offset serves all pieces; the pixel part goes into the per-lane offset as well
    // (not into soffset: the range check must see it).
    // A wave stages ONE plane (wave & 1) and every second (prologue: fourth) piece of it: piece r = r0 + step * k.
    const int hplane = wave & 1;
    int hr0 = wh >> 1;                                   // main loop (B waves): r = (wh >> 1) + 2k; prologue: (wave >> 1) + 4k
    auto halo_piece = [&](int k, int step, int hm0, int c, int buf) {
        const int hslot = (lane >> 2) + 16 * hr0;        // slot of piece r0 (per-lane, loop-invariant)
        const unsigned hvbase = (unsigned)((hslot * Cs + (((lane & 3) ^ l2_key(lane >> 4)) << 3)) * 2);
        unsigned voff = hvbase + (unsigned)((hm0 + 16 * step * k) * Cs * 2);
        if (hslot + 16 * step * k >= nslots) voff = kL2Oob;
        l2_dma(voff, rs_src, __builtin_amdgcn_readfirstlane((unsigned)(c * 128 + hplane * 64)),
               __builtin_amdgcn_readfirstlane(lds0 + buf * kL2Halo + hplane * kL2Plane + (hr0 + step * k) * 1024));
    };
    // Weight half-tile piece pc (0..7): rows 16pc .. 16pc+15, this lane: row 16pc + lane/4, LDS chunk lane%4 <- source
    // chunk (lane%4) ^ key(row >> 2); (row >> 2) & 3 = lane >> 4 for every piece.
    const unsigned wvoff = (unsigned)(((lane >> 2) * klen + (((lane & 3) ^ l2_key(lane >> 4)) << 3)) * 2);
    auto wt_piece = [&](int pc, int n0_, int tap, int c, int half, unsigned lds_base) {
        const unsigned soff = (unsigned)((((long)(n0_ + 16 * pc)) * klen + tap * Cs + c * 64 + half * 32) * 2);
        l2_dma(wvoff, rs_wt, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(lds0 + lds_base + pc * 1024));
    };

    // ---- fragment read addresses ---------------------------------------------------------------------------
    // weights: row wn*64 + 16i + fr of a half tile, 16-byte chunk fg; + i * 1024
    const int aoff = (wn * 64 + fr) * 64 + ((fg ^ l2_key(fr >> 2)) << 4);
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
        lh_tile_writeback<BM, JW, F0, ACC, LH2_NT != 0, (LH2_DBG & 1) != 0, FLIP && !ACC>(acc, p.dst, p.acc_mask, p.stat_partials != nullptr,
                                                                            (float*)(smem + kL2OffScr), p.M, Nd, m0, n0, wn, fr, fg, p.bnb);
    };
    // B half, one segment after both halves' write-back: thread -> (q, channel); groups added in the order 0,1,2,3
    auto stat_combine = [&](int tm_, int n0_) {
        if (ACC || !p.stat_partials || ISA) return;
        const int t = tid & 255;
        const int q = t >> 7, ch = t & 127;
        const float* scr = (const float*)(smem + kL2OffScr);
        float s = scr[q * 128 + ch];
#pragma unroll
        for (int g = 1; g < 4; ++g) s += scr[g * 256 + q * 128 + ch];
        p.stat_partials[((long)tm_ * 2 + q) * Nd + n0_ + ch] = s;
    };

    // ---- the two kinds of segment ---------------------------------------------------------------------------
    int par = 0;        // first-half ring slot of the step being loaded (t & 1)
    int tri = 0;        // second-half ring slot of the step being loaded (t % 3)
    int tri_m = 0;      // ... of the step whose matrix segment runs next
    int hbuf = 0;       // halo buffer of the chunk being loaded
    // LH2_RF = 1 (experiment, measured SLOWER: 58.9 -> 60.6 / 51.9 -> 54.1 us, layer2 / layer3 forward): a load segment
    // requests its fragment reads first and its LDS-DMA pieces behind them.  The load segments do get shorter (B half:
    // 33 k -> 25 k cycles per launch), but the partner's matrix segments grow by as much (35.7 k -> 38.7 k) and so do the
    // barrier waits: the pieces are accepted later, land later, and their LDS writes then sit in the matrix segments' reads
    auto load_issue = [&](auto tap_tag) {
        constexpr int tap = decltype(tap_tag)::value;
        if (!(LH2_DBG & 8)) {
            const char* w0p = smem + kL2OffA0 + par * kL2Half;
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[i] = *(const bf16x8_t*)(w0p + (aoff + i * 1024));
            constexpr int tr = tap / 3, ts = tap - 3 * tr;
            const int slot = sj0 + (FLIP ? (1 - tr) * W + (1 - ts) : (tr - 1) * W + (ts - 1));
            const int offt = slot * 64 + ((fg ^ l2_key(slot >> 2)) << 4) + hbuf * kL2Halo;
            const int zoff = kL2ZeroSlot * 64 + hbuf * kL2Halo;
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                // an invalid tap reads the zero slot (minus the fragment's immediate offset)
                bad[j] = ((pmask[j / 3] >> (9 * (j % 3) + tap)) & 1u) ? offt : zoff - j * 1024;
                b[j] = *(const bf16x8_t*)(smem + (bad[j] + j * 1024));
            }
        }
    };
    auto load_finish = [&]() {
        // (the builtin, not inline asm: the compiler's own wait-count bookkeeping then knows these reads have landed and
        // puts no further waits for them between the MFMAs of the matrix segment)
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        par ^= 1;
        tri_m = tri;
        tri = tri == 2 ? 0 : tri + 1;
    };
    auto mfma_segment = [&]() {
#if LH2_PRIO || LH2_PRIO_LOAD
        __builtin_amdgcn_s_setprio(LH2_PRIO);   // the partner's load segment must not take issue slots from the MFMAs
#endif
        if (!(LH2_DBG & 8)) {
            const char* w1p = smem + kL2OffA1 + tri_m * kL2Half;
#pragma unroll
            for (int i = 0; i < 4; ++i) a1[i] = *(const bf16x8_t*)(w1p + (aoff + i * 1024));
        }
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            if (!(LH2_DBG & 4)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b[j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(LH2_DBG & 8))
                b[j] = *(const bf16x8_t*)(smem + (bad[j] + (j * 1024 + kL2Plane)));  // half 1 of this fragment, same registers
        }
        if (!(LH2_DBG & 4)) {
            // fragment order pinned: the second half of fragment 0 was requested first and has long landed when this
            // phase starts, that of fragment JW-1 arrives while the earlier groups multiply (left to itself the
            // scheduler starts with a late fragment and waits for ALL reads: lgkmcnt(0), ~150-300 exposed cycles)
#pragma unroll
            for (int j = 0; j < JW; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#if LH2_PRIO || LH2_PRIO_LOAD
        __builtin_amdgcn_s_setprio(LH2_PRIO_LOAD);
#endif
    };

    // ---- prologue: first tile's chunk 0 (all waves), first half of step 0 (A), second halves of steps 0 and 1 (B) ----
    tile_coords(tile_first, tm, m0, n0);
    {
        hr0 = wave >> 1;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (hr0 + 4 * k < kL2Slots / 16) halo_piece(k, 4, m0 - (W + 1), 0, 0);
        hr0 = wh >> 1;
        // second halves: pieces 0..3 belong to the A waves (requested one step ahead), 4..7 to the B waves (two ahead)
        if (ISA) {
            wt_piece(wh, n0, 0, 0, 0, kL2OffA0);
            wt_piece(wh + 4, n0, 0, 0, 0, kL2OffA0);
            wt_piece(wh, n0, 0, 0, 1, kL2OffA1);
        } else {
            wt_piece(wh + 4, n0, 0, 0, 1, kL2OffA1);
            // step 1: tap 1 of chunk 0 (a tile has at least 9 steps)
            wt_piece(wh + 4, n0, 1, 0, 1, kL2OffA1 + kL2Half);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // Segment numbering: segment 2t = { A: load(t) | B: mfma(t-1) }, segment 2t+1 = { A: mfma(t) | B: load(t) }, a barrier
    // after each.  Requests (all in LOAD segments, drained by the same wave at the end of its NEXT matrix segment):
    //   A, load(t): first half of step t+1 -> slot (t+1) & 1, last read by B's load(t-1) one segment earlier;
    //   B, load(t): second half of step t+2 -> slot (t+2) % 3, last read by B's mfma(t-1) one segment earlier; needed
    //               by A's mfma(t+2), four segments later;
    //   B, load(t), taps 0..6 of chunk g: the halo pieces of chunk g+1 -> buffer (g+1) & 1, last read in chunk g-1.
    int prev_tm = 0, prev_n0 = 0;
    int issued = 0;                   // B: DMA pieces requested in the latest load segment
    bool pending_combine = false;
    for (int it = 0; it < tile_count; ++it) {
        const bool more_tiles = it + 1 < tile_count;
        int ntm = 0, nm0 = 0, nn0 = 0;
        if (more_tiles) tile_coords(tile_first + it + 1, ntm, nm0, nn0);
        if (ISA) tile_setup();
        for (int c = 0; c < nchunks; ++c) {
            // (the selects of the load segments are invariant over the chunks of a tile: left alone, the compiler hoists
            // all 9 x JW of them out of this loop and keeps them in registers)
            asm volatile("" : "+v"(pmask[0]), "+v"(pmask[1]), "+v"(pmask[2]), "+v"(sj0), "+s"(hr0));
            const bool last_chunk = c + 1 == nchunks;
            // the chunk after this one (same tile, or the next tile's first)
            const bool has_next = !last_chunk || more_tiles;
            const int nx_hm0 = (last_chunk ? nm0 : m0) - (W + 1), nx_c = last_chunk ? 0 : c + 1;
            const int nx_n0 = last_chunk ? nn0 : n0;
            auto step = [&](auto tap_tag) {
                constexpr int tap = decltype(tap_tag)::value;
                if constexpr (ISA) {
                    if (LH2_RF) {
                        load_issue(tap_tag);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (staging) {      // first half of step t+1, and this half's share of its second half
                        const int t1 = tri == 2 ? 0 : tri + 1;       // (t + 1) % 3
                        if (tap < 8) {
                            wt_piece(wh, n0, tap + 1, c, 0, kL2OffA0 + (par ^ 1) * kL2Half);
                            wt_piece(wh + 4, n0, tap + 1, c, 0, kL2OffA0 + (par ^ 1) * kL2Half);
                            wt_piece(wh, n0, tap + 1, c, 1, kL2OffA1 + t1 * kL2Half);
                        } else if (has_next) {
                            wt_piece(wh, nx_n0, 0, nx_c, 0, kL2OffA0 + (par ^ 1) * kL2Half);
                            wt_piece(wh + 4, nx_n0, 0, nx_c, 0, kL2OffA0 + (par ^ 1) * kL2Half);
                            wt_piece(wh, nx_n0, 0, nx_c, 1, kL2OffA1 + t1 * kL2Half);
                        }
                    }
                    if (!LH2_RF) load_issue(tap_tag);
                    load_finish();
                    LH2_MARK(0)
                    __builtin_amdgcn_s_barrier();
                    LH2_MARK(2)
                    mfma_segment();
                    LH2_MARK(1)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    LH2_MARK(4)
                    __builtin_amdgcn_s_barrier();
                    LH2_MARK(2)
                } else {
                    // mfma of the previous step (tap - 1 of this chunk, or the last step of the previous chunk / tile)
                    const bool first_step = it == 0 && c == 0 && tap == 0;
                    if (!first_step) {
                        mfma_segment();
                        LH2_MARK(1)
                        l2_wait_vmcnt(issued);       // everything but the pieces of the latest load segment has landed
                        LH2_MARK(4)
                    }
                    if (tap == 0 && c == 0) {          // the previous step closed a tile (or nothing has run yet)
                        if (!first_step) {
                            const int km0 = m0, kn0 = n0;
                            m0 = prev_tm * BM; n0 = prev_n0;       // write-back addresses of the tile just finished
                            epilogue();
                            m0 = km0; n0 = kn0;
                            pending_combine = true;
                        }
                        tile_setup();
                        LH2_MARK(3)
                    }
                    __builtin_amdgcn_s_barrier();
                    LH2_MARK(2)
                    issued = 0;
                    if (LH2_RF) {
                        load_issue(tap_tag);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (staging) {
                        // this half's share of the second half of step t+2
                        const int t2 = tri == 0 ? 2 : tri - 1;       // (t + 2) % 3
                        if (tap < 7) {
                            wt_piece(wh + 4, n0, tap + 2, c, 1, kL2OffA1 + t2 * kL2Half);
                            ++issued;
                        } else if (has_next) {
                            wt_piece(wh + 4, nx_n0, tap - 7, nx_c, 1, kL2OffA1 + t2 * kL2Half);
                            ++issued;
                        }
                        // halo pieces of the next chunk: 3, 2, 2, 2, 2, 2, 2 over the taps 0..6
                        if (tap <= 6 && has_next) {
                            constexpr int k0 = tap == 0 ? 0 : 2 * tap + 1, k1 = 2 * tap + 3;
#pragma unroll
                            for (int k = k0; k < k1; ++k) {
                                if (hr0 + 2 * k < kL2Slots / 16) {
                                    halo_piece(k, 2, nx_hm0, nx_c, hbuf ^ 1);
                                    ++issued;
                                }
                            }
                        }
                    }
                    if (!LH2_RF) load_issue(tap_tag);
                    load_finish();
                    if (tap == 0 && c == 0 && pending_combine) {
                        stat_combine(prev_tm, prev_n0);
                        pending_combine = false;
                    }
                    LH2_MARK(0)
                    __builtin_amdgcn_s_barrier();
                    LH2_MARK(2)
                }
            };
            step(LhJ2<0>{}); step(LhJ2<1>{}); step(LhJ2<2>{}); step(LhJ2<3>{}); step(LhJ2<4>{});
            step(LhJ2<5>{}); step(LhJ2<6>{}); step(LhJ2<7>{}); step(LhJ2<8>{});
            hbuf ^= 1;
        }
        // tile boundary
        if (ISA) {
            epilogue();            // beside B's last matrix segment of this tile
            LH2_MARK(3)
        }
        prev_tm = tm; prev_n0 = n0;
        if (more_tiles) { tm = ntm; m0 = nm0; n0 = nn0; }
    }
    // drain: B's last matrix segment and write-back, then the statistics of the last tile
    if (!ISA) {
        mfma_segment();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LH2_MARK(1)
        const int km0 = m0, kn0 = n0;
        m0 = prev_tm * BM; n0 = prev_n0;
        epilogue();
        m0 = km0; n0 = kn0;
        LH2_MARK(3)
    }
    if (!ACC && p.stat_partials) {
        __syncthreads();
        stat_combine(prev_tm, prev_n0);
    }
#ifdef LH2_PROF
    if (p.prof && lane == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) p.prof[((long)blockIdx.x * 8 + wave) * 5 + k] = prof_t[k];
    }
#endif
}

// BM = 392: fragments 7 | 6 | 6 | 6; BM = 196: 4 | 3 | 3 | 3
template <int BM, bool FLIP, bool ACC>
__global__ __launch_bounds__(512) void conv3x3_lh2_kernel(Lh2Params p) {
    constexpr int J0 = BM == 392 ? 7 : 4, J = BM == 392 ? 6 : 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // contiguous tile range of this block, XCD-aware: blocks b, b+8, ... share an XCD and get neighbouring ranges
    const int nb = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, nb);
    const int first = (int)(((long)lb * p.ntiles) / nb), last = (int)(((long)(lb + 1) * p.ntiles) / nb);
    const int count = last - first;
    if (count <= 0) return;
#if LH2_SWAP
    const int wm = (wave >> 1) ^ 2;      // experiment: the YOUNGER waves (4-7) play the A role
#else
    const int wm = wave >> 1;
#endif
    if (wm == 0) lh2_run<BM, J0, 0, true, FLIP, ACC>(p, smem, first, count);
    else if (wm == 1) lh2_run<BM, J, J0, true, FLIP, ACC>(p, smem, first, count);
    else if (wm == 2) lh2_run<BM, J, J0 + J, false, FLIP, ACC>(p, smem, first, count);
    else lh2_run<BM, J, J0 + 2 * J, false, FLIP, ACC>(p, smem, first, count);
}

#ifdef LH2_PROF
unsigned long long* lh2_prof_buffer = nullptr;
#endif

static int lh2_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// tile height in pixels for a shape (0: not served).  392 when that gives every CU at least one tile, else 196.
static int lh2_bm(int N, int H, int W, int Cs, int Nd) {
    const int off = !PRIMIA_OPT(lh2), force = PRIMIA_OPT(lh2_bm);
    if (off || W > 28 || W < 2 || H < 2 || Cs % 64 || Nd % 128) return 0;
    const long M = (long)N * H * W;
    if (M * (Cs > Nd ? Cs : Nd) >= (1L << 30)) return 0;     // byte offsets stay below 2^31
    if (force == 392 || force == 196) return force;
    const long t392 = (M + 391) / 392 * (Nd / 128);
    return t392 >= lh2_num_cus() ? 392 : 196;
}

int conv3x3_lh4_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                         int accumulate, hipStream_t st, float* stat_partials, const uint8_t* acc_mask, const LhBnBwd* bnb);

int conv3x3_lh2_tiles_m(int N, int H, int W, int Cs, int Nd) {
    const int bm = lh2_bm(N, H, W, Cs, Nd);
    if (!bm) return PRIMIA_ERR_UNSUPPORTED;
    return (int)(((long)N * H * W + bm - 1) / bm);
}

// 4: conv3x3_lh2_kernel serves the shape, 6: conv3x3_lh4_kernel does (196-pixel tiles, option lh4), 0: neither
int conv3x3_lh_kernel_of(int N, int H, int W, int Cs, int Nd) {
    const int bm = lh2_bm(N, H, W, Cs, Nd);
    if (!bm) return 0;
    return bm == 196 && PRIMIA_OPT(lh4) ? 6 : 4;
}

int conv3x3_lh2_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                         int accumulate, hipStream_t st, float* stat_partials, const uint8_t* acc_mask, const LhBnBwd* bnb) {
    const int bm = lh2_bm(N, H, W, Cs, Nd);
    if (!bm) return PRIMIA_ERR_UNSUPPORTED;
    // partial sums: forward statistics (plain forward launch), or a BatchNorm's backward sums (plain data-gradient launch + bnb)
    if (stat_partials && (accumulate || (flip != 0) != (bnb != nullptr && bnb->y != nullptr))) return PRIMIA_ERR_ARG;
    if (bnb && bnb->y && !stat_partials) return PRIMIA_ERR_ARG;
    if (accumulate && !flip) return PRIMIA_ERR_UNSUPPORTED;
    const long M = (long)N * H * W;
    Lh2Params p;
    p.src = src; p.wt = wt; p.dst = dst;
    p.H = H; p.W = W; p.Cs = Cs; p.Nd = Nd; p.M = (int)M;
    p.acc_mask = accumulate ? acc_mask : nullptr;
    p.stat_partials = stat_partials;
    p.bnb = (bnb && bnb->y) ? *bnb : LhBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr};
    p.ntile_n = Nd / 128;
    p.ntiles = (int)((M + bm - 1) / bm) * p.ntile_n;
    p.magicW = (65536u + W - 1) / W;
    p.magicH = (65536u + H - 1) / H;
#ifdef LH2_PROF
    p.prof = lh2_prof_buffer;
#else
    p.prof = nullptr;
#endif
    if (bm == 196 && PRIMIA_OPT(lh4))
        return conv3x3_lh4_dispatch(src, wt, dst, N, H, W, Cs, Nd, flip, accumulate, st, stat_partials, acc_mask, bnb);
    const int ncu = lh2_num_cus();
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    void (*kern)(Lh2Params);
    int slot;
    if (bm == 392) {
        kern = !flip ? conv3x3_lh2_kernel<392, false, false> : (accumulate ? conv3x3_lh2_kernel<392, true, true> : conv3x3_lh2_kernel<392, true, false>);
        slot = !flip ? 0 : (accumulate ? 2 : 1);
    } else {
        kern = !flip ? conv3x3_lh2_kernel<196, false, false> : (accumulate ? conv3x3_lh2_kernel<196, true, true> : conv3x3_lh2_kernel<196, true, false>);
        slot = 3 + (!flip ? 0 : (accumulate ? 2 : 1));
    }
    static bool attr_set[6] = {false, false, false, false, false, false};
    if (!attr_set[slot]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kL2Lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set[slot] = true;
    }
    kern<<<grid, 512, kL2Lds, st>>>(p);
    return launch_status();
}

}  // namespace primia
