// 3x3 / stride-1 / pad-1 convolution with 64 input and 64 output channels (ResNet-18 layer1: forward
// and data gradient), bf16, gfx950 — weight-stationary halo-patch kernel.
//
// The generic implicit GEMM (conv_igemm.hip) re-stages a 128-pixel activation tile for every one
// of the 9 taps: each activation travels L2 -> LDS nine times and the K = 64 tile leaves only 2x2
// fragments per wave.  For C = K = 64 the whole filter (64 x 576 bf16 = 72 KiB) fits in the
// REGISTERS of one block (4 waves; two such blocks share a CU):
//     wave (kh, ph):  out-channels 32*kh .. +31  x  all 576 reduction elements  = 144 VGPRs
// so the block is persistent, loads its weights once, and then streams 8x8 output patches:
//     stage  : the patch's 10x10 halo x 64 channels (12.5 KiB) by LDS-DMA, ONCE for all 9 taps
//     compute: wave (kh, ph) owns patch rows 4*ph .. 4*ph+3 (two 16-pixel MFMA column blocks);
//              per tap and channel half ONE ds_read_b128 feeds two MFMAs (A = weights in registers)
//     store  : 4 consecutive out-channels per lane, straight from the accumulators.
// L2 -> LDS traffic is 1.56x the activation tensor instead of 9x, LDS reads are 1 per 2 MFMAs and
// there is no weight traffic at all inside the loop.
//
// Data gradient = the same kernel on (dy, w_dgrad): the w_dgrad layout keeps the forward tap order,
// the source pixel of tap (r, s) is (h + 1 - r, w + 1 - s), i.e. halo slot (2 - r, 2 - s) (`flip`).
//
// LDS layout of a halo: slot = hy*10 + hx (pixel), 128 B per slot = 8 chunks of 16 B; chunk c of a
// slot is stored at chunk (c ^ key(slot)), key from a 10-entry table (period 20 slots) found by
// exhaustive search: with it every ds_read_b128 of the compute phase (16 pixels of two patch rows x
// 4 chunks, any tap shift) is bank-conflict free.  The permutation is applied on the SOURCE side of
// the DMA (lane-linear destination).
#include <stdlib.h>

#include "conv3x3_lh.h"

// timing-experiment bits (option c64_dbg; results are WRONG when set): compiled in by probe builds only (-DPRIMIA_PROBE=1,
// tools/micro/c64_probe.sh) — the shipped kernel carries no p.debug branch and primia_set_option refuses the option
#ifndef PRIMIA_PROBE
#define PRIMIA_PROBE 0
#endif
#if PRIMIA_PROBE
#define C64_DBG(bit) (p.debug & (bit))
#else
#define C64_DBG(bit) 0
#endif

namespace primia {

__device__ __attribute__((aligned(16))) const unsigned char kC64ZeroPage[16] = {0};

// see conv_wgrad_patch.hip: inline-asm LDS-DMA keeps the compiler from draining vmcnt before LDS reads
__device__ __forceinline__ void c64_dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ int c64_key(int slot) {
    // table {0,1,2,4,1,5,6,2,6,4}[(slot >> 1) % 10], 3 bits per entry
    return (0x265a9888u >> (3 * ((slot >> 1) % 10))) & 7;
}

// C64_PROBE (experiment builds only: hipcc -DC64_PROBE=1, tools/micro/c64_probe.sh): per-wave cycles (s_memtime) spent in the
// phases of an iteration — [top wait | barrier | rows + requests | stores + sums | matrix loop | tail], summed over the block's
// patches, + the iteration count; tools/micro/c64_probe.py reads them.  s_memtime returns through lgkmcnt, so every probe also
// drains the wave's LDS queue.
#ifndef C64_PROBE
#define C64_PROBE 0
#endif
#if C64_PROBE
__device__ unsigned long long c64_probe_buf[1024 * 4 * 16];
#define C64_T(i)                                                      \
    {                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        tacc[i] += now_ - tlast;                                      \
        tlast = now_;                                                 \
    }
#else
#define C64_T(i)
#endif

typedef float c64_f32x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void c64_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform n: the two values the steady state takes first (A, B: compile-time), then the rest
template <int A, int B>
__device__ __forceinline__ void c64_wait_vm_pick(int n) {
    if (n == A) c64_wait_vm<A>();
    else if (n == B) c64_wait_vm<B>();
    else {
        switch (n) {
            case 2: c64_wait_vm<2>(); break;
            case 3: c64_wait_vm<3>(); break;
            case 4: c64_wait_vm<4>(); break;
            case 5: c64_wait_vm<5>(); break;
            case 6: c64_wait_vm<6>(); break;
            case 7: c64_wait_vm<7>(); break;
            case 8: c64_wait_vm<8>(); break;
            case 9: c64_wait_vm<9>(); break;
            case 10: c64_wait_vm<10>(); break;
            case 11: c64_wait_vm<11>(); break;
            case 12: c64_wait_vm<12>(); break;
            case 14: c64_wait_vm<14>(); break;
            case 16: c64_wait_vm<16>(); break;
            default: c64_wait_vm<0>(); break;
        }
    }
}

struct C64Params {
    const bf16* src;
    const bf16* wt;   // [64][576] forward-layout weights (fwd: w_fwd, dgrad: w_dgrad)
    bf16* dst;
    int N, H, W;
    int flip;         // 0 forward, 1 data gradient
    int accumulate;   // dst += result
    const uint8_t* acc_mask;  // accumulate form: bit i of byte (pixel*64 + 8*chunk)/8 keeps the old channel 8*chunk + i
    int PH, PW, PPI;  // 8x8 patches per image column / row / image
    int total;        // patches overall
    int per_block;    // patches per block
    float* stat_partials;  // if set: per-block BatchNorm partial sums [grid][2][64] of the values AS STORED — or, with bnb_y (plain
                           // data gradient, round 5), of the BatchNorm BACKWARD of the layer in front: sum g, sum g * xhat with
                           // g = dz AS STORED * [bn(y) > 0] (the reduction pass over (y, dz) is dropped)
    const bf16* bnb_y;     // [N*H*W][64]: that BatchNorm's input (null: forward statistics)
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_gamma;
    const float* bnb_beta;
    const uint8_t* bnb_mask;   // BNB = 2: the ReLU-mask bytes of the layer whose sums are formed (one per pixel and 8 channels)
    int debug;        // timing experiments only (option c64_dbg, 0 otherwise): 1 no stores, 2 no staging, 4 no MFMA loop,
                      // 8 no BatchNorm-sum arithmetic, 16 no BatchNorm-row prefetch, 32 no mask-word loads, 64 no old-row prefetch
};

// Block = 4 waves (256 threads), TWO blocks per CU: the two waves of a SIMD belong to different blocks, so they are
// never in the same phase — one block's barrier, write-back, DMA issue and first-fragment latency run beside the other
// block's MFMAs.  (The 8-wave form ran every wave of the CU through those phases in lockstep: with staging and stores
// switched off it still took 60 us for 32 us worth of MFMAs.)  Wave (kh, ph): out-channels 32*kh..+31, patch rows
// 4*ph..4*ph+3 = two 16-pixel MFMA column blocks q = 0, 1; one 8x8 patch per stage.
// BNB: the write-back also forms the two backward sums of a BatchNorm whose OUTPUT gradient this launch writes (sum g, sum g * xhat;
// g = the value AS STORED where the layer's ReLU passed):
//   1  plain data gradient; ReLU recomputed: [fma(y - mean, invstd * gamma, beta) > 0]          (bn1 of a block, conv2's data gradient)
//   2  accumulate form; ReLU from the stored mask bytes; xhat = (y - mean) * invstd             (residual bn2 in front of an identity block)
//   3  accumulate form; [p > 0]; xhat = (p - beta) / gamma from the POOLED activation p         (the stem's bn1, seen through the max-pool:
//      a window's gradient reaches exactly its argmax, whose activation is the pooled value — PoolScatterFn of bn.hip)
// EXACT (round 6): H and W multiples of 8 (layer1 at every input size that is a multiple of 32).  Every patch is whole, so a
// pixel's liveness is the patch's (a scalar), and its address is the patch origin (one scalar per cursor, advanced by
// additions) + a per-lane constant computed once per block: the side streams' per-patch address arithmetic (~110 vector and
// ~70 scalar instructions per patch in the accumulate + sums forms, profiles/r06_isa_budget.txt) shrinks to an add and a
// select per request; the mask words travel through buffer descriptors like the rows (no 64-bit pointer arithmetic).
template <bool ACC, int STAGES = 3, int BNB = 0, bool EXACT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_c64_kernel(C64Params p) {
    static_assert(ACC ? BNB != 1 : BNB < 2, "1: plain data gradient; 2, 3: accumulate form");
    constexpr int HALO = 13 * 1024;        // 100 slots used, 104 staged (13 DMA instructions)
    constexpr int STAGE = HALO;            // one patch per stage
    // output rows of one stage: 64 pixels x 64 channels, bf16.  (The accumulate form used to stage fp32 rows and add the old
    // values in the write-back lanes: 32 KiB of LDS and 40 KiB of LDS traffic per patch.  It now adds them to the ACCUMULATORS
    // — the old rows arrive by LDS-DMA in a layout the accumulator lanes read without bank conflicts — rounds once, and
    // shares the plain form's bf16 write-back: 63 KiB per block, which is what lets the BatchNorm rows in beside them.)
    constexpr int OPIX = 128;              // bytes per staged pixel row
    constexpr int OUTB = 64 * OPIX;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES * STAGE + 2 * OUTB (+ 8 KiB old rows: ACC) (+ 9 KiB: BNB)
    char* const sout = smem + STAGES * STAGE;
    // accumulate form: the OLD rows of the patch being computed.  Wave (kh, ph) fetches and reads exactly its own part — patch
    // rows 4 * ph .. + 3 x channel half kh = two DMA instructions of 16 pixels x 64 B — so no barrier stands between the DMA
    // and the read; LDS position (pixel, 16-B slot c) holds chunk c ^ ((pixel >> 1) & 3): conflict-free ds_read_b64
    char* const sold = sout + 2 * OUTB;
    // BNB: the BatchNorm input rows of the patch being written back (row group g at g * 1 KiB, lane-linear: pixel lane / 8,
    // 16-B chunk lane % 8), then 1 KiB of per-channel constants
    char* const saux = sold + (ACC ? 8192 : 0);
    // the ReLU-mask words of both side streams also travel by LDS-DMA (a dword per lane, [wave][q][64 lanes]): an inline-asm
    // load into a REGISTER that is consumed an iteration later is a loop-carried value the compiler may copy right after the
    // load was issued — before its data arrived (seen: v_mov of the two mask registers at the loop latch, wrong masks once a
    // block ran more than a few patches)
    constexpr int SMK_OFF = STAGES * STAGE + 2 * OUTB + (ACC ? 8192 : 0) + (BNB ? 8192 + 1024 : 0);
    char* const smk = smem + SMK_OFF;                 // accumulate form: 1 KiB ([wave][64 lanes])
    char* const sbm = smk + (ACC ? 1024 : 0);         // BNB = 2: 1 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 1, ph = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    const int t0 = blockIdx.x * p.per_block;
    int t1 = t0 + p.per_block;
    if (t1 > p.total) t1 = p.total;
    const int nstages = t1 - t0;
    if (nstages <= 0) return;
#if C64_PROBE
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // 6 .. 8: the accumulate form's tail in parts
    unsigned long long tlast = 0;
#endif

    // ---- weights -> registers: A fragment (i, j): row 32*kh + 16*i + fr, elements j*32 + 8*fg .. +7 -------
    bf16x8_t wreg[18][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16* wrow = p.wt + (long)(32 * kh + 16 * i + fr) * 576 + 8 * fg;
#pragma unroll
        for (int j = 0; j < 18; ++j) wreg[j][i] = *(const bf16x8_t*)(wrow + j * 32);
    }
    // (no wait here: the first patches' halo, BatchNorm and old rows are requested behind these loads and ONE wait below covers
    // both — the block's prologue is one memory round trip, not two; round 6)

    // patch cursor: (n, ph, pw) of consecutive patch ids, advanced by one
    struct Cursor {
        int n, ph, pw, t;
        int org;      // EXACT: pixel index of the patch origin, (n * H + 8 ph) * W + 8 pw
    };
    auto make_cursor = [&](int t) {
        Cursor c;
        c.t = t;
        c.n = t / p.PPI;
        const int rem = t - c.n * p.PPI;
        c.ph = rem / p.PW;
        c.pw = rem - c.ph * p.PW;
        c.org = (c.n * p.H + c.ph * 8) * p.W + c.pw * 8;
        return c;
    };
    auto advance = [&](Cursor& c) {
        ++c.t;
        if (++c.pw == p.PW) {
            c.pw = 0;
            c.org += 7 * p.W + 8;     // (EXACT: 8 PW = W; the next image follows the last patch row: 8 PH = H)
            if (++c.ph == p.PH) {
                c.ph = 0;
                ++c.n;
            }
        } else {
            c.org += 8;
        }
    };
    Cursor cs = make_cursor(t0), cw = make_cursor(t0);  // staging / write-back

    // ---- staging: 13 DMA instructions per stage, round-robin over the 4 waves (wave 0 issues 4, the others 3) ----
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // Buffer-addressed LDS-DMA: per piece the lane's byte offset from the patch origin (rel) and one validity bit for
    // its halo row and one for its halo column are lane constants; per stage a piece costs and + compare + add +
    // select — the first version unpacked (hy, hx), compared both against the image, built a 64-bit pointer and
    // selected a zero page (~15 vector instructions per piece; the kernel issues ~400 instructions per wave and stage
    // for 72 MFMAs).  Out-of-image / dead lanes get an offset beyond num_records: the hardware returns zeros.
    typedef int c64_i32x4 __attribute__((ext_vector_type(4)));
    c64_i32x4 rsrc;
    {
        const unsigned long long a = (unsigned long long)p.src;
        const long bytes = (long)p.N * p.H * p.W * 128;
        rsrc[0] = (int)(unsigned)a;
        rsrc[1] = (int)(unsigned)(a >> 32) & 0xffff;
        rsrc[2] = (int)(unsigned)(bytes > 0xfffffff0L ? 0xfffffff0L : bytes);
        rsrc[3] = 0x00020000;
#pragma unroll
        for (int j = 0; j < 4; ++j) rsrc[j] = __builtin_amdgcn_readfirstlane(rsrc[j]);
    }
    constexpr unsigned kOob = 0xfffffff0u;
    int rel[4];
    unsigned lbits[4];      // bit hy: halo row hy (image row rb + hy - 1), bit 16 + hx: halo column hx; pad slots: bit 31
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = (wave + 4 * it) * 8 + (lane >> 3);
        const int hy = slot / 10, hx = slot - hy * 10;       // hy >= 10: pad slot (never valid)
        const int chunk = (lane & 7) ^ c64_key(slot);
        rel[it] = (((hy - 1) * p.W + (hx - 1)) * 64 + chunk * 8) * 2;
        lbits[it] = hy < 10 ? (1u << hy) | (1u << (16 + hx)) : 0x80000000u;
    }
    auto stage = [&](int buf) {
        const bool live = cs.t < t1;
        int org;
        unsigned rowm, colm;
        if constexpr (EXACT) {      // whole patches: only the image's first / last patch row and column lose a halo line
            org = cs.org * 128;
            rowm = (cs.ph + 1 == p.PH ? 0x1ffu : 0x3ffu) & ~(cs.ph == 0 ? 1u : 0u);
            colm = (cs.pw + 1 == p.PW ? 0x1ffu : 0x3ffu) & ~(cs.pw == 0 ? 1u : 0u);
        } else {
            const int rb = cs.ph * 8, cb = cs.pw * 8;
            org = ((cs.n * p.H + rb) * p.W + cb) * 128;
            // valid halo rows hy: 0 <= rb + hy - 1 < H, columns likewise
            int rhi = p.H - rb + 1, chi = p.W - cb + 1;
            rhi = rhi > 10 ? 10 : rhi;
            chi = chi > 10 ? 10 : chi;
            rowm = ((1u << rhi) - 1u) & ~(rb == 0 ? 1u : 0u);
            colm = ((1u << chi) - 1u) & ~(cb == 0 ? 1u : 0u);
        }
        const unsigned m = live ? rowm | (colm << 16) : 0u;
        advance(cs);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = wave + 4 * it;  // wave-uniform
            if (idx >= 13) break;
            const unsigned voff = (lbits[it] & m) == lbits[it] ? (unsigned)(rel[it] + org) : kOob;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + idx * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                         ::"v"(voff), "s"(rsrc), "s"(m0v) : "memory");
        }
    };

    // ---- per-lane LDS offsets of the B fragments: column block q: pixel (4*ph + 2*q + (fr >> 3), fr & 7) --------
    // (column block 1 is two halo rows = 20 slots further: the key table has period 20 slots, so its offsets are
    // those of block 0 + 2560 bytes — an immediate, not nine more registers)
    int offb[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int r = t / 3, s = t - 3 * r;
        const int rr = p.flip ? 2 - r : r, ss = p.flip ? 2 - s : s;
        const int slot = (4 * ph + (fr >> 3) + rr) * 10 + (fr & 7) + ss;
        offb[t] = slot * 128 + ((fg ^ c64_key(slot)) << 4);   // channel half 1: offb ^ 64
    }
    // output staging: pixel opix = 32*ph + 16*q + fr of the patch, 8-byte column 8*kh + 4*i + fg of its 128-B row,
    // stored at 16-B chunk (col >> 1) ^ ((opix >> 1) & 7): conflict-free ds_write_b64 / ds_read_b128

    // ---- side streams of the write-back ------------------------------------------------------------------------------
    // Accumulate form: the old rows (and their ReLU-mask bytes) of patch s are requested at the END of compute(s - 1) and
    // added to the accumulators at the end of compute(s): a whole iteration of latency, one 8-KiB buffer, and the issuing
    // wave is the reading wave.  (History: loaded inside the write-back, a full memory latency per stage stood in front of
    // the stores and the compiler's conservative vmcnt(0) drained the DMA ring on top: 93 us against 58 for the plain form.)
    // The mask words travel by an inline-asm load the compiler's wait counting does not see.
    // BNB: the BatchNorm rows of patch s are requested in iteration s (after the write-back of patch s - 1 has read the
    // buffer) and read by the write-back of patch s in iteration s + 1 — older than the halo pieces requested after them,
    // so the counted wait at the top of the loop covers them.
    constexpr int NOLD = ACC ? 3 : 0;                          // vector-memory loads of one prefetch_old()
    constexpr int NAUX = BNB ? (BNB == 2 ? 3 : 2) : 0;         // ... of one prefetch_aux()
    c64_i32x4 rsrc_aux = rsrc;
    // (the output rows leave through a buffer descriptor too: scalar base + 32-bit lane offset, out-of-image lanes get an offset
    // beyond num_records and are dropped — no 64-bit pointers, no exec-mask branch around the store; the accumulate form's old
    // rows come in through the same descriptor)
    c64_i32x4 rsrc_st = rsrc;
    {
        const unsigned long long a = (unsigned long long)(const void*)p.dst;
        rsrc_st[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rsrc_st[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    }
    if constexpr (BNB != 0) {
        const unsigned long long a = (unsigned long long)(const void*)p.bnb_y;
        rsrc_aux[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rsrc_aux[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    }
    // EXACT: the mask bytes through descriptors too, and the per-lane parts of every side-stream address, once per block
    c64_i32x4 rsrc_mk = rsrc, rsrc_bm = rsrc;
    unsigned old_off[2] = {0u, 0u}, aux_off[2] = {0u, 0u}, mk_off = 0u, bm_off = 0u;
    if constexpr (EXACT) {
        const long mbytes = (long)p.N * p.H * p.W * 8;
        const int nrec = (int)(unsigned)(mbytes > 0xfffffff0L ? 0xfffffff0L : mbytes);
        if constexpr (ACC) {
            const unsigned long long a = (unsigned long long)(const void*)(p.acc_mask ? (const void*)p.acc_mask : (const void*)p.wt);
            rsrc_mk[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            rsrc_mk[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
            rsrc_mk[2] = __builtin_amdgcn_readfirstlane(p.acc_mask ? nrec : 0);     // (no mask: every lane out of range)
            const int pxl = lane >> 2, cc = lane & 3;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                old_off[q] = (unsigned)((((4 * ph + 2 * q + (pxl >> 3)) * p.W + (pxl & 7)) * 64 + (4 * kh + (cc ^ ((pxl >> 1) & 3))) * 8) * 2);
            const int qm = (lane >> 4) & 1, f = lane & 15;
            mk_off = (unsigned)(((4 * ph + 2 * qm + (f >> 3)) * p.W + (f & 7)) * 8 + 4 * kh);
        }
        {
            const int px = lane >> 3, c16 = lane & 7;
#pragma unroll
            for (int q = 0; q < 2; ++q) aux_off[q] = (unsigned)((((2 * wave + q) * p.W + px) * 64 + c16 * 8) * 2);
        }
        if constexpr (BNB == 2) {
            const unsigned long long a = (unsigned long long)(const void*)p.bnb_mask;
            rsrc_bm[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
            rsrc_bm[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
            rsrc_bm[2] = __builtin_amdgcn_readfirstlane(nrec);
            const int q = lane >> 5, pxm = (lane >> 2) & 7, hw = (lane >> 1) & 1;
            bm_off = (unsigned)(((2 * wave + q) * p.W + pxm) * 8 + 4 * hw);
        }
    }
    // BNB: per-channel constants in LDS behind the row buffer.  1: [mean | invstd * gamma | beta][64]; 2: [mean]; 3: [beta]
    float* const bnc = (float*)(saux + 8192);
    if constexpr (BNB != 0) {
        if (tid < 64) {
            if constexpr (BNB == 1) {
                bnc[tid] = p.bnb_mean[tid];
                bnc[64 + tid] = p.bnb_invstd[tid] * p.bnb_gamma[tid];
                bnc[128 + tid] = p.bnb_beta[tid];
            } else {
                bnc[tid] = BNB == 2 ? p.bnb_mean[tid] : p.bnb_beta[tid];
            }
        }
    }
    Cursor co = make_cursor(t0);       // patch whose old rows are requested next
    auto prefetch_old = [&]() {
        if constexpr (ACC && EXACT) {
            const bool live = co.t < t1;
            const unsigned base = (unsigned)co.org * 128u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned voff = live && !C64_DBG(64) ? base + old_off[q] : kOob;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + STAGES * STAGE + 2 * OUTB + (wave * 2 + q) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(voff), "s"(rsrc_st), "s"(m0v) : "memory");
            }
            {   // the mask words of the accumulator pixels: one dword per lane (see below), descriptor-addressed
                const unsigned voff = live ? (unsigned)co.org * 8u + mk_off : kOob;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + SMK_OFF + wave * 256);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds"
                             ::"v"(voff), "s"(rsrc_mk), "s"(m0v) : "memory");
            }
            advance(co);
        } else if constexpr (ACC) {
            const int pxl = lane >> 2, cc = lane & 3;     // pixel of the 16 (two patch rows), 16-B slot of the 64-B half row
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ho = co.ph * 8 + 4 * ph + 2 * q + (pxl >> 3), wo = co.pw * 8 + (pxl & 7);
                const bool live = co.t < t1 && ho < p.H && wo < p.W && !C64_DBG(64);
                const unsigned pix = (unsigned)((co.n * p.H + ho) * p.W + wo);
                const unsigned voff = live ? (pix * 64u + (unsigned)(4 * kh + (cc ^ ((pxl >> 1) & 3))) * 8u) * 2u : kOob;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + STAGES * STAGE + 2 * OUTB + (wave * 2 + q) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(voff), "s"(rsrc_st), "s"(m0v) : "memory");
            }
            // the mask words of the accumulator pixels (the write order above is by pixel lane / 4; the READ is by fr): ONE
            // instruction for both column blocks — lane l fetches the word of (q = (l >> 4) & 1, pixel l & 15); an LDS-DMA
            // instruction costs ~300 cycles of issue in this phase whatever it moves (tools/micro/c64_probe.py)
            {
                const int q = (lane >> 4) & 1, f = lane & 15;
                const int ho = co.ph * 8 + 4 * ph + 2 * q + (f >> 3), wo = co.pw * 8 + (f & 7);
                const bool live = co.t < t1 && ho < p.H && wo < p.W;
                const unsigned pix = (unsigned)((co.n * p.H + ho) * p.W + wo);
                // (no mask: the load is issued all the same — the wait counts below are per prefetch — and ignored)
                const uint8_t* mp = p.acc_mask ? p.acc_mask + (live ? pix * 8u + 4u * kh : 0u) : (const uint8_t*)p.wt;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + SMK_OFF + wave * 256);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(mp), "s"(m0v) : "memory");
            }
            advance(co);
        }
    };
    auto prefetch_aux = [&]() {
        if (BNB != 0 && EXACT && !C64_DBG(32)) {
            const bool live = cw.t < t1 && !C64_DBG(16);
            const unsigned base = (unsigned)cw.org * 128u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned voff = live ? base + aux_off[q] : kOob;
                const unsigned m0v =
                    __builtin_amdgcn_readfirstlane(lds0 + STAGES * STAGE + 2 * OUTB + (ACC ? 8192 : 0) + (2 * wave + q) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(voff), "s"(rsrc_aux), "s"(m0v) : "memory");
            }
            if constexpr (BNB == 2) {
                const unsigned voff = live ? (unsigned)cw.org * 8u + bm_off : kOob;
                const unsigned m1v = __builtin_amdgcn_readfirstlane(lds0 + SMK_OFF + (ACC ? 1024 : 0) + wave * 256);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds"
                             ::"v"(voff), "s"(rsrc_bm), "s"(m1v) : "memory");
            }
        } else if (BNB != 0 && !C64_DBG(32)) {
            const int px = lane >> 3, c16 = lane & 7;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int g = 2 * wave + q;
                const int ho = cw.ph * 8 + g, wo = cw.pw * 8 + px;
                const bool live = cw.t < t1 && ho < p.H && wo < p.W && !C64_DBG(16);
                const unsigned eoff = (unsigned)(((cw.n * p.H + ho) * p.W + wo) * 64 + c16 * 8);   // elements (< 2^31)
                const unsigned voff = live ? eoff * 2u : kOob;
                const unsigned m0v =
                    __builtin_amdgcn_readfirstlane(lds0 + STAGES * STAGE + 2 * OUTB + (ACC ? 8192 : 0) + g * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(voff), "s"(rsrc_aux), "s"(m0v) : "memory");
            }
            if constexpr (BNB == 2) {     // the dwords that hold the wave's mask bytes (pixel * 8 + chunk): ONE instruction for both
                                          // row groups — lane l: row group l >> 5, pixel (l >> 2) & 7, word (l >> 1) & 1
                const int q = lane >> 5, pxm = (lane >> 2) & 7, hw = (lane >> 1) & 1;
                const int ho = cw.ph * 8 + 2 * wave + q, wo = cw.pw * 8 + pxm;
                const bool live = cw.t < t1 && ho < p.H && wo < p.W && !C64_DBG(16);
                const unsigned pix = (unsigned)((cw.n * p.H + ho) * p.W + wo);
                const uint8_t* mp = p.bnb_mask + (live ? pix * 8u + 4u * hw : 0u);
                const unsigned m1v = __builtin_amdgcn_readfirstlane(lds0 + SMK_OFF + (ACC ? 1024 : 0) + wave * 256);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(mp), "s"(m1v) : "memory");
            }
        }
    };

    const bool exact = EXACT || ((p.H % 8 == 0) && (p.W % 8 == 0) && !C64_DBG(3));
    const int dstage = wave == 0 ? 4 : 3;      // halo pieces per stage issued by this wave

    // `younger` = vector-memory loads this wave issued after the old rows of the patch being computed (0: unknown, drain)
    auto compute = [&](int buf, int obuf, int younger) {
        const char* sb = smem + buf * STAGE;
        f32x4 acc[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // B fragments: 3-deep register ring, reads run two steps ahead of the MFMAs that consume them (4-deep: measured, no
        // change — the matrix loop does not wait for its fragments)
        bf16x8_t bq[3][2];
        auto rd = [&](int step, int slot) {
            const int x = (step & 1) * 64;
            const char* q0 = sb + (offb[step >> 1] ^ x);
            bq[slot][0] = *(const bf16x8_t*)q0;
            bq[slot][1] = *(const bf16x8_t*)(q0 + 20 * 128);
        };
        if (!C64_DBG(4)) {
            rd(0, 0);
            rd(1, 1);
#pragma unroll
            for (int step = 0; step < 18; ++step) {
                if (step + 2 < 18) rd(step + 2, (step + 2) % 3);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[step][i], bq[step % 3][0], acc[0][i], 0, 0, 0);
                    acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[step][i], bq[step % 3][1], acc[1][i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#if C64_PROBE
        {
            const unsigned long long now_ = __builtin_readcyclecounter();
            tacc[4] += now_ - tlast;
            tlast = now_;
        }
#endif
        if constexpr (ACC) {
            // the old rows of this patch have landed once only the loads issued after them remain in flight (loads retire
            // in order; stores do not count against them: see the loop's wait below)
            c64_wait_vm_pick<NAUX + 3, NAUX + 4>(younger);
            C64_T(6)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const char* ob = sold + (wave * 2 + q) * 1024 + fr * 64 + (fg & 1) * 8;
                // ReLU-mask bytes of the wave's channel half (4 chunks) for accumulator pixel fr
                const unsigned mkq = p.acc_mask ? *(const unsigned*)(smk + wave * 256 + (q * 16 + fr) * 4) : 0xffffffffu;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c4 = 2 * i + (fg >> 1);          // 16-B chunk of the wave's channel half
                    const u32x2 o = *(const u32x2*)(ob + ((c4 ^ ((fr >> 1) & 3)) << 4));
                    // old value = gradient through a ReLU whose mask is applied here, not stored: bit -> all-ones / zero word
                    // (v_bfe_i32) ANDed onto the value's bits — two instructions per element where test + compare + select were three
                    const int m = (int)(mkq >> (8 * c4 + 4 * (fg & 1)));
                    acc[q][i][0] += __uint_as_float((o[0] << 16) & (unsigned)__builtin_amdgcn_sbfe(m, 0, 1));
                    acc[q][i][1] += __uint_as_float((o[0] & 0xffff0000u) & (unsigned)__builtin_amdgcn_sbfe(m, 1, 1));
                    acc[q][i][2] += __uint_as_float((o[1] << 16) & (unsigned)__builtin_amdgcn_sbfe(m, 2, 1));
                    acc[q][i][3] += __uint_as_float((o[1] & 0xffff0000u) & (unsigned)__builtin_amdgcn_sbfe(m, 3, 1));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next patch's rows may land
            C64_T(7)
            prefetch_old();
            C64_T(8)
        }
        // ---- results -> LDS rows (bf16, ONE rounding) -----------------------------------------------------------
        // pixel opix = 32*ph + 16*q + fr of the patch, 8-byte column 8*kh + 4*i + fg of its 128-B row,
        // stored at 16-B chunk (col >> 1) ^ ((opix >> 1) & 7): conflict-free ds_write_b64 / ds_read_b128
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int opix = 32 * ph + 16 * q + fr;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = 8 * kh + 4 * i + fg;   // 4-channel column of the pixel's row
                u32x2 o;
                o[0] = (uint32_t)f32_to_bf16(acc[q][i][0]) | ((uint32_t)f32_to_bf16(acc[q][i][1]) << 16);
                o[1] = (uint32_t)f32_to_bf16(acc[q][i][2]) | ((uint32_t)f32_to_bf16(acc[q][i][3]) << 16);
                *(u32x2*)(sout + obuf * OUTB + opix * 128 + ((((col >> 1) ^ ((opix >> 1) & 7)) << 4) | ((col & 1) << 3))) = o;
            }
        }
    };

    // write-back of one stage's rows: 8 row groups (8 pixels x 128 B = 1 KiB contiguous in memory), 2 per wave
    // BatchNorm batch statistics of the NEXT layer, for free: the write-back lane holds 8 stored channels of one
    // pixel; per-lane fp32 sums over the block's pixels, combined per block at the end (deterministic), replace
    // a full read pass over the output (primia_bn_fwd_train_from_sums consumes the per-block partials).
    // (pairs of channels in 64-bit registers: v_pk_add_f32 / v_pk_fma_f32 form two sums per instruction — the same additions and
    // fused multiply-adds per element as before, half the instructions)
    c64_f32x2 sp1[4], sp2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sp1[k] = sp2[k] = c64_f32x2{0.f, 0.f};

    // The write-back of a patch in two parts, so that the loop can put its requests between them: wb_load reads the rows (and,
    // sums forms, the BatchNorm rows and mask words) from LDS into registers — the fragment ring and the accumulators are dead
    // here, the registers are free — wb_finish stores the rows and does the arithmetic.
    struct WbRegs {
        u32x4 v[2], yv[2];
        unsigned bm[2], so[2];     // mask byte of the lane's chunk; byte offset of the lane's 16-byte chunk in dst
        bool live[2];
    };
    auto wb_load = [&](int obuf, WbRegs& r) {
        const int px = lane >> 3, c16 = lane & 7;  // pixel of the row, 16-B chunk
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = 2 * wave + q;                // patch row handled by this wave
            if constexpr (EXACT) {
                r.live[q] = cw.t < t1;
                r.so[q] = r.live[q] && !C64_DBG(1) ? (unsigned)cw.org * 128u + aux_off[q] : kOob;
            } else {
                const int ho = cw.ph * 8 + g, wo = cw.pw * 8 + px;
                r.live[q] = cw.t < t1 && ho < p.H && wo < p.W;
                r.so[q] = r.live[q] && !C64_DBG(1) ? (unsigned)(((cw.n * p.H + ho) * p.W + wo) * 128 + c16 * 16) : kOob;
            }
            const int opx2 = g * 8 + px;
            r.v[q] = *(const u32x4*)(sout + obuf * OUTB + opx2 * 128 + ((c16 ^ ((opx2 >> 1) & 7)) << 4));
            if constexpr (BNB != 0) {
                // the BatchNorm input row chunk of this pixel (requested an iteration ago, by THIS wave)
                r.yv[q] = *(const u32x4*)(saux + g * 1024 + lane * 16);
                if constexpr (BNB == 2)
                    r.bm[q] = *(const unsigned*)(sbm + wave * 256 + (q * 32 + px * 4 + (c16 >> 2) * 2) * 4) >> (8 * (c16 & 3));
            }
        }
        advance(cw);
    };
    auto wb_finish = [&](const WbRegs& r) {
        const int c16 = lane & 7;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const u32x4 v = r.v[q];
            // rows out through the buffer descriptor: out-of-image lanes carry an offset beyond num_records and are dropped
            // (s_nop 1: a 16-byte store reads its data registers for two more cycles; the hazard recognizer does not look inside asm)
            // (non-temporal stores: measured, no change — 4.862 vs 4.861 ms per step)
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(r.so[q]), "s"(rsrc_st) : "memory");
            if (r.live[q]) {
                if (BNB != 0 && !C64_DBG(8)) {
                    const u32x4 yv = r.yv[q];
                    const int bmq = (int)r.bm[q];
                    const f32x4 mu0 = *(const f32x4*)(bnc + c16 * 8), mu1 = *(const f32x4*)(bnc + c16 * 8 + 4);
                    f32x4 sc0, sc1, be0, be1;
                    if constexpr (BNB == 1) {
                        sc0 = *(const f32x4*)(bnc + 64 + c16 * 8), sc1 = *(const f32x4*)(bnc + 64 + c16 * 8 + 4);
                        be0 = *(const f32x4*)(bnc + 128 + c16 * 8), be1 = *(const f32x4*)(bnc + 128 + c16 * 8 + 4);
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {     // channels 2 kk, 2 kk + 1 of the lane's chunk
                        const unsigned yw = yv[kk], dw = v[kk];
                        const c64_f32x2 y2 = {__uint_as_float(yw << 16), __uint_as_float(yw & 0xffff0000u)};
                        const c64_f32x2 mu2 = kk < 2 ? c64_f32x2{mu0[2 * kk], mu0[2 * kk + 1]} : c64_f32x2{mu1[2 * kk - 4], mu1[2 * kk - 3]};
                        const c64_f32x2 t2 = y2 - mu2;
                        c64_f32x2 g2;
                        if constexpr (BNB == 1) {
                            const c64_f32x2 sc2 = kk < 2 ? c64_f32x2{sc0[2 * kk], sc0[2 * kk + 1]} : c64_f32x2{sc1[2 * kk - 4], sc1[2 * kk - 3]};
                            const c64_f32x2 be2 = kk < 2 ? c64_f32x2{be0[2 * kk], be0[2 * kk + 1]} : c64_f32x2{be1[2 * kk - 4], be1[2 * kk - 3]};
                            const c64_f32x2 z2 = __builtin_elementwise_fma(t2, sc2, be2);
                            g2 = c64_f32x2{z2[0] > 0.f ? __uint_as_float(dw << 16) : 0.f, z2[1] > 0.f ? __uint_as_float(dw & 0xffff0000u) : 0.f};
                        } else if constexpr (BNB == 2) {   // mask bit -> all-ones / zero word, ANDed onto the value's bits
                            g2 = c64_f32x2{__uint_as_float((dw << 16) & (unsigned)__builtin_amdgcn_sbfe(bmq, 2 * kk, 1)),
                                           __uint_as_float((dw & 0xffff0000u) & (unsigned)__builtin_amdgcn_sbfe(bmq, 2 * kk + 1, 1))};
                        } else {
                            g2 = c64_f32x2{y2[0] > 0.f ? __uint_as_float(dw << 16) : 0.f, y2[1] > 0.f ? __uint_as_float(dw & 0xffff0000u) : 0.f};
                        }
                        sp1[kk] += g2;
                        sp2[kk] = __builtin_elementwise_fma(g2, t2, sp2[kk]);      // (x invstd, or 1 / gamma, once per channel at the end)
                    }
                } else if (p.stat_partials) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const c64_f32x2 x2 = {__uint_as_float(v[k] << 16), __uint_as_float(v[k] & 0xffff0000u)};
                        sp1[k] += x2;
                        sp2[k] = __builtin_elementwise_fma(x2, x2, sp2[k]);
                    }
                }
            }
        }
    };

    // STAGES-deep LDS ring, one raw barrier per stage, counted vmcnt.  Per iteration s a wave issues, in this order: NAUX
    // loads (BatchNorm rows of patch s), d halo pieces of patch s + STAGES - 1 (d = 4 for wave 0, else 3), 2 row stores
    // (write-back of patch s - 1), and at the end of compute(s) NOLD loads (old rows of patch s + 1).  Loads retire in order,
    // so "at most n in flight" means "only the n youngest": at the top of iteration s the halo of patch s (and the
    // BatchNorm rows requested after it) have landed once only the younger halo pieces and the old rows remain.
    // Ragged images and the last stage drain (vmcnt(0)).
    // (Counting the row stores in as well is WRONG: stores and loads retire out of order with respect to each other, so two
    // early store completions would let the wave through with two halo pieces still in flight.  Seen as run-to-run
    // differences of the training loss once every other source of nondeterminism was gone.)
    if constexpr (BNB != 0) __syncthreads();        // (the constants in LDS are visible to every wave)
    prefetch_aux();            // patch 0's BatchNorm rows: older than every halo piece, landed by the first counted wait
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nstages) stage(s);
    prefetch_old();            // patch 0's old rows: the youngest loads at the top of iteration 0, as in every iteration
    // weights resident — and everything requested so far landed — before any counted wait below.  The builtin, not inline asm:
    // it tells the compiler's wait-count pass that the weight registers are ready, so that it puts no wait of its own in front
    // of their first use INSIDE the loop (which would drain the LDS-DMA ring every iteration).
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    int cur = 0, nxt = STAGES - 1;
#if C64_PROBE
    tlast = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < nstages; ++s) {
        // 4-deep ring (plain / accumulate): patches s + 1 and s + 2 may stay in flight; with BatchNorm rows (3-deep): s + 1 only
        const int keep = (STAGES == 4 && s + 2 < nstages) ? 2 : (s + 1 < nstages ? 1 : 0);
        if (exact && keep) {
            // (4-deep accumulate form: the old rows of patch s - 1, long landed, still sit between the two halos in issue order)
            // (what this wait waits for is the halo, not the row stores that also count: allowing two more — unsafe, timing
            // only — moved it from 507 to 483 cycles per patch; with the write-back phase shortened the kernel runs at 4.0 TB/s
            // and the requests of three patches ahead queue behind the memory system's throughput)
            constexpr int kSteady = STAGES == 4 ? 6 + 2 * NOLD : 3 + NOLD;      // waves 1-3 (3 pieces); wave 0: + 2 | + 1
            c64_wait_vm_pick<kSteady, kSteady + (STAGES == 4 ? 2 : 1)>(keep * dstage + (keep == 2 && s > 0 ? 2 : 1) * NOLD);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the result rows this wave wrote to LDS in the previous iteration must have LANDED before the barrier lets
        // the other waves read them (a raw s_barrier does not wait for the wave's own outstanding ds_write; with two
        // blocks per CU competing for the LDS the write-back occasionally read a stale 1-KiB row)
        C64_T(0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        C64_T(1)
        // Outside the matrix loop the wave issues its vector-memory instructions ahead of the other block's wave on this SIMD,
        // which is multiplying (the reverse — matrix loop first — is slower than no priorities at all: 73.8 us).
        __builtin_amdgcn_s_setprio(3);
        int younger = 0;       // loads issued in this iteration, after the old rows of patch s
        // Order: the previous patch's rows LDS -> registers; then this iteration's requests — the BatchNorm rows of patch s
        // (their buffer has just been read), the halo pieces — and only then the row stores, which fill the memory pipe, and
        // the arithmetic.  Layer1's forward convolution alone (tools/micro/c64_probe.py): 71.7 us with the stores first through
        // 64-bit pointers, 68.3 with the priority and the descriptor, 64.7 with the halo requests before the stores, 63.4 with
        // the LDS reads before the requests.
        WbRegs wb;
        if (s > 0) wb_load((s - 1) & 1, wb);
        if (BNB != 0 && s > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the reads above are done before new rows may land in the buffer
            prefetch_aux();    // the cursor now points at patch s, written back in iteration s + 1
            younger += C64_DBG(32) ? 0 : NAUX;
        }
        if (s + STAGES - 1 < nstages && !C64_DBG(2)) {
            stage(nxt);
            younger += dstage;
        }
        C64_T(2)
        if (s > 0) wb_finish(wb);
        C64_T(3)
        __builtin_amdgcn_s_setprio(0);
        compute(cur, s & 1, exact ? younger : 0);
        C64_T(5)
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }
#if C64_PROBE
    if (lane == 0 && blockIdx.x < 1024) {
        for (int i = 0; i < 10; ++i) c64_probe_buf[(blockIdx.x * 4 + wave) * 16 + i] = tacc[i];
        c64_probe_buf[(blockIdx.x * 4 + wave) * 16 + 10] = (unsigned long long)nstages;
    }
#endif
    if constexpr (ACC || BNB != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        WbRegs wb;
        wb_load((nstages - 1) & 1, wb);
        wb_finish(wb);
    }
    if (p.stat_partials) {
        float st1[8], st2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            st1[k] = sp1[k >> 1][k & 1];
            st2[k] = sp2[k >> 1][k & 1];
        }
        // lanes with equal (lane & 7) hold the same 8 channels: fold the 8 pixel lanes, then the 4 waves
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                st1[k] += __shfl_xor(st1[k], o, 64);
                st2[k] += __shfl_xor(st2[k], o, 64);
            }
        }
        __syncthreads();
        float* red = (float*)smem;  // [4 waves][2][64]
        if (lane < 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                red[(wave * 2 + 0) * 64 + lane * 8 + k] = st1[k];
                red[(wave * 2 + 1) * 64 + lane * 8 + k] = st2[k];
            }
        }
        __syncthreads();
        if (tid < 128) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) a += red[w * 128 + tid];
            if (BNB != 0 && tid >= 64) {      // sum g * (y - mean) -> sum g * xhat
                if constexpr (BNB == 3) {
                    // (gamma 0 or tiny against beta: xhat is not recoverable from the stored p — the consumer's rare path)
                    const float gm = p.bnb_gamma[tid - 64];
                    a *= pool_xhat_recoverable(gm, p.bnb_beta[tid - 64]) ? 1.f / gm : 0.f;
                } else {
                    a *= p.bnb_invstd[tid - 64];
                }
            }
            p.stat_partials[(long)blockIdx.x * 128 + tid] = a;   // [block][2][64]
        }
    }
}

// returns PRIMIA_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the implicit GEMM)
int conv3x3_c64_grid(int N, int H, int W) {
    const long total = (long)N * ((H + 7) / 8) * ((W + 7) / 8);
    const int target = PRIMIA_OPT(c64_blocks) > 0 ? PRIMIA_OPT(c64_blocks) : 512;
    long per = (total + target - 1) / target;
    if (per < 1) per = 1;
    return (int)((total + per - 1) / per);
}

int conv3x3_c64_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int flip, int accumulate,
                         hipStream_t st, float* stat_partials, const uint8_t* acc_mask, const LhBnBwd* bnb,
                         const C64AccBnb* abnb) {
    if ((long)N * H * W * 64 >= (1L << 31)) return PRIMIA_ERR_UNSUPPORTED;
    const bool with_bnb = bnb && bnb->y;
    const int amode = (abnb && abnb->aux) ? abnb->mode : 0;
    if (with_bnb && (!flip || accumulate || !stat_partials || amode)) return PRIMIA_ERR_ARG;
    if (amode && (!flip || !accumulate || !stat_partials || (amode != 2 && amode != 3) || (amode == 2 && !abnb->mask)))
        return PRIMIA_ERR_ARG;
    C64Params p;
    p.bnb_y = with_bnb ? bnb->y : nullptr;
    p.bnb_mean = with_bnb ? bnb->mean : nullptr; p.bnb_invstd = with_bnb ? bnb->invstd : nullptr;
    p.bnb_gamma = with_bnb ? bnb->gamma : nullptr; p.bnb_beta = with_bnb ? bnb->beta : nullptr;
    p.bnb_mask = nullptr;
    if (amode == 2) {          // residual BatchNorm: mask bytes, mean, invstd
        p.bnb_y = abnb->aux; p.bnb_mask = abnb->mask; p.bnb_mean = abnb->c0; p.bnb_invstd = abnb->c1;
    } else if (amode == 3) {   // through the max-pool: pooled activation, beta, gamma
        p.bnb_y = abnb->aux; p.bnb_beta = abnb->c0; p.bnb_gamma = abnb->c1;
    }
    p.src = src; p.wt = wt; p.dst = dst;
    p.N = N; p.H = H; p.W = W; p.flip = flip; p.accumulate = accumulate;
    p.stat_partials = stat_partials;
    p.acc_mask = accumulate ? acc_mask : nullptr;
    p.PH = (H + 7) / 8; p.PW = (W + 7) / 8; p.PPI = p.PH * p.PW;
    p.total = N * p.PPI;
    const int target = PRIMIA_OPT(c64_blocks) > 0 ? PRIMIA_OPT(c64_blocks) : 512;   // 2 per CU
    long per = (p.total + target - 1) / target;
    if (per < 1) per = 1;
    p.per_block = (int)per;
    p.debug = PRIMIA_PROBE ? PRIMIA_OPT(c64_dbg) : 0;       // (timing experiments: probe builds only, tools/micro/c64_probe.sh)
    const int grid = (int)((p.total + per - 1) / per);
    // a 4-deep ring (plain 68 KiB, accumulate 76 KiB per block, two blocks per CU) keeps 78 KB per CU in flight instead of 52
    const int deep = PRIMIA_OPT(c64_stages);
    const int stages = (!with_bnb && !amode && deep == 4) ? 4 : 3;
    auto lds_of = [](int stg, bool acc, bool aux) {
        return stg * 13 * 1024 + 2 * 64 * 128 + (acc ? 8192 + 1024 : 0) + (aux ? 8192 + 1024 : 0);
    };
    const size_t lds = (size_t)lds_of(stages, accumulate != 0, with_bnb || amode) + (amode == 2 ? 1024 : 0);
    // EXACT forms (whole 8 x 8 patches: layer1 at every input size that is a multiple of 32) — the general forms serve ragged images
    const bool ex = (H % 8 == 0) && (W % 8 == 0) && !p.debug;
    const int form = with_bnb ? 2 : (amode == 2 ? 5 : (amode == 3 ? 6 : (accumulate ? (stages == 4 ? 4 : 3) : (stages == 4 ? 1 : 0))));
    typedef void (*kern_t)(C64Params);
    static const kern_t kerns[2][7] = {
        {conv3x3_c64_kernel<false, 3>, conv3x3_c64_kernel<false, 4>, conv3x3_c64_kernel<false, 3, 1>, conv3x3_c64_kernel<true, 3>,
         conv3x3_c64_kernel<true, 4>, conv3x3_c64_kernel<true, 3, 2>, conv3x3_c64_kernel<true, 3, 3>},
        {conv3x3_c64_kernel<false, 3, 0, true>, conv3x3_c64_kernel<false, 4, 0, true>, conv3x3_c64_kernel<false, 3, 1, true>,
         conv3x3_c64_kernel<true, 3, 0, true>, conv3x3_c64_kernel<true, 4, 0, true>, conv3x3_c64_kernel<true, 3, 2, true>,
         conv3x3_c64_kernel<true, 3, 3, true>}};
    static bool attr_set[2][7] = {{false}};
    const kern_t kern = kerns[ex ? 1 : 0][form];
    if (!attr_set[ex ? 1 : 0][form]) {
        // the largest request of the form: plain 3 / 4 stages, + BatchNorm rows, accumulate 3 / 4 stages, + rows (+ mask words)
        const int need[7] = {lds_of(3, false, false), lds_of(4, false, false), lds_of(3, false, true), lds_of(3, true, false),
                             lds_of(4, true, false), lds_of(3, true, true) + 1024, lds_of(3, true, true)};
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, need[form]) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set[ex ? 1 : 0][form] = true;
    }
    kern<<<grid, 256, lds, st>>>(p);
    return launch_status();
}

}  // namespace primia

#if C64_PROBE
extern "C" int primia_c64_probe_read(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(primia::c64_probe_buf), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#endif
