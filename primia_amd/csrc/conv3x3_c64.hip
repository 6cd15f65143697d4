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
    int debug;        // timing experiments only (set by tools/micro builds, 0 in the library): 1 no stores, 2 no staging, 4 no MFMA loop
};

// Block = 4 waves (256 threads), TWO blocks per CU: the two waves of a SIMD belong to different blocks, so they are
// never in the same phase — one block's barrier, write-back, DMA issue and first-fragment latency run beside the other
// block's MFMAs.  (The 8-wave form ran every wave of the CU through those phases in lockstep: with staging and stores
// switched off it still took 60 us for 32 us worth of MFMAs.)  Wave (kh, ph): out-channels 32*kh..+31, patch rows
// 4*ph..4*ph+3 = two 16-pixel MFMA column blocks q = 0, 1; one 8x8 patch per stage.
template <bool ACC, int STAGES = 3, bool BNB = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_c64_kernel(C64Params p) {
    static_assert(!(ACC && BNB), "the BatchNorm-sums form is a plain data gradient");
    constexpr int HALO = 13 * 1024;        // 100 slots used, 104 staged (13 DMA instructions)
    constexpr int STAGE = HALO;            // one patch per stage
    // output rows of one stage: 64 pixels x 64 channels, bf16 — or fp32 in the accumulate form, which adds the old
    // values in the write-back phase (coalesced row loads, fp32 add, ONE rounding)
    constexpr int OPIX = ACC ? 256 : 128;  // bytes per staged pixel row
    constexpr int OUTB = 64 * OPIX;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES * STAGE + 2 * OUTB (+ 8 KiB of old rows: ACC)
    char* const sout = smem + STAGES * STAGE;
    // accumulate form: the OLD rows of the patch being computed, fetched by LDS-DMA one iteration before the write-back
    // adds them (64 pixels x 128 B, row group g at g * 1 KiB, lane-linear: pixel lane / 8, 16-B chunk lane % 8)
    char* const sold = sout + 2 * OUTB;       // (BNB: the BatchNorm input rows of the patch being written back, then 1 KiB of constants)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 1, ph = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    const int t0 = blockIdx.x * p.per_block;
    int t1 = t0 + p.per_block;
    if (t1 > p.total) t1 = p.total;
    const int nstages = t1 - t0;
    if (nstages <= 0) return;

    // ---- weights -> registers: A fragment (i, j): row 32*kh + 16*i + fr, elements j*32 + 8*fg .. +7 -------
    bf16x8_t wreg[18][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16* wrow = p.wt + (long)(32 * kh + 16 * i + fr) * 576 + 8 * fg;
#pragma unroll
        for (int j = 0; j < 18; ++j) wreg[j][i] = *(const bf16x8_t*)(wrow + j * 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // weights resident before any counted wait below

    // patch cursor: (n, ph, pw) of consecutive patch ids, advanced by one
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
        const int rb = cs.ph * 8, cb = cs.pw * 8;
        const int org = ((cs.n * p.H + rb) * p.W + cb) * 128;
        // valid halo rows hy: 0 <= rb + hy - 1 < H, columns likewise
        int rhi = p.H - rb + 1, chi = p.W - cb + 1;
        rhi = rhi > 10 ? 10 : rhi;
        chi = chi > 10 ? 10 : chi;
        const unsigned rowm = ((1u << rhi) - 1u) & ~(rb == 0 ? 1u : 0u);
        const unsigned colm = ((1u << chi) - 1u) & ~(cb == 0 ? 1u : 0u);
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

    auto compute = [&](int buf, int obuf) {
        const char* sb = smem + buf * STAGE;
        f32x4 acc[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[q][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // B fragments: 3-deep register ring, reads run two steps ahead of the MFMAs that consume them
        bf16x8_t bq[3][2];
        auto rd = [&](int step, int slot) {
            const int x = (step & 1) * 64;
            const char* q0 = sb + (offb[step >> 1] ^ x);
            bq[slot][0] = *(const bf16x8_t*)q0;
            bq[slot][1] = *(const bf16x8_t*)(q0 + 20 * 128);
        };
        if (!(p.debug & 4)) {
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
        // ---- results -> LDS rows (bf16; fp32 for the accumulate form) ---------------------------------------
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int opix = 32 * ph + 16 * q + fr;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = 8 * kh + 4 * i + fg;   // 4-channel column of the pixel's row
                if constexpr (ACC) {
                    // 16-byte fp32 chunks, chunk index swizzled with the pixel: conflict-free ds_write_b128
                    *(f32x4*)(sout + obuf * OUTB + opix * OPIX + ((col ^ (opix & 15)) << 4)) = acc[q][i];
                } else {
                    u32x2 o;
                    o[0] = (uint32_t)f32_to_bf16(acc[q][i][0]) | ((uint32_t)f32_to_bf16(acc[q][i][1]) << 16);
                    o[1] = (uint32_t)f32_to_bf16(acc[q][i][2]) | ((uint32_t)f32_to_bf16(acc[q][i][3]) << 16);
                    *(u32x2*)(sout + obuf * OUTB + opix * 128 + ((((col >> 1) ^ ((opix >> 1) & 7)) << 4) | ((col & 1) << 3))) = o;
                }
            }
        }
    };

    // write-back of one stage's rows: 8 row groups (8 pixels x 128 B = 1 KiB contiguous in memory), 2 per wave
    // BatchNorm batch statistics of the NEXT layer, for free: the write-back lane holds 8 stored channels of one
    // pixel; per-lane fp32 sums over the block's pixels, combined per block at the end (deterministic), replace
    // a full read pass over the output (primia_bn_fwd_train_from_sums consumes the per-block partials).
    float st1[8], st2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) st1[k] = st2[k] = 0.f;

    // Accumulate form: the old rows (and their ReLU-mask bytes) of the patch the write-back cursor points at, requested
    // one iteration AHEAD of the write-back that adds them — the first version loaded them inside the write-back, a full
    // memory latency per stage in front of the stores, with the compiler's conservative vmcnt(0) draining the DMA ring
    // on top (93 us against 58 us for the plain form, for 103 MB more).  The rows travel by LDS-DMA (a row group = 8
    // pixels x 128 B = one instruction, no registers across the matrix phase: the kernel sits at 242 of 256), the mask
    // bytes by an inline-asm byte load the compiler's wait counting does not see; both are older than the halo pieces
    // requested after them, so the counted wait at the top of the next iteration covers them.
    c64_i32x4 rsrc_dst = rsrc;
    unsigned mk[2] = {0xffu, 0xffu};
    if constexpr (ACC || BNB) {
        const unsigned long long a = (unsigned long long)(BNB ? (const void*)p.bnb_y : (const void*)p.dst);
        rsrc_dst[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rsrc_dst[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);
    }
    // BNB: per-channel constants of the BatchNorm in LDS behind the row buffer: [mean | invstd * gamma | beta | invstd][64]
    float* const bnc = (float*)(sold + 8192);
    if constexpr (BNB) {
        if (tid < 64) {
            const float is = p.bnb_invstd[tid];
            bnc[tid] = p.bnb_mean[tid];
            bnc[64 + tid] = is * p.bnb_gamma[tid];
            bnc[128 + tid] = p.bnb_beta[tid];
            bnc[192 + tid] = is;
        }
    }
    auto prefetch_old = [&]() {
        if constexpr (ACC || BNB) {
            const int px = lane >> 3, c16 = lane & 7;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int g = 2 * wave + q;
                const int ho = cw.ph * 8 + g, wo = cw.pw * 8 + px;
                const bool live = cw.t < t1 && ho < p.H && wo < p.W;
                const unsigned eoff = (unsigned)(((cw.n * p.H + ho) * p.W + wo) * 64 + c16 * 8);   // elements (< 2^31)
                const unsigned voff = live ? eoff * 2u : kOob;
                const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + STAGES * STAGE + 2 * OUTB + g * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(voff), "s"(rsrc_dst), "s"(m0v) : "memory");
                if (ACC && p.acc_mask) {
                    const uint8_t* mp = p.acc_mask + (live ? (eoff >> 3) : 0u);
                    asm volatile("global_load_ubyte %0, %1, off" : "=v"(mk[q]) : "v"(mp) : "memory");
                }
            }
        }
    };

    auto writeback = [&](int obuf) {
        const int px = lane >> 3, c16 = lane & 7;  // pixel of the row, 16-B chunk
        bool live[2];
        bf16* gp[2];
        u32x4 old[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int g = 2 * wave + q;                // patch row handled by this wave
            const int ho = cw.ph * 8 + g, wo = cw.pw * 8 + px;
            live[q] = cw.t < t1 && ho < p.H && wo < p.W;
            gp[q] = p.dst + ((long)(cw.n * p.H + ho) * p.W + wo) * 64 + c16 * 8;
            old[q] = u32x4{0, 0, 0, 0};
            if constexpr (ACC) {
                // (requested by prefetch_old() an iteration ago, by THIS wave: no barrier between the DMA and this read)
                old[q] = *(const u32x4*)(sold + g * 1024 + lane * 16);
                if (p.acc_mask) {   // old value = gradient through a ReLU whose mask is applied here, not stored
                    const unsigned m = mk[q];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        old[q][k] &= ((m >> (2 * k)) & 1u ? 0x0000ffffu : 0u) | ((m >> (2 * k + 1)) & 1u ? 0xffff0000u : 0u);
                }
            }
        }
        advance(cw);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int opx2 = (2 * wave + q) * 8 + px;
            u32x4 v;
            if constexpr (ACC) {
                const char* row = sout + obuf * OUTB + opx2 * OPIX;
                const f32x4 lo = *(const f32x4*)(row + (((2 * c16) ^ (opx2 & 15)) << 4));
                const f32x4 hi = *(const f32x4*)(row + (((2 * c16 + 1) ^ (opx2 & 15)) << 4));
                float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f[2 * k] += __uint_as_float(old[q][k] << 16);
                    f[2 * k + 1] += __uint_as_float(old[q][k] & 0xffff0000u);
                    v[k] = (uint32_t)f32_to_bf16(f[2 * k]) | ((uint32_t)f32_to_bf16(f[2 * k + 1]) << 16);
                }
            } else {
                v = *(const u32x4*)(sout + obuf * OUTB + opx2 * 128 + ((c16 ^ ((opx2 >> 1) & 7)) << 4));
            }
            if (live[q] && !((p.debug & 1) && v[0] != 12345u)) {
                *(u32x4*)gp[q] = v;      // (non-temporal stores: measured, no change — 4.862 vs 4.861 ms per step)
                if constexpr (BNB) {
                    // the BatchNorm input row chunk of this pixel (requested an iteration ago, by THIS wave, into `sold`)
                    const u32x4 yv = *(const u32x4*)(sold + (2 * wave + q) * 1024 + lane * 16);
                    const f32x4 mu0 = *(const f32x4*)(bnc + c16 * 8), mu1 = *(const f32x4*)(bnc + c16 * 8 + 4);
                    const f32x4 sc0 = *(const f32x4*)(bnc + 64 + c16 * 8), sc1 = *(const f32x4*)(bnc + 64 + c16 * 8 + 4);
                    const f32x4 be0 = *(const f32x4*)(bnc + 128 + c16 * 8), be1 = *(const f32x4*)(bnc + 128 + c16 * 8 + 4);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float yk = __uint_as_float((k & 1) ? (yv[k >> 1] & 0xffff0000u) : (yv[k >> 1] << 16));
                        const float dk = __uint_as_float((k & 1) ? (v[k >> 1] & 0xffff0000u) : (v[k >> 1] << 16));
                        const float t = yk - (k < 4 ? mu0[k & 3] : mu1[k & 3]);
                        const float zz = __builtin_fmaf(t, k < 4 ? sc0[k & 3] : sc1[k & 3], k < 4 ? be0[k & 3] : be1[k & 3]);
                        const float gk = zz > 0.f ? dk : 0.f;
                        st1[k] += gk;
                        st2[k] = __builtin_fmaf(gk, t, st2[k]);      // (x invstd once per channel at the end)
                    }
                } else if (p.stat_partials) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __uint_as_float(v[k] << 16), hi = __uint_as_float(v[k] & 0xffff0000u);
                        st1[2 * k] += lo;
                        st2[2 * k] += lo * lo;
                        st1[2 * k + 1] += hi;
                        st2[2 * k + 1] += hi * hi;
                    }
                }
            }
        }
    };

    // STAGES-deep LDS ring, one raw barrier per stage, counted vmcnt.  Per iteration a wave issues, in this order,
    // 2 row stores (write-back of the previous stage) and d DMA instructions (d = 4 for wave 0, else 3).  At the top
    // of iteration s, DMA(s) has landed once at most the d instructions of DMA(s+1) remain in flight.  Ragged
    // images, the accumulate form (its loads are waited for by the compiler, conservatively) and the last stage use
    // vmcnt(0).
    const bool exact = (p.H % 8 == 0) && (p.W % 8 == 0) && !(p.debug & 3);
    if constexpr (BNB) __syncthreads();        // (the constants in LDS are visible to every wave)
    prefetch_old();            // patch 0's old rows: older than every halo piece, landed by the first counted wait
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nstages) stage(s);
    int cur = 0, nxt = STAGES - 1;
    for (int s = 0; s < nstages; ++s) {
        if (exact && STAGES == 4 && s + 2 < nstages) {
            // 4-deep ring: DMA(s+1) and DMA(s+2) may stay in flight
            if (wave == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else if (exact && s + 1 < nstages) {
            // only DMA(s+1) may stay in flight.  (Counting the row stores of iteration s-1 in as well — vmcnt(d + 2) —
            // is WRONG: stores and loads retire out of order with respect to each other, so two early store
            // completions let the wave through with two pieces of DMA(s) still in flight.  Seen as run-to-run
            // differences of the training loss once every other source of nondeterminism was gone.)
            if (wave == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the result rows this wave wrote to LDS in the previous iteration must have LANDED before the barrier lets
        // the other waves read them (a raw s_barrier does not wait for the wave's own outstanding ds_write; with two
        // blocks per CU competing for the LDS the write-back occasionally read a stale 1-KiB row)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (ACC) asm volatile("" : "+v"(mk[0]), "+v"(mk[1])::"memory");   // (the mask bytes are valid from here)
        __builtin_amdgcn_s_barrier();
        if (s > 0) {
            writeback((s - 1) & 1);
            prefetch_old();    // the cursor now points at patch s, written back in iteration s + 1
        }
        if (s + STAGES - 1 < nstages && !(p.debug & 2)) stage(nxt);
        compute(cur, s & 1);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }
    if constexpr (ACC || BNB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(mk[0]), "+v"(mk[1])::"memory");
    }
    __syncthreads();
    writeback((nstages - 1) & 1);
    if (p.stat_partials) {
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
            if (BNB && tid >= 64) a *= p.bnb_invstd[tid - 64];   // sum g * (y - mean) -> sum g * xhat
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
                         hipStream_t st, float* stat_partials, const uint8_t* acc_mask, const LhBnBwd* bnb) {
    if ((long)N * H * W * 64 >= (1L << 31)) return PRIMIA_ERR_UNSUPPORTED;
    const bool with_bnb = bnb && bnb->y;
    if (with_bnb && (!flip || accumulate || !stat_partials)) return PRIMIA_ERR_ARG;
    C64Params p;
    p.bnb_y = with_bnb ? bnb->y : nullptr;
    p.bnb_mean = with_bnb ? bnb->mean : nullptr; p.bnb_invstd = with_bnb ? bnb->invstd : nullptr;
    p.bnb_gamma = with_bnb ? bnb->gamma : nullptr; p.bnb_beta = with_bnb ? bnb->beta : nullptr;
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
    p.debug = 0;       // (timing-experiment bits of tools/micro; never set by the library)
    const int grid = (int)((p.total + per - 1) / per);
    // plain form: a 4-deep ring (68 KiB per block, two blocks per CU) keeps 78 KB per CU in flight instead of 52
    const int deep = PRIMIA_OPT(c64_stages);
    const int stages = (!accumulate && !with_bnb && deep == 4) ? 4 : 3;
    const size_t lds = (size_t)stages * 13 * 1024 + 2 * 64 * (accumulate ? 256 : 128) + (accumulate ? 8192 : 0) + (with_bnb ? 8192 + 1024 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        const int lds_plain = 3 * 13 * 1024 + 2 * 64 * 128, lds_acc = 3 * 13 * 1024 + 2 * 64 * 256 + 8192;
        const int lds_plain4 = 4 * 13 * 1024 + 2 * 64 * 128;
        const int lds_bnb = 3 * 13 * 1024 + 2 * 64 * 128 + 8192 + 1024;
        if (hipFuncSetAttribute((const void*)conv3x3_c64_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_plain) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_c64_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_plain4) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_c64_kernel<false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bnb) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_c64_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_acc) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    if (with_bnb)
        conv3x3_c64_kernel<false, 3, true><<<grid, 256, lds, st>>>(p);
    else if (accumulate)
        conv3x3_c64_kernel<true, 3><<<grid, 256, lds, st>>>(p);
    else if (stages == 4)
        conv3x3_c64_kernel<false, 4><<<grid, 256, lds, st>>>(p);
    else
        conv3x3_c64_kernel<false, 3><<<grid, 256, lds, st>>>(p);
    return launch_status();
}

}  // namespace primia
