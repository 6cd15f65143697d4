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

#include "conv_wgrad.h"

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
    float* stat_partials;  // if set: per-block BatchNorm partial sums [grid][2][64] of the values as stored
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
    // bn1's batch statistics for free (see conv3x3_c64.hip): per-lane sums of the 8 stored channels it writes
    float st1[8], st2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) st1[k] = st2[k] = 0.f;

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
                if (live) {
                    // (non-temporal: -0.2 % on the step, same-box A/B; y == NULL: statistics only, nothing is stored)
                    if (p.y) __builtin_nontemporal_store(v, (u32x4*)(rowp + px * 64 + c16 * 8));
                    if (p.stat_partials) {
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
            advance(cw);
        }
    };

    // 3-deep LDS ring, one raw barrier per stage, counted vmcnt.  Per iteration a wave issues, in this order, 4 row
    // stores (write-back of the previous stage) and d DMA instructions (d = 2 for waves 0..5, 1 for waves 6, 7).
    // At the top of iteration s at most DMA(s+1) may remain in flight.  The last stage (a patch may be dead) uses
    // vmcnt(0).
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nstages) stage(s);
    int cur = 0, nxt = STAGES - 1;
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) {
            // only DMA(s+1) may stay in flight (stores retire out of order with respect to loads: they must not be
            // counted in, see conv3x3_c64.hip)
            if (wave < 6) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the result rows this wave wrote to LDS in the previous iteration must have LANDED before the barrier lets
        // the other waves read them (a raw s_barrier does not wait for the wave's own outstanding ds_write; with two
        // blocks per CU competing for the LDS the write-back occasionally read a stale 1-KiB row)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s > 0) writeback((s - 1) & 1);
        if (s + STAGES - 1 < nstages) stage(nxt);
        compute(cur, s & 1);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }
    __syncthreads();
    writeback((nstages - 1) & 1);
    if (p.stat_partials) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                st1[k] += __shfl_xor(st1[k], o, 64);
                st2[k] += __shfl_xor(st2[k], o, 64);
            }
        }
        __syncthreads();
        float* red = (float*)smem;  // [8 waves][2][64]
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
            for (int w = 0; w < 8; ++w) a += red[w * 128 + tid];
            p.stat_partials[(long)blockIdx.x * 128 + tid] = a;
        }
    }
}

static int stem_fwd_grid(int N, int H, int W, int* per_block) {
    const long total = (long)N * (H / 2 / 8) * (W / 2 / 16);
    const int target = PRIMIA_OPT(stem_blocks) > 0 ? PRIMIA_OPT(stem_blocks) : 256;
    long per = (total + target - 1) / target;
    per = (per + 1) & ~1L;
    if (per < 2) per = 2;
    if (per_block) *per_block = (int)per;
    return (int)((total + per - 1) / per);
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

static int stem_conv_fwd_impl(const void* x_padded, const void* w_fwd, void* y, float* stat_partials, int N, int H,
                              int W, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(x_padded && w_fwd && (y || stat_partials) && N > 0 && H > 0 && W > 0);
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
    const int grid = stem_fwd_grid(N, H, W, &p.per_block);
    p.stat_partials = stat_partials;
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

int primia_stem_conv_fwd(const void* x_padded, const void* w_fwd, void* y, int N, int H, int W, int dtype,
                         primia_stream_t stream) {
    return stem_conv_fwd_impl(x_padded, w_fwd, y, nullptr, N, H, W, dtype, stream);
}

int primia_stem_conv_stat_slots(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32) return PRIMIA_ERR_ARG;
    return stem_fwd_grid(N, H, W, nullptr);
}

// pass 1 of the stem without its activation (stem_fwd_fused.hip): conv1's tiles -> bn1's partial sums, nothing stored
int primia_stem_conv_stats(const void* x_padded, const void* w_fwd, float* stat_partials, int N, int H, int W, int dtype,
                           primia_stream_t stream) {
    PRIMIA_REQUIRE(stat_partials);
    return stem_conv_fwd_impl(x_padded, w_fwd, nullptr, stat_partials, N, H, W, dtype, stream);
}

int primia_stem_conv_fwd_stats(const void* x_padded, const void* w_fwd, void* y, float* stat_partials, int N, int H,
                               int W, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(stat_partials);
    return stem_conv_fwd_impl(x_padded, w_fwd, y, stat_partials, N, H, W, dtype, stream);
}

}  // extern "C"

// =================================================================================================
// Weight gradient of conv1 from the padded input (bf16):
//   dw[k, r*32 + s*4 + c] += sum over pixels dy[n, oy, ox, k] * xp[n, 2*oy + r, 2*ox + s, c]
// One 8-wave block owns the whole 64 x 256 accumulator (wave (kh, rq): out-channels 32*kh..+31, kernel
// rows 2*rq, 2*rq+1 = 32 VGPRs) and walks 8 x 16 output patches: the dy tile (128 pixels x 64 channels)
// and the 21 x 40 input patch are staged once per patch by LDS-DMA; both MFMA operands are transposing
// reads (ds_read_b64_tr_b16): A from the dy tile, B straight from the input patch, where consecutive
// output pixels are 16 bytes apart and a kernel row's 32 elements are 64 contiguous bytes.
// Kernel row 7 and tap s = 7 carry no weight: their accumulator slots are never written (stay zero).
// =================================================================================================
namespace primia {

struct StemWgParams {
    const bf16* xp;
    const bf16* dy;
    float* dw;
    float* ws;      // atomic-free path: one [64][256] fp32 slab per block (or null: atomics into dw)
    double* sqnorm; // DP-SGD norm pass: one block per image, adds ||dW_n||^2 to sqnorm[blockIdx.x] instead of writing
    int N, Hp, Wp, Ho, Wo;
    int PH, PW, PPI;
    int total, per_block;
};

__device__ __forceinline__ int stem_key_lin(int slot) { return ((slot >> 1) & 1) | (((slot >> 3) & 1) << 1); }

__global__ __launch_bounds__(512) void stem_conv_wgrad_kernel(StemWgParams p) {
    constexpr int STAGES = 3;
    constexpr int XB = 7 * 1024, DYB = 16 * 1024, STAGE = XB + DYB;
    typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES * STAGE
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, rq = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int tp = fr >> 2, tc8 = (fr & 3) * 8;

    const int t0 = blockIdx.x * p.per_block;
    int t1 = t0 + p.per_block;
    if (t1 > p.total) t1 = p.total;
    const int nstages = t1 - t0;
    if (nstages <= 0) return;

    int sn = t0 / p.PPI, sph, spw;
    {
        const int rem = t0 - sn * p.PPI;
        sph = rem / p.PW;
        spw = rem - sph * p.PW;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // 23 DMA instructions per stage: 7 input-patch + 16 dy; wave w issues w, w + 8, w + 16 (< 23)
    auto stage = [&](int buf) {
        const long xbase = ((long)(sn * p.Hp + sph * 16) * p.Wp + spw * 32) * 4;
        const long ybase = ((long)(sn * p.Ho + sph * 8) * p.Wo + spw * 16) * 64;
        if (++spw == p.PW) {
            spw = 0;
            if (++sph == p.PH) {
                sph = 0;
                ++sn;
            }
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int idx = wave + 8 * it;
            if (idx >= 23) break;
            const bf16* g;
            if (idx < 7) {
                const int G = idx * 64 + lane;
                const int row = G / 20, c16 = G - row * 20;
                g = G < 420 ? p.xp + xbase + ((long)row * p.Wp + 2 * c16) * 4 : (const bf16*)kStemZeroPage;
            } else {
                const int slot = (idx - 7) * 8 + (lane >> 3), sl = lane & 7;
                const int chunk = ((((sl >> 1) ^ stem_key_lin(slot)) << 1) | (sl & 1));
                g = p.dy + ybase + ((long)(slot >> 4) * p.Wo + (slot & 15)) * 64 + chunk * 8;
            }
            stem_dma16(g, __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE + idx * 1024));
        }
    };

    f32x4 acc[2][2][2];  // [K fragment i][kernel row rr][element half h]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[i][rr][h] = f32x4{0.f, 0.f, 0.f, 0.f};

    // lane-constant parts of the read addresses
    const int a_slot = 8 * fg + tp;                                      // + 32*ks (+4)
    const int b_off = ((fg >> 1) * 2) * 320 + 16 * (8 * (fg & 1) + tp) + tc8;  // + (4*ks + r)*320 + 32*h (+64 for the hi read)

    auto compute = [&](int buf) {
        const char* lx = smem + buf * STAGE;
        const char* la = lx + XB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t a[2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s0 = 32 * ks + a_slot, s1 = s0 + 4, cg = 2 * kh + i;
                bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(la + s0 * 128 + ((cg ^ stem_key_lin(s0)) << 5) + tc8));
                bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(la + s1 * 128 + ((cg ^ stem_key_lin(s1)) << 5) + tc8));
                a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * rq + rr;  // r == 7: no such kernel row (skipped below, wave-uniform)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const char* q = lx + b_off + (4 * ks + r) * 320 + 32 * h;
                    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)q);
                    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(q + 64));
                    b[rr][h] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if (2 * rq + rr == 7) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][rr][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[rr][h], acc[i][rr][h], 0, 0, 0);
            }
        }
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nstages) stage(s);
    int cur = 0, nxt = STAGES - 1;
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) {
            if (wave < 7)
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s + STAGES - 1 < nstages) stage(nxt);
        compute(cur);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }

    // ---- accumulate: lane holds out-chan rows 32*kh + 16*i + 4*fg + j, element r*32 + 16*h + fr ----
    if (p.sqnorm) {   // per-sample norm pass: this block holds image blockIdx.x's complete gradient
        double sq = 0.0;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * rq + rr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (r == 7 || 16 * h + fr >= 28) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sq += (double)acc[i][rr][h][j] * (double)acc[i][rr][h][j];
            }
        }
        wave_sqnorm_add(sq, p.sqnorm + blockIdx.x);
        if (!p.ws) return;       // (ws set: the sample's tile is KEPT for the clipped sum, slot = image)
    }
    if (p.ws) {   // atomic-free path: the block's whole [64][256] slab (zeros in the padding) to ITS workspace slot
        float* o = p.ws + (long)blockIdx.x * (64 * 256);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * rq + rr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool live = r < 7 && 16 * h + fr < 28;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[(32 * kh + 16 * i + 4 * fg + j) * 256 + r * 32 + 16 * h + fr] = live ? acc[i][rr][h][j] : 0.f;
            }
        }
        return;
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * rq + rr;
        if (r == 7) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (16 * h + fr >= 28) continue;  // tap s = 7: no weight
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    unsafeAtomicAdd(p.dw + (long)(32 * kh + 16 * i + 4 * fg + j) * 256 + r * 32 + 16 * h + fr, acc[i][rr][h][j]);
        }
    }
}

static bool stem_wgrad_halo_ok(int N, int H, int W) {
    if (H % 32 != 0 || W % 32 != 0) return false;
    if ((long)N * (H + 6) * (W + 8) * 4 >= (1L << 31) || (long)N * (H / 2) * (W / 2) * 64 >= (1L << 31)) return false;
    return PRIMIA_OPT(stem_wgrad_halo) != 0;     // (0: the per-tap stem kernel of conv_wgrad.hip, for A/B)
}

// two blocks per CU (76 VGPRs, 69 KiB of LDS each): independent blocks cover each other's barriers and DMA waits
// (127 -> 113 us); every block flushes one [64][256] slab
static void stem_wgrad_geometry(int N, int H, int W, int& total, int& per_block, int& grid) {
    total = N * (H / 2 / 8) * (W / 2 / 16);
    const int target = PRIMIA_OPT(stem_wg_blocks) > 0 ? PRIMIA_OPT(stem_wg_blocks) : 512;
    long per = (total + target - 1) / target;
    if (per < 1) per = 1;
    per_block = (int)per;
    grid = (int)((total + per - 1) / per);
}

// the split of the 8 x 16 patches over blocks, for the fused backward kernel (stem_bwd_fused.hip), which must use the
// same one to produce the same slabs; false: shape not served
bool stem_wgrad_halo_blocks(int N, int H, int W, int* total, int* per_block, int* grid) {
    if (!stem_wgrad_halo_ok(N, H, W)) return false;
    stem_wgrad_geometry(N, H, W, *total, *per_block, *grid);
    return true;
}

size_t stem_wgrad_halo_ws_bytes(int N, int H, int W) {
    if (!stem_wgrad_halo_ok(N, H, W)) return 0;
    int total, per, grid;
    stem_wgrad_geometry(N, H, W, total, per, grid);
    return (size_t)grid * 64 * 256 * sizeof(float);
}

// PRIMIA_ERR_UNSUPPORTED -> caller uses the per-tap stem kernel of conv_wgrad.hip
int stem_wgrad_halo_dispatch(const bf16* xp, const bf16* dy, float* dw, int N, int H, int W, hipStream_t st, float* ws,
                             size_t ws_bytes, double* sqnorm) {
    if (!stem_wgrad_halo_ok(N, H, W)) return PRIMIA_ERR_UNSUPPORTED;
    StemWgParams p;
    p.xp = xp; p.dy = dy; p.dw = dw;
    p.N = N; p.Hp = H + 6; p.Wp = W + 8; p.Ho = H / 2; p.Wo = W / 2;
    p.PH = p.Ho / 8; p.PW = p.Wo / 16; p.PPI = p.PH * p.PW;
    int grid;
    stem_wgrad_geometry(N, H, W, p.total, p.per_block, grid);
    p.sqnorm = sqnorm;
    if (sqnorm) {          // one block per image
        p.per_block = p.PPI;
        grid = N;
    }
    const bool store = ws && ws_bytes >= (size_t)grid * 64 * 256 * sizeof(float);   // (norm pass: one slab per image)
    p.ws = store ? ws : nullptr;
    if (sqnorm && ws && !store) return PRIMIA_ERR_WORKSPACE;
    const size_t lds = (size_t)3 * (7 + 16) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)stem_conv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    stem_conv_wgrad_kernel<<<grid, 512, lds, st>>>(p);
    // the slabs are [k][e] tiles of the accumulator itself: one "tile" of 64 x 256, one tap, `grid` splits
    if (store && !sqnorm) wgrad_tile_reduce(ws, dw, grid, 1, 64, 256, 1, 1, 4, 256, 1, st);
    return launch_status();
}

}  // namespace primia
