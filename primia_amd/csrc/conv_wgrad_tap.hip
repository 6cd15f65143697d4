// Per-tap weight gradient of the stride-2 3x3 and the 1x1 convolutions (ResNet-18's transition blocks), bf16, gfx950:
//
//   dw[k, (r, s), c] = sum over output pixels m  dy[m, k] * x[pix(m, r, s), c]         (torchlib/models.py:272-275, 395-402:
//                                                                                        conv3x3(stride 2) / conv1x1 downsample)
//
// Second generation of conv_wgrad_dma_kernel (conv_wgrad.hip).  The instruction counters of round 3
// (profiles/r03_inst_mix.txt) showed what bound the first one: 25 M vector instructions per launch next to 1.8 M MFMAs —
// every DMA piece of every stage decoded its pixel with two reciprocal divisions and built a 64-bit pointer, every
// fragment read recomputed its swizzled address — i.e. ~40 us of pure VALU issue in a 75-us kernel, on top of a
// two-buffer loop whose builtin LDS-DMA made the compiler drain vmcnt before every fragment read.  Here
//   * one block = one (tap, out-channel tile, in-channel tile, pixel range), tiles of 256 x 128 (wide layers: 87 flop per
//     staged byte instead of 64) or 128 x 64 channels, 64 pixels per stage, v_mfma_f32_32x32x16_bf16;
//   * staging is buffer-addressed LDS-DMA with 32-bit lane offsets: dy rows are lane constant + stage origin, x rows
//     carry (n, ho, wo) counters that advance by 64 pixels per stage (adds and compares, no division), pixels outside
//     the image or the block's range get an offset beyond num_records and the hardware fills zeros;
//   * a pixel row is ROW bytes of 64-byte granules (32 channels = one MFMA fragment's channels); granule g of row r sits
//     at g ^ key(r) (key = r & 3, or bit 1 of r for 128-byte rows), so the four rows x 64 B one phase of a transposing
//     read touches cover all 64 banks; the key depends only on lane bits, so every fragment address is a lane constant
//     + an immediate k-step offset: (FM + FN) x 2 address registers per stage and no address arithmetic at the reads;
//   * 3-stage ring, one raw barrier per stage, counted vmcnt (inline-asm DMA: the compiler's wait-count pass does not
//     see it and leaves the fragment reads alone);
//   * the partial tile of every block goes to its own workspace slot, wgrad_tile_reduce adds the slots in split order
//     (deterministic, as every weight-gradient path of this library).
// The kernel stays bound by operand staging (a tap of a stride-2 layer shares nothing with the other taps): 48 KB per
// stage through a ~17-23 B/clk LDS-DMA path next to 1,024 cycles of MFMA work.
#include <stdlib.h>

#include "conv_wgrad.h"

namespace primia {

typedef float tap_f32x16 __attribute__((ext_vector_type(16)));
typedef int tap_i32x4 __attribute__((ext_vector_type(4)));

struct TapParams {
    const bf16* x;
    const bf16* dy;
    const bf16* dy2;    // paired launch: tap index `ntaps` = the block's 1x1 / stride-2 downsample (same x pixel as the
                        // centre tap of the 3x3, its own dy), or null
    float* ws;
    int H, W, C, K, S, stride, pad, Ho, Wo;
    int Md;
    int nkt, nct, ntaps, nsplit;
    int pps;            // pixels per split (multiple of 64)
    int dq, dr;         // 64 / Wo, 64 % Wo
    int slow;           // counters cannot be advanced with two conditional subtractions: divide every stage
    // per-sample norm pass (DP-SGD, template flag PS): a block walks whole images, each padded to `spi` stages of 64
    // pixels; after an image's last stage the squares of its tile are added to sqnorm[image] and the tile starts over
    double* sqnorm;
    int nimg, ipb, spi;
};

__device__ __forceinline__ void tap_bdma16(unsigned voff, tap_i32x4 rsrc, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ tap_i32x4 tap_rsrc(const void* base, long bytes) {
    const unsigned long long a = (unsigned long long)base;
    tap_i32x4 r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)(unsigned)(a >> 32) & 0xffff;       // stride 0: raw buffer
    r[2] = (int)(unsigned)(bytes > 0xfffffff0L ? 0xfffffff0L : bytes);
    r[3] = 0x00020000;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_readfirstlane(r[j]);
    return r;
}

template <int ROW>
__device__ __forceinline__ int tap_key(int row) {
    return ROW >= 256 ? (row & 3) : ((row >> 1) & 1);
}

template <int BMK, int BNC, int WM, int WN, bool PS = false>
__global__ __launch_bounds__(512) void conv_wgrad_tap_kernel(TapParams p) {
    constexpr int KP = 64, STAGES = 3;
    constexpr int ROW_A = BMK * 2, ROW_B = BNC * 2;
    constexpr int NA = KP * ROW_A / 1024, NX = KP * ROW_B / 1024;     // DMA pieces per stage
    constexpr int PA = NA / 8, PX = NX / 8;                           // ... per wave
    static_assert(NA % 8 == 0 && NX % 8 == 0 && WM * WN == 8, "piece / wave bookkeeping");
    constexpr int STAGE = KP * (ROW_A + ROW_B);
    constexpr int FM = BMK / WM / 32, FN = BNC / WN / 32;
    static_assert(FM >= 1 && FN >= 1, "tile too small");
    constexpr unsigned kOob = 0xfffffff0u;
    typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;
    typedef __attribute__((address_space(3))) char* ldsp_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % p.nct; bid /= p.nct;
    const int kt = bid % p.nkt; bid /= p.nkt;
    const int ntaps_all = p.ntaps + (p.dy2 ? 1 : 0);
    const int tap = bid % ntaps_all;
    const int split = bid / ntaps_all;
    const bool second = tap >= p.ntaps;          // block-uniform
    const int tr = second ? p.pad : tap / p.S, ts = second ? p.pad : tap - (tap / p.S) * p.S;
    const int ppi = p.Ho * p.Wo;                  // PS: pixels per image
    const int i0 = split * p.ipb;                 // PS: first image of this block
    int i1 = i0 + p.ipb;
    if (i1 > p.nimg) i1 = p.nimg;
    const int ms = PS ? i0 * ppi : split * p.pps;
    int me = ms + p.pps;
    if (me > p.Md) me = p.Md;
    const int nsteps = PS ? (i1 - i0) * p.spi : (me - ms + KP - 1) / KP;

    const tap_i32x4 rs_x = tap_rsrc(p.x, (long)(p.Md / (p.Ho * p.Wo)) * p.H * p.W * p.C * 2);
    const tap_i32x4 rs_dy = tap_rsrc(second ? p.dy2 : p.dy, (long)p.Md * p.K * 2);

    // ---- staging: lane constants ---------------------------------------------------------------------------------
    unsigned rel_a[PA];      // dy piece: byte offset of the lane's 16 bytes from the stage's first pixel
    int row_a[PA];
#pragma unroll
    for (int it = 0; it < PA; ++it) {
        const int a = wave + 8 * it;
        const int row = a * (1024 / ROW_A) + lane / (ROW_A / 16), c16 = lane % (ROW_A / 16);
        const int chunk = (((c16 >> 2) ^ tap_key<ROW_A>(row)) << 2) | (c16 & 3);
        row_a[it] = row;
        rel_a[it] = (unsigned)((row * p.K + kt * BMK + chunk * 8) * 2);
    }
    unsigned rel_b[PX];      // x piece: channel part of the offset
    int xm[PX], xw[PX], xh[PX], xn[PX];     // pixel of this lane's row: index, (wo, ho, n)
    int row_b[PX], xw0[PX], xh0[PX];        // PS: the row, and its (wo, ho) in the first stage of every image
#pragma unroll
    for (int it = 0; it < PX; ++it) {
        const int b = wave + 8 * it;
        const int row = b * (1024 / ROW_B) + lane / (ROW_B / 16), c16 = lane % (ROW_B / 16);
        const int chunk = (((c16 >> 2) ^ tap_key<ROW_B>(row)) << 2) | (c16 & 3);
        rel_b[it] = (unsigned)((ct * BNC + chunk * 8) * 2);
        const int m = PS ? row : ms + row;
        xm[it] = m;
        xw[it] = m % p.Wo;
        const int t = m / p.Wo;
        xh[it] = PS ? t : t % p.Ho;
        xn[it] = t / p.Ho;
        row_b[it] = row; xw0[it] = xw[it]; xh0[it] = xh[it];
    }
    const int hoff = tr - p.pad, woff = ts - p.pad;
    int s_issue = 0;        // stages issued so far
    int s_img = i0, s_sj = 0;   // PS: image and stage-in-image being issued
    auto stage = [&](int buf) {
        const unsigned base = lds0 + buf * STAGE;
        if constexpr (PS) {
            const int r0 = s_sj * KP;                       // first pixel-in-image of the stage
            const unsigned p0 = (unsigned)(s_img * ppi + r0);
#pragma unroll
            for (int it = 0; it < PA; ++it) {
                const unsigned voff = r0 + row_a[it] < ppi ? rel_a[it] + p0 * (unsigned)(p.K * 2) : kOob;
                tap_bdma16(voff, rs_dy, __builtin_amdgcn_readfirstlane(base + (wave + 8 * it) * 1024));
            }
#pragma unroll
            for (int it = 0; it < PX; ++it) {
                const int hs = xh[it] * p.stride + hoff, ws_ = xw[it] * p.stride + woff;
                const bool ok = r0 + row_b[it] < ppi && (unsigned)hs < (unsigned)p.H && (unsigned)ws_ < (unsigned)p.W;
                const unsigned pix = (unsigned)((s_img * p.H + hs) * p.W + ws_);
                const unsigned voff = ok ? pix * (unsigned)(p.C * 2) + rel_b[it] : kOob;
                tap_bdma16(voff, rs_x, __builtin_amdgcn_readfirstlane(base + KP * ROW_A + (wave + 8 * it) * 1024));
                int w = xw[it] + p.dr, h = xh[it] + p.dq;   // (rows past the image end are masked by the row test)
                if (w >= p.Wo) { w -= p.Wo; ++h; }
                xw[it] = w;
                xh[it] = h;
            }
            if (++s_sj == p.spi) {
                s_sj = 0;
                ++s_img;
#pragma unroll
                for (int it = 0; it < PX; ++it) { xw[it] = xw0[it]; xh[it] = xh0[it]; }
            }
            return;
        }
        const int p0 = ms + s_issue * KP;
        ++s_issue;
#pragma unroll
        for (int it = 0; it < PA; ++it) {
            const unsigned voff = p0 + row_a[it] < me ? rel_a[it] + (unsigned)p0 * (unsigned)(p.K * 2) : kOob;
            tap_bdma16(voff, rs_dy, __builtin_amdgcn_readfirstlane(base + (wave + 8 * it) * 1024));
        }
#pragma unroll
        for (int it = 0; it < PX; ++it) {
            const int hs = xh[it] * p.stride + hoff, ws_ = xw[it] * p.stride + woff;
            const bool ok = xm[it] < me && (unsigned)hs < (unsigned)p.H && (unsigned)ws_ < (unsigned)p.W;
            const unsigned pix = (unsigned)((xn[it] * p.H + hs) * p.W + ws_);
            const unsigned voff = ok ? pix * (unsigned)(p.C * 2) + rel_b[it] : kOob;
            tap_bdma16(voff, rs_x, __builtin_amdgcn_readfirstlane(base + KP * ROW_A + (wave + 8 * it) * 1024));
            // next stage: 64 pixels further
            xm[it] += KP;
            if (p.slow) {
                const int m = xm[it];
                xw[it] = m % p.Wo;
                const int t = m / p.Wo;
                xh[it] = t % p.Ho;
                xn[it] = t / p.Ho;
            } else {
                int w = xw[it] + p.dr, h = xh[it] + p.dq;
                if (w >= p.Wo) { w -= p.Wo; ++h; }
                if (h >= p.Ho) { h -= p.Ho; ++xn[it]; }
                if (h >= p.Ho) { h -= p.Ho; ++xn[it]; }
                xw[it] = w;
                xh[it] = h;
            }
        }
    };

    tap_f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addresses: lane (16-lane group g16, fr) reads pixel kk = 8*(g16>>1) + (fr>>2) (+4) of the k-step,
    //      channels 16*(g16&1) + 4*(fr&3) .. +3 of the fragment's 32; the swizzle key of its row is a lane constant ------
    const int fr = lane & 15, g16 = lane >> 4;
    const int cbyte = (16 * (g16 & 1) + 4 * (fr & 3)) * 2;
    int offa[FM][2], offb[FN][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kk = 8 * (g16 >> 1) + (fr >> 2) + 4 * h;
#pragma unroll
        for (int i = 0; i < FM; ++i)
            offa[i][h] = kk * ROW_A + (((wm * FM + i) ^ tap_key<ROW_A>(kk)) << 6) + cbyte;
#pragma unroll
        for (int j = 0; j < FN; ++j)
            offb[j][h] = KP * ROW_A + kk * ROW_B + (((wn * FN + j) ^ tap_key<ROW_B>(kk)) << 6) + cbyte;
    }

    auto compute = [&](int buf) {
        const ldsp_t sb = (ldsp_t)(size_t)(lds0 + buf * STAGE);
        ldsp_t pa[FM][2], pb[FN][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < FM; ++i) pa[i][h] = sb + offa[i][h];
#pragma unroll
            for (int j = 0; j < FN; ++j) pb[j][h] = sb + offb[j][h];
        }
#pragma unroll
        for (int ks = 0; ks < KP / 16; ++ks) {
            bf16x8_t a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pa[i][0] + ks * 16 * ROW_A));
                bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pa[i][1] + ks * 16 * ROW_A));
                a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pb[j][0] + ks * 16 * ROW_B));
                bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(pb[j][1] + ks * 16 * ROW_B));
                b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    auto wait_pieces = [&](int n) {      // wave-uniform: at most n of this wave's DMA pieces may still be in flight
        switch (n) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        }
    };
    static_assert(PA + PX == 3 || PA + PX == 6, "wait_pieces cases");

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nsteps) stage(s);
    int cur = 0, nxt = STAGES - 1;
    int c_img = i0, c_sj = 0;     // PS: image and stage-in-image being multiplied
    for (int s = 0; s < nsteps; ++s) {
        wait_pieces(s + 1 < nsteps ? PA + PX : 0);      // stage s has landed; stage s + 1 may be in flight
        __builtin_amdgcn_s_barrier();                    // ... for every wave; buffer `nxt` is no longer read
        if (s + STAGES - 1 < nsteps) stage(nxt);
        compute(cur);
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        if constexpr (PS) {
            if (++c_sj == p.spi) {     // the image's complete (tap, kt, ct) tile: its squares, then start over
                double sq = 0.0;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            sq += (double)acc[i][j][e] * (double)acc[i][j][e];
                            acc[i][j][e] = 0.f;
                        }
                wave_sqnorm_add(sq, p.sqnorm + c_img);
                c_sj = 0;
                ++c_img;
            }
        }
    }
    if constexpr (PS) return;

    // ---- this block's BMK x BNC partial tile -> ITS workspace slot, row-major [k_local][c_local] ------------------
    // lane holds rows 32*(wm*FM + i) + 8*m + 4*(lane >> 5) + t (register 4*m + t), column 32*(wn*FN + j) + (lane & 31)
    float* o = p.ws + ((long)(((tap * p.nkt + kt) * p.nct + ct)) * p.nsplit + split) * (BMK * BNC);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    o[(32 * (wm * FM + i) + 8 * m + 4 * (lane >> 5) + t) * BNC + 32 * (wn * FN + j) + (lane & 31)] =
                        acc[i][j][4 * m + t];
}

struct TapGeom {
    bool ok, wide;
    int nkt, nct, nsplit, pps, combos, BMK, BNC;
};

// extra_taps: taps of a paired layer that ride in the same launch (they count for the one-round block budget)
static TapGeom tap_geom(const WgradParams& w, int extra_taps = 0) {
    TapGeom g{};
    const bool off = !PRIMIA_OPT(wgtap);
    g.ok = !off && !w.persample && !w.xpad && (w.stride == 2 || (w.R == 1 && w.S == 1)) && w.ntaps == w.R * w.S &&
           w.Md < (1L << 24) && (long)w.N * w.H * w.W * w.C < (1L << 31) && w.Md * w.K < (1L << 31) && w.K % 128 == 0 &&
           w.C % 64 == 0;
    if (!g.ok) return g;
    g.wide = w.K % 256 == 0 && w.C % 128 == 0;
    g.BMK = g.wide ? 256 : 128;
    g.BNC = g.wide ? 128 : 64;
    g.nkt = w.K / g.BMK;
    g.nct = w.C / g.BNC;
    g.combos = w.ntaps * g.nkt * g.nct;
    // one round of blocks: one per CU for the wide tile (144 KiB of LDS), two for the narrow one (72 KiB)
    const int tb = PRIMIA_OPT(wgtap_blocks);
    const int target = tb ? tb : (g.wide ? 256 : 512);
    long want = target / ((w.ntaps + extra_taps) * g.nkt * g.nct);
    if (want < 1) want = 1;
    const long max_split = (w.Md + 8 * 64 - 1) / (8 * 64);      // at least 8 stages per block
    if (want > max_split) want = max_split;
    long pps = (w.Md + want - 1) / want;
    pps = (pps + 63) / 64 * 64;
    g.pps = (int)pps;
    g.nsplit = (int)((w.Md + pps - 1) / pps);
    return g;
}

size_t wgrad_tap_ws_bytes(const WgradParams& w) {
    const TapGeom g = tap_geom(w);
    if (!g.ok) return 0;
    return (size_t)g.combos * g.nsplit * g.BMK * g.BNC * sizeof(float);
}

// 17 = conv_wgrad_tap_kernel; 0: shape not served
int wgrad_tap_kernel_id(const WgradParams& w) { return tap_geom(w).ok ? 17 : 0; }

template <int BMK, int BNC, int WM, int WN>
static int launch_tap(const WgradParams& w, const TapGeom& g, hipStream_t st) {
    TapParams p{};
    p.x = (const bf16*)w.x; p.dy = (const bf16*)w.dy; p.dy2 = nullptr; p.ws = w.ws;
    p.H = w.H; p.W = w.W; p.C = w.C; p.K = w.K; p.S = w.S; p.stride = w.stride; p.pad = w.pad; p.Ho = w.Ho; p.Wo = w.Wo;
    p.Md = (int)w.Md;
    p.nkt = g.nkt; p.nct = g.nct; p.ntaps = w.ntaps; p.nsplit = g.nsplit; p.pps = g.pps;
    p.dq = 64 / w.Wo; p.dr = 64 % w.Wo;
    p.slow = (p.dq + 1 > 2 * w.Ho) ? 1 : 0;
    const int lds = 3 * 64 * (BMK + BNC) * 2;
    auto kern = conv_wgrad_tap_kernel<BMK, BNC, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    kern<<<g.combos * g.nsplit, 512, lds, st>>>(p);
    wgrad_tile_reduce(w.ws, w.dw, g.nsplit, g.combos, BMK, BNC, g.nkt, g.nct, w.C, w.klen, 0, st);
    return launch_status();
}

// conv1 (3x3 / stride 2) and the downsample (1x1 / stride 2) of a transition block in ONE launch: both read the same
// x — the downsample's pixel (2 ho, 2 wo) is the 3x3's centre tap — so it is simply a tenth tap with its own dy.
// Workspace: (9 + 1) * nkt * nct * nsplit tiles; two ordered reductions (into dw and dw2).
size_t wgrad_tap_pair_ws_bytes(const WgradParams& w, const WgradParams& w2) {
    const TapGeom g = tap_geom(w, 1);
    const bool pair = g.ok && tap_geom(w2).ok && w.R == 3 && w.S == 3 && w.pad == 1 && w.stride == 2 && w2.R == 1 &&
                      w2.S == 1 && w2.pad == 0 && w2.stride == 2 && w2.K == w.K && w2.C == w.C && w2.Md == w.Md &&
                      w2.H == w.H && w2.W == w.W;
    if (!pair) return 0;
    return (size_t)(g.combos + g.nkt * g.nct) * g.nsplit * g.BMK * g.BNC * sizeof(float);
}

template <int BMK, int BNC, int WM, int WN>
static int launch_tap_pair(const WgradParams& w, const WgradParams& w2, const TapGeom& g, hipStream_t st) {
    TapParams p{};
    p.x = (const bf16*)w.x; p.dy = (const bf16*)w.dy; p.dy2 = (const bf16*)w2.dy; p.ws = w.ws;
    p.H = w.H; p.W = w.W; p.C = w.C; p.K = w.K; p.S = w.S; p.stride = w.stride; p.pad = w.pad; p.Ho = w.Ho; p.Wo = w.Wo;
    p.Md = (int)w.Md;
    p.nkt = g.nkt; p.nct = g.nct; p.ntaps = w.ntaps; p.nsplit = g.nsplit; p.pps = g.pps;
    p.dq = 64 / w.Wo; p.dr = 64 % w.Wo;
    p.slow = (p.dq + 1 > 2 * w.Ho) ? 1 : 0;
    const int lds = 3 * 64 * (BMK + BNC) * 2;
    auto kern = conv_wgrad_tap_kernel<BMK, BNC, WM, WN>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return PRIMIA_ERR_LAUNCH;
    const int combos2 = g.nkt * g.nct;
    kern<<<(g.combos + combos2) * g.nsplit, 512, lds, st>>>(p);
    // (both filters' tiles in one reduce launch: the downsample's sit behind conv1's in the workspace, same split count and
    // tile shape, w2.C == w.C)
    wgrad_tile_reduce_pair(w.ws, w.dw, w2.dw, g.nsplit, g.combos, combos2, BMK, BNC, g.nkt, g.nct, w.C, w.klen, w2.klen, st);
    return launch_status();
}

int wgrad_tap_pair_dispatch(const WgradParams& w, const WgradParams& w2, hipStream_t st) {
    const size_t need = wgrad_tap_pair_ws_bytes(w, w2);
    if (!need || !w.ws || !w.dw || !w2.dw || w.ws_bytes < need) return PRIMIA_ERR_UNSUPPORTED;
    TapGeom g = tap_geom(w, 1);
    return g.wide ? launch_tap_pair<256, 128, 2, 4>(w, w2, g, st) : launch_tap_pair<128, 64, 4, 2>(w, w2, g, st);
}

// DP-SGD norm pass: sqnorm[n] += ||dW_n||_F^2 of a stride-2 / 1x1 layer, per-sample gradients never written.  The
// older kernels ran one block per (tile, tap, image) — at 49 or 196 pixels per image that is all prologue.
template <int BMK, int BNC, int WM, int WN>
static int launch_tap_persample(const WgradParams& w, const TapGeom& g, hipStream_t st) {
    TapParams p{};
    p.x = (const bf16*)w.x; p.dy = (const bf16*)w.dy; p.dy2 = nullptr; p.ws = nullptr;
    p.H = w.H; p.W = w.W; p.C = w.C; p.K = w.K; p.S = w.S; p.stride = w.stride; p.pad = w.pad; p.Ho = w.Ho; p.Wo = w.Wo;
    p.Md = (int)w.Md;
    p.nkt = g.nkt; p.nct = g.nct; p.ntaps = w.ntaps;
    p.dq = 64 / w.Wo; p.dr = 64 % w.Wo;
    p.slow = 0;
    p.sqnorm = w.sqnorm;
    p.nimg = w.N;
    p.spi = (w.Ho * w.Wo + 63) / 64;
    long want = (g.wide ? 256 : 512) / g.combos;
    if (want < 1) want = 1;
    if (want > w.N) want = w.N;
    p.ipb = (int)((w.N + want - 1) / want);
    p.nsplit = (w.N + p.ipb - 1) / p.ipb;
    p.pps = 0;
    const int lds = 3 * 64 * (BMK + BNC) * 2;
    auto kern = conv_wgrad_tap_kernel<BMK, BNC, WM, WN, true>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return PRIMIA_ERR_LAUNCH;
    kern<<<g.combos * p.nsplit, 512, lds, st>>>(p);
    return launch_status();
}

// 26 = conv_wgrad_tap_kernel's norm pass (whole images per block), 0 = shape not served
int wgrad_tap_persample_kernel_id(const WgradParams& w) {
    const bool off = !PRIMIA_OPT(wgtap_persample);
    WgradParams b = w;
    b.persample = 0;
    return (off || !tap_geom(b).ok || w.Ho * w.Wo < 1) ? 0 : 26;
}

int wgrad_tap_persample_dispatch(const WgradParams& w, hipStream_t st) {
    const bool off = !PRIMIA_OPT(wgtap_persample);
    if (off || !w.persample || !w.sqnorm) return PRIMIA_ERR_UNSUPPORTED;
    WgradParams b = w;
    b.persample = 0;
    const TapGeom g = tap_geom(b);
    if (!g.ok || w.Ho * w.Wo < 1) return PRIMIA_ERR_UNSUPPORTED;
    return g.wide ? launch_tap_persample<256, 128, 2, 4>(w, g, st) : launch_tap_persample<128, 64, 4, 2>(w, g, st);
}

// PRIMIA_ERR_UNSUPPORTED: shape not served, or no (large enough) workspace — the caller falls back to the older kernels
int wgrad_tap_dispatch(const WgradParams& w, hipStream_t st) {
    const TapGeom g = tap_geom(w);
    if (!g.ok || !w.ws || !w.dw || w.ws_bytes < (size_t)g.combos * g.nsplit * g.BMK * g.BNC * sizeof(float))
        return PRIMIA_ERR_UNSUPPORTED;
    return g.wide ? launch_tap<256, 128, 2, 4>(w, g, st) : launch_tap<128, 64, 4, 2>(w, g, st);
}

}  // namespace primia
