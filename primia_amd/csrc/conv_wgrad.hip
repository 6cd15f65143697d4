// Weight-gradient convolution on gfx950 MFMA.
//
//   dw[k, (r,s,c)] += sum_m dy[m, k] * x[pix(m, r, s), c]        m = (n, ho, wo)
//
// GEMM view per kernel tap: D[64..128 out-chan][64..128 in-chan], reduction over pixels.  Both
// operands are stored pixel-major (NHWC), i.e. the reduction index is the SLOW axis of both, so
// the MFMA fragments (8 consecutive reduction elements per lane) need a transpose:
//   bf16: tiles are staged to LDS as they are ([pixel][channel]) and read back with
//         ds_read_b64_tr_b16 — each 16-lane group reads a 4-pixel x 16-channel block and every
//         lane receives the 4 pixels of ITS channel; two reads give the 8 pixels of one MFMA
//         operand (v_mfma_f32_16x16x32_bf16);
//   f32 : v_mfma_f32_16x16x4_f32 takes one element per lane, read directly with ds_read_b32.
// The pixel range is split over blocks (split-K); partial results are accumulated with fp32
// atomics into a zeroed [K][klen] buffer (fwd weight layout, see conv_common.h).
#include <stdlib.h>

#include "conv_wgrad.h"

namespace primia {

// option wgrad_kernel: 0 = by shape, else the per-tap kernel named: 'o' register-staged, 'd' LDS-DMA, 't' either by shape
static inline char wgrad_force() {
    const int v = PRIMIA_OPT(wgrad_kernel);
    return v == 1 ? 'o' : v == 2 ? 'd' : v == 3 ? 't' : 0;
}


template <typename T, int BMK, int BNC, bool STEM>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    constexpr int ES = sizeof(T);
    constexpr int KP = (ES == 2) ? 64 : 32;  // pixels per step
    constexpr int CH = 16 / ES;              // elements per 16-B chunk
    constexpr int ROW_A = BMK * ES + 16;     // LDS row strides (bytes), padded
    constexpr int ROW_B = BNC * ES + 16;
    constexpr int CPR_A = BMK / CH, CPR_B = BNC / CH;  // chunks per row
    constexpr int NA = (KP * CPR_A + 255) / 256;       // staged chunks per thread
    constexpr int NB = (KP * CPR_B + 255) / 256;
    constexpr int FM = BMK / 32, FN = BNC / 32;        // fragments per wave (2x2 waves)
    static_assert(FN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * KP * (ROW_A + ROW_B) bytes
    auto lds_a = [&](int buf) { return smem + buf * KP * (ROW_A + ROW_B); };
    auto lds_b = [&](int buf) { return smem + buf * KP * (ROW_A + ROW_B) + KP * ROW_A; };

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % p.nct; bid /= p.nct;
    const int kt = bid % p.nkt; bid /= p.nkt;
    const int tap = bid % p.ntaps;
    const int split = bid / p.ntaps;
    const int tr = STEM ? tap : tap / p.S;
    const int ts = STEM ? 0 : tap - tr * p.S;

    const long ms = (long)split * p.pix_per_split;
    long me = ms + p.pix_per_split;
    if (me > p.Md) me = p.Md;
    const int nsteps = (int)((me - ms + KP - 1) / KP);

    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ dy = (const T*)p.dy;

    // ---- staging assignments -------------------------------------------------------------------
    // A (dy): chunk id q = tid + 256*j -> pixel row q / CPR_A, chunk q % CPR_A.
    // B (x):  same with CPR_B; needs the pixel's (n, ho, wo), tracked incrementally per step.
    int b_row[NB], b_chunk[NB], b_wo[NB], b_ho[NB], b_n[NB];
    bool b_use[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = tid + 256 * j;
        b_use[j] = q < KP * CPR_B;
        b_row[j] = q / CPR_B;
        b_chunk[j] = q % CPR_B;
        long m = ms + b_row[j];
        if (m >= p.Md) m = p.Md - 1;
        b_wo[j] = (int)(m % p.Wo);
        long t = m / p.Wo;
        b_ho[j] = (int)(t % p.Ho);
        b_n[j] = (int)(t / p.Ho);
    }

    u32x4 ra[NA], rb[NB];

    auto load_global = [&](int step) {
        const long mb = ms + (long)step * KP;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int q = tid + 256 * j;
            const int row = q / CPR_A, chunk = q % CPR_A;
            const long m = mb + row;
            u32x4 v = {0, 0, 0, 0};
            if (q < KP * CPR_A && m < me) v = *(const u32x4*)(dy + m * p.K + kt * BMK + chunk * CH);
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const long m = mb + b_row[j];
            const bool ok = b_use[j] && m < me;
            const int hs = b_ho[j] * p.stride - p.pad + tr;
            u32x4 v = {0, 0, 0, 0};
            if (STEM) {
                // "channel" axis = 32 elements = 8 consecutive input pixels x 4 channels.
                // The 8th pixel of a row group (s == 7) belongs to no kernel tap: it is loaded as zero so
                // the padded accumulator slots stay exactly 0 (per-sample norms sum the whole slab).
                const int e0 = b_chunk[j] * CH;
                const int sx = e0 >> 2;
                if (p.xpad) {
                    // padded input: tap (tr, sx) of output pixel (ho, wo) is padded pixel (2*ho + tr, 2*wo + sx)
                    const long base = (((long)b_n[j] * (p.H + 6) + b_ho[j] * 2 + tr) * (p.W + 8) + b_wo[j] * 2 + sx) * 4;
                    if (ES == 4) {
                        if (ok && sx < p.S) v = *(const u32x4*)(x + base);
                    } else {
                        u32x2 a = {0, 0}, b = {0, 0};
                        if (ok && sx < p.S) a = *(const u32x2*)(x + base);
                        if (ok && sx + 1 < p.S) b = *(const u32x2*)(x + base + 4);
                        v = u32x4{a[0], a[1], b[0], b[1]};
                    }
                } else {
                const int ws = b_wo[j] * 2 - 3 + sx;
                const bool rowok = ok && hs >= 0 && hs < p.H;
                const long base = (((long)b_n[j] * p.H + hs) * p.W + ws) * 4;
                if (ES == 4) {
                    if (rowok && sx < p.S && ws >= 0 && ws < p.W) v = *(const u32x4*)(x + base);
                } else {
                    u32x2 a = {0, 0}, b = {0, 0};
                    if (rowok && sx < p.S && ws >= 0 && ws < p.W) a = *(const u32x2*)(x + base);
                    if (rowok && sx + 1 < p.S && ws + 1 >= 0 && ws + 1 < p.W) b = *(const u32x2*)(x + base + 4);
                    v = u32x4{a[0], a[1], b[0], b[1]};
                }
                }
            } else {
                const int ws = b_wo[j] * p.stride - p.pad + ts;
                if (ok && hs >= 0 && hs < p.H && ws >= 0 && ws < p.W)
                    v = *(const u32x4*)(x + (((long)b_n[j] * p.H + hs) * p.W + ws) * p.C + ct * BNC + b_chunk[j] * CH);
            }
            rb[j] = v;
            // advance this row's pixel by KP for the next step
            b_wo[j] += KP;
            while (b_wo[j] >= p.Wo) {
                b_wo[j] -= p.Wo;
                if (++b_ho[j] == p.Ho) {
                    b_ho[j] = 0;
                    ++b_n[j];
                }
            }
        }
    };

    auto store_lds = [&](int buf) {
        char* la = lds_a(buf);
        char* lb = lds_b(buf);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int q = tid + 256 * j;
            if (q < KP * CPR_A) *(u32x4*)(la + (q / CPR_A) * ROW_A + (q % CPR_A) * 16) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (b_use[j]) *(u32x4*)(lb + b_row[j] * ROW_B + b_chunk[j] * 16) = rb[j];
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int ca0 = wm * (BMK / 2), cb0 = wn * (BNC / 2);  // wave's channel offsets in the tiles

    auto compute = [&](int buf) {
        const char* la = lds_a(buf);
        const char* lb = lds_b(buf);
        if constexpr (ES == 2) {
            // transpose-read: lane L of a 16-lane group supplies the address of pixel (L >> 2),
            // channels 4*(L & 3)..+3 of the 4x16 block and receives the 4 pixels of channel L.
            const int tp = fr >> 2, tc = (fr & 3) * 4;
#pragma unroll
            for (int kk = 0; kk < KP / 32; ++kk) {
                bf16x8_t a[FM], b[FN];
                const int prow = kk * 32 + fg * 8 + tp;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const char* q = la + prow * ROW_A + (ca0 + 16 * i + tc) * 2;
                    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)q);
                    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(q + 4 * ROW_A));
                    a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const char* q = lb + prow * ROW_B + (cb0 + 16 * j + tc) * 2;
                    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)q);
                    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4_t*)(q + 4 * ROW_B));
                    b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < KP / 4; ++t) {
                float a[FM], b[FN];
                const int prow = t * 4 + fg;
#pragma unroll
                for (int i = 0; i < FM; ++i) a[i] = *(const float*)(la + prow * ROW_A + (ca0 + 16 * i + fr) * 4);
#pragma unroll
                for (int j = 0; j < FN; ++j) b[j] = *(const float*)(lb + prow * ROW_B + (cb0 + 16 * j + fr) * 4);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    if (nsteps > 0) {
        load_global(0);
        store_lds(0);
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            const bool more = step + 1 < nsteps;
            if (more) load_global(step + 1);
            compute(cur);
            if (more) store_lds(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- accumulate: lane holds rows (out-chan) fg*4+reg, column (in-chan) fr ---------------------
    if (p.sqnorm) {  // per-sample norm pass: this block's tile is the sample's complete gradient tile
        double sq = 0.0;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sq += (double)acc[i][j][r] * (double)acc[i][j][r];
        wave_sqnorm_add(sq, p.sqnorm + split);
        return;
    }
    if (p.ws) {
        // atomic-free path: this block's BMK x BNC partial tile goes to ITS slot of the workspace, row-major
        // [k_local][c_local]; wgrad_tile_reduce_kernel adds the slots of a (tap, kt, ct) tile in split order
        float* o = p.ws + ((long)(((tap * p.nkt + kt) * p.nct + ct)) * p.nsplit + split) * (BMK * BNC);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[(ca0 + 16 * i + fg * 4 + r) * BNC + cb0 + 16 * j + fr] = acc[i][j][r];
        return;
    }
    const int ebase = STEM ? tr * 32 : tap * p.C + ct * BNC;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kt * BMK + ca0 + 16 * i + fg * 4 + r;
                const int e = ebase + cb0 + 16 * j + fr;
                unsafeAtomicAdd(p.dw + (long)split * p.split_stride + (long)k * p.klen + e, acc[i][j][r]);
            }
}

// dw[kt*BMK + k][ebase(tap, ct) + c] = sum over splits of the partial tiles, in split order (deterministic).
// Block = CL float4 chunks x SL split lanes (CL * SL = 256), as wgrad_patch_reduce_kernel.
template <int SL>
__global__ __launch_bounds__(256) void wgrad_tile_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                int nsplit, int BMK, int BNC, int nkt, int nct, int C,
                                                                int klen, int stem, const float* __restrict__ wgt,
                                                                int combos1 = 0x7fffffff, float* __restrict__ dw2 = nullptr,
                                                                int klen2 = 0) {
    // (combos1, dw2, klen2: a transition block's pair — the tiles from combos1 on are the 1x1 downsample's, a second filter
    // with its own row length; one launch reduces both, round 6)
    constexpr int CL = 256 / SL;
    __shared__ f32x4 red[SL][CL];
    const int tile4 = BMK * BNC / 4;                // float4 chunks per tile
    const int cpb = (tile4 + CL - 1) / CL;          // blocks per tile
    const int combo = blockIdx.x / cpb;
    const int q = (blockIdx.x % cpb) * CL + (threadIdx.x % CL);
    const int sl = threadIdx.x / CL;
    const bool live = q < tile4;
    const float* src = ws + (long)combo * nsplit * (BMK * BNC) + (live ? q : 0) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (live && wgt) {      // DP-SGD: split s = sample s, weighted by its clip factor (same order)
        for (int s = sl; s < nsplit; s += SL) a += *(const f32x4*)(src + (long)s * (BMK * BNC)) * wgt[s];
    } else if (live) {
        int s = sl;
#pragma unroll 1
        for (; s + 3 * SL < nsplit; s += 4 * SL) {
            const f32x4 v0 = *(const f32x4*)(src + (long)s * (BMK * BNC));
            const f32x4 v1 = *(const f32x4*)(src + (long)(s + SL) * (BMK * BNC));
            const f32x4 v2 = *(const f32x4*)(src + (long)(s + 2 * SL) * (BMK * BNC));
            const f32x4 v3 = *(const f32x4*)(src + (long)(s + 3 * SL) * (BMK * BNC));
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; s < nsplit; s += SL) a += *(const f32x4*)(src + (long)s * (BMK * BNC));
    }
    if (SL > 1) {
        red[sl][threadIdx.x % CL] = a;
        __syncthreads();
        if (sl != 0) return;
#pragma unroll
        for (int k = 1; k < SL; ++k) a += red[k][threadIdx.x % CL];
    }
    if (!live) return;
    const bool second = combo >= combos1;
    const int cb = second ? combo - combos1 : combo;
    const int ct = cb % nct, kt = (cb / nct) % nkt, tap = cb / (nct * nkt);
    const int kl = (q * 4) / BNC, cl = (q * 4) % BNC;
    const int ebase = stem ? tap * 32 : tap * C + ct * BNC;
    *(f32x4*)((second ? dw2 : dw) + (long)(kt * BMK + kl) * (second ? klen2 : klen) + ebase + cl) = a;
}

void wgrad_tile_reduce(const float* ws, float* dw, int nsplit, int combos, int BMK, int BNC, int nkt, int nct, int C,
                       int klen, int stem, hipStream_t st, const float* wgt) {
    const int tile4 = BMK * BNC / 4;
    if (nsplit >= 32)
        wgrad_tile_reduce_kernel<16><<<combos * ((tile4 + 15) / 16), 256, 0, st>>>(ws, dw, nsplit, BMK, BNC, nkt, nct, C,
                                                                                 klen, stem, wgt);
    else if (nsplit >= 4)
        wgrad_tile_reduce_kernel<4><<<combos * ((tile4 + 63) / 64), 256, 0, st>>>(ws, dw, nsplit, BMK, BNC, nkt, nct, C,
                                                                                klen, stem, wgt);
    else
        wgrad_tile_reduce_kernel<1><<<combos * ((tile4 + 255) / 256), 256, 0, st>>>(ws, dw, nsplit, BMK, BNC, nkt, nct, C,
                                                                                  klen, stem, wgt);
}

// conv1 (combos1 tiles) + downsample (combos2 tiles, stored behind them) of a transition block in ONE launch
void wgrad_tile_reduce_pair(const float* ws, float* dw, float* dw2, int nsplit, int combos1, int combos2, int BMK, int BNC,
                            int nkt, int nct, int C, int klen, int klen2, hipStream_t st) {
    const int tile4 = BMK * BNC / 4, combos = combos1 + combos2;
    if (nsplit >= 32)
        wgrad_tile_reduce_kernel<16><<<combos * ((tile4 + 15) / 16), 256, 0, st>>>(ws, dw, nsplit, BMK, BNC, nkt, nct, C, klen, 0,
                                                                                 nullptr, combos1, dw2, klen2);
    else if (nsplit >= 4)
        wgrad_tile_reduce_kernel<4><<<combos * ((tile4 + 63) / 64), 256, 0, st>>>(ws, dw, nsplit, BMK, BNC, nkt, nct, C, klen, 0,
                                                                                nullptr, combos1, dw2, klen2);
    else
        wgrad_tile_reduce_kernel<1><<<combos * ((tile4 + 255) / 256), 256, 0, st>>>(ws, dw, nsplit, BMK, BNC, nkt, nct, C, klen, 0,
                                                                                  nullptr, combos1, dw2, klen2);
}

static void launch_tile_reduce(const WgradParams& p, int BMK, int BNC, int stem, hipStream_t st) {
    wgrad_tile_reduce(p.ws, p.dw, p.nsplit, p.ntaps * p.nkt * p.nct, BMK, BNC, p.nkt, p.nct, p.C, p.klen, stem, st);
}

// split geometry of the per-tap kernel (shared by the launcher and the workspace query)
template <int ES, int BMK, int BNC, bool STEM>
static void wgrad_geometry(WgradParams& p) {
    constexpr int KP = (ES == 2) ? 64 : 32;
    p.nkt = p.K / BMK;
    p.nct = STEM ? 1 : p.C / BNC;
    const int combos = p.ntaps * p.nkt * p.nct;
    // enough blocks to fill 256 CUs a few times over, but at least 8 steps per block
    const int target_blocks = PRIMIA_OPT(wgt_blocks) > 0 ? PRIMIA_OPT(wgt_blocks) : 4 * 256;
    long want = (target_blocks + combos - 1) / combos;
    long max_split = (p.Md + 8 * KP - 1) / (8 * KP);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    long pps = (p.Md + want - 1) / want;
    pps = (pps + KP - 1) / KP * KP;
    if (p.persample) pps = (long)p.Ho * p.Wo;  // one split per image
    p.pix_per_split = pps;
    p.nsplit = (int)((p.Md + pps - 1) / pps);
    p.split_stride = p.persample ? (long)p.K * p.klen : 0;
}

template <typename T, int BMK, int BNC, bool STEM>
static size_t wgrad_ws_need(WgradParams p) {
    wgrad_geometry<sizeof(T), BMK, BNC, STEM>(p);
    return (size_t)p.ntaps * p.nkt * p.nct * p.nsplit * BMK * BNC * sizeof(float);
}

template <typename T, int BMK, int BNC, bool STEM>
static int launch_wgrad(WgradParams p, hipStream_t st) {
    constexpr int KP = (sizeof(T) == 2) ? 64 : 32;
    wgrad_geometry<sizeof(T), BMK, BNC, STEM>(p);
    const int combos = p.ntaps * p.nkt * p.nct;
    const bool store = !p.persample && p.ws && p.ws_bytes >= (size_t)combos * p.nsplit * BMK * BNC * sizeof(float);
    if (!store) p.ws = nullptr;
    const int grid = combos * p.nsplit;
    const size_t lds = 2 * KP * ((BMK + BNC) * sizeof(T) + 32);
    auto kern = conv_wgrad_kernel<T, BMK, BNC, STEM>;
    static bool attr_set = false;  // once per instantiation (not per launch: keeps graph capture legal)
    if (lds > 48 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    kern<<<grid, 256, lds, st>>>(p);
    if (store) launch_tile_reduce(p, BMK, BNC, STEM ? 1 : 0, st);
    return launch_status();
}

}  // namespace primia

using namespace primia;

static int conv2d_wgrad_impl(const primia_conv_desc* d, const void* x, const void* dy, float* dw_acc, int persample,
                             int dtype, primia_stream_t stream, double* sqnorm = nullptr, float* ws = nullptr,
                             size_t ws_bytes = 0);

static int stem_conv_wgrad_impl(const void* x_padded, const void* dy, float* dw_acc, float* ws, size_t ws_bytes, int N,
                                int H, int W, int dtype, primia_stream_t stream);

extern "C" int primia_stem_conv_wgrad(const void* x_padded, const void* dy, float* dw_acc, int N, int H, int W, int dtype,
                                      primia_stream_t stream) {
    return stem_conv_wgrad_impl(x_padded, dy, dw_acc, nullptr, 0, N, H, W, dtype, stream);
}

// DP-SGD norm pass for conv1 on the padded input: sqnorm[n] += ||dW_n||_F^2 (see primia_conv2d_wgrad_persample_sqnorm);
// PRIMIA_ERR_UNSUPPORTED where the halo kernel does not serve the shape (caller uses the generic form on the
// unpadded input)
extern "C" int primia_stem_conv_wgrad_persample_sqnorm(const void* x_padded, const void* dy, double* sqnorm, int N, int H,
                                                       int W, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(x_padded && dy && sqnorm && N > 0 && H > 0 && W > 0);
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_UNSUPPORTED;
    return stem_wgrad_halo_dispatch((const bf16*)x_padded, (const bf16*)dy, nullptr, N, H, W, (hipStream_t)stream, nullptr,
                                    0, sqnorm);
}

extern "C" int64_t primia_stem_conv_wgrad_ws_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return PRIMIA_ERR_ARG;
    return (int64_t)stem_wgrad_halo_ws_bytes(N, H, W);
}

extern "C" int primia_stem_conv_wgrad_ws(const void* x_padded, const void* dy, float* dw_acc, void* ws, int64_t ws_bytes,
                                         int N, int H, int W, int dtype, primia_stream_t stream) {
    return stem_conv_wgrad_impl(x_padded, dy, dw_acc, (float*)ws, ws_bytes > 0 ? (size_t)ws_bytes : 0, N, H, W, dtype,
                                stream);
}

static int stem_conv_wgrad_impl(const void* x_padded, const void* dy, float* dw_acc, float* ws, size_t ws_bytes, int N,
                                int H, int W, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(x_padded && dy && dw_acc && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0);
    WgradParams p;
    p.x = x_padded; p.dy = dy; p.dw = dw_acc;
    p.N = N; p.H = H; p.W = W; p.C = 4; p.K = 64; p.R = 7; p.S = 7; p.stride = 2; p.pad = 3;
    p.Ho = H / 2; p.Wo = W / 2;
    p.klen = 256;
    p.Md = (long)N * p.Ho * p.Wo;
    p.ntaps = 7;
    p.persample = 0;
    p.split_stride = 0;
    p.xpad = 1;
    p.sqnorm = nullptr;
    p.ws = nullptr; p.ws_bytes = 0;
    if (dtype == PRIMIA_F32) return launch_wgrad<float, 64, 32, true>(p, (hipStream_t)stream);
    if (dtype == PRIMIA_BF16) {
        const int rc = stem_wgrad_halo_dispatch((const bf16*)x_padded, (const bf16*)dy, dw_acc, N, H, W, (hipStream_t)stream,
                                                ws, ws_bytes);
        if (rc != PRIMIA_ERR_UNSUPPORTED) return rc;
        return launch_wgrad<bf16, 64, 32, true>(p, (hipStream_t)stream);
    }
    return PRIMIA_ERR_ARG;
}

extern "C" int primia_conv2d_wgrad(const primia_conv_desc* d, const void* x, const void* dy, float* dw_acc,
                                   int dtype, primia_stream_t stream) {
    return conv2d_wgrad_impl(d, x, dy, dw_acc, 0, dtype, stream);
}

static bool fill_wgrad_params(const primia_conv_desc* d, WgradParams& p, ConvGeom& g) {
    if (!d || !g.init(*d)) return false;
    p.N = g.N; p.H = g.H; p.W = g.W; p.C = g.C; p.K = g.K; p.R = g.R; p.S = g.S;
    p.stride = g.stride; p.pad = g.pad; p.Ho = g.Ho; p.Wo = g.Wo;
    p.klen = g.klen;
    p.Md = (long)g.N * g.Ho * g.Wo;
    p.ntaps = g.stem ? g.R : g.R * g.S;
    p.persample = 0;
    p.split_stride = 0;
    p.xpad = 0;
    p.sqnorm = nullptr;
    p.ws = nullptr; p.ws_bytes = 0;
    return true;
}

extern "C" int64_t primia_conv_wgrad_ws_bytes(const primia_conv_desc* d, int dtype) {
    WgradParams p;
    ConvGeom g;
    if (!fill_wgrad_params(d, p, g)) return PRIMIA_ERR_ARG;
    // mirrors the kernel choice of conv2d_wgrad_impl below
    if (dtype == PRIMIA_F32) {
        if (g.stem) return (int64_t)wgrad_ws_need<float, 64, 32, true>(p);
        if (g.K % 128 == 0 && g.C % 128 == 0) return (int64_t)wgrad_ws_need<float, 128, 128, false>(p);
        return (int64_t)wgrad_ws_need<float, 64, 64, false>(p);
    }
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_ARG;
    const char force = wgrad_force();
    const bool dma = force == 'd' || (force != 'o' && (g.C >= 256 || (g.K >= 256 && g.C >= 128)));
    if (!force && !g.stem) {
        const size_t n = wgrad_patch_ws_bytes(p);
        if (n > 0) return (int64_t)n;
        const size_t nt = wgrad_tap_ws_bytes(p);
        if (nt > 0) return (int64_t)nt;
    }
    if (!g.stem && dma) return (int64_t)wgrad_dma_ws_bytes(p);
    if (g.stem) return (int64_t)wgrad_ws_need<bf16, 64, 32, true>(p);
    if (g.K % 128 == 0 && g.C % 128 == 0) return (int64_t)wgrad_ws_need<bf16, 128, 128, false>(p);
    return (int64_t)wgrad_ws_need<bf16, 64, 64, false>(p);
}

extern "C" int primia_conv2d_wgrad_ws(const primia_conv_desc* d, const void* x, const void* dy, float* dw,
                                      void* ws, int64_t ws_bytes, int dtype, primia_stream_t stream) {
    return conv2d_wgrad_impl(d, x, dy, dw, 0, dtype, stream, nullptr, (float*)ws, ws_bytes > 0 ? (size_t)ws_bytes : 0);
}

extern "C" int64_t primia_conv_wgrad_pair_ws_bytes(const primia_conv_desc* d, const primia_conv_desc* d2, int dtype) {
    WgradParams p, p2;
    ConvGeom g, g2;
    if (!fill_wgrad_params(d, p, g) || !fill_wgrad_params(d2, p2, g2)) return PRIMIA_ERR_ARG;
    if (dtype != PRIMIA_BF16 || wgrad_force()) return 0;
    return (int64_t)wgrad_tap_pair_ws_bytes(p, p2);
}

extern "C" int primia_conv2d_wgrad_pair_ws(const primia_conv_desc* d, const void* x, const void* dy, float* dw,
                                           const primia_conv_desc* d2, const void* dy2, float* dw2, void* ws,
                                           int64_t ws_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && d2 && x && dy && dy2 && dw && dw2);
    WgradParams p, p2;
    ConvGeom g, g2;
    PRIMIA_REQUIRE(fill_wgrad_params(d, p, g) && fill_wgrad_params(d2, p2, g2));
    if (dtype != PRIMIA_BF16 || wgrad_force()) return PRIMIA_ERR_UNSUPPORTED;
    p.x = x; p.dy = dy; p.dw = dw; p.ws = (float*)ws; p.ws_bytes = ws && ws_bytes > 0 ? (size_t)ws_bytes : 0;
    p2.x = x; p2.dy = dy2; p2.dw = dw2;
    return wgrad_tap_pair_dispatch(p, p2, (hipStream_t)stream);
}

// Several layers of ONE shape in one launch (conv_wgrad_patch.hip: grouped launch).  n layers, 2 <= n <= 4; unused
// (x, dy, dw_acc) triples are null.
extern "C" int primia_conv_wgrad_group_size(const primia_conv_desc* d, int count, int dtype) {
    ConvGeom g;
    WgradParams p;
    if (!d || count < 1 || !fill_wgrad_params(d, p, g)) return PRIMIA_ERR_ARG;
    const char force = wgrad_force();
    if (dtype != PRIMIA_BF16 || g.stem || force) return 0;
    p.persample = 0;
    return wgrad_patch_group_size(p, count);
}

extern "C" int64_t primia_conv_wgrad_group_ws_bytes(const primia_conv_desc* d, int n, int dtype) {
    ConvGeom g;
    WgradParams p;
    if (!d || !fill_wgrad_params(d, p, g)) return PRIMIA_ERR_ARG;
    if (dtype != PRIMIA_BF16 || g.stem) return 0;
    p.persample = 0;
    return (int64_t)wgrad_patch_group_ws_bytes(p, n);
}

extern "C" int primia_conv2d_wgrad_group_ws(const primia_conv_desc* d, int n, const void* x0, const void* dy0, float* dw0,
                                            const void* x1, const void* dy1, float* dw1, const void* x2, const void* dy2,
                                            float* dw2, const void* x3, const void* dy3, float* dw3, void* ws,
                                            int64_t ws_bytes, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && n >= 2 && n <= 4 && ws && ws_bytes > 0);
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_UNSUPPORTED;
    const void* xs[4] = {x0, x1, x2, x3};
    const void* dys[4] = {dy0, dy1, dy2, dy3};
    float* dws[4] = {dw0, dw1, dw2, dw3};
    WgradParams ps[4];
    for (int i = 0; i < n; ++i) {
        ConvGeom g;
        PRIMIA_REQUIRE(xs[i] && dys[i] && dws[i] && fill_wgrad_params(d, ps[i], g));
        if (g.stem) return PRIMIA_ERR_UNSUPPORTED;
        ps[i].x = xs[i]; ps[i].dy = dys[i]; ps[i].dw = dws[i];
        ps[i].persample = 0; ps[i].sqnorm = nullptr;
        ps[i].ws = (float*)ws; ps[i].ws_bytes = (size_t)ws_bytes;
    }
    return wgrad_patch_group_dispatch(ps, n, (hipStream_t)stream);
}

// ---- DP-SGD: per-sample gradient tiles KEPT by the norm pass, clipped sum = a weighted reduce --------------------------
// Where a layer's per-sample gradients are small (the stem: 64 KiB per sample; the 64 -> 64 convs of layer1: 144 KiB),
// the norm pass stores each sample's complete tile next to adding its squares, and the clipped sum  sum_n clip_n g_n  is
// ONE ordered reduce over those tiles weighted by the clip factors — instead of scaling the rows of dy (a read + write
// of the layer's whole dy: 411 MB for the stem) and running the weight gradient a second time.
extern "C" int64_t primia_stem_conv_wgrad_persample_slab_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return PRIMIA_ERR_ARG;
    return stem_wgrad_halo_ws_bytes(N, H, W) > 0 ? (int64_t)N * 64 * 256 * (int64_t)sizeof(float) : 0;
}

extern "C" int primia_stem_conv_wgrad_persample_sqnorm_keep(const void* x_padded, const void* dy, double* sqnorm, void* slabs,
                                                            int64_t slab_bytes, int N, int H, int W, int dtype,
                                                            primia_stream_t stream) {
    PRIMIA_REQUIRE(x_padded && dy && sqnorm && slabs && N > 0 && H > 0 && W > 0);
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_UNSUPPORTED;
    if (slab_bytes < (int64_t)N * 64 * 256 * (int64_t)sizeof(float)) return PRIMIA_ERR_WORKSPACE;
    return stem_wgrad_halo_dispatch((const bf16*)x_padded, (const bf16*)dy, nullptr, N, H, W, (hipStream_t)stream,
                                    (float*)slabs, (size_t)slab_bytes, sqnorm);
}

extern "C" int primia_stem_conv_wgrad_clipped_sum(const void* slabs, const float* clip, float* dw_acc, int N,
                                                  primia_stream_t stream) {
    PRIMIA_REQUIRE(slabs && clip && dw_acc && N > 0);
    wgrad_tile_reduce((const float*)slabs, dw_acc, N, 1, 64, 256, 1, 1, 4, 256, 1, (hipStream_t)stream, clip);
    return launch_status();
}

extern "C" int64_t primia_conv_wgrad_persample_slab_bytes(const primia_conv_desc* d, int dtype) {
    ConvGeom g;
    WgradParams p;
    if (!d || !fill_wgrad_params(d, p, g)) return PRIMIA_ERR_ARG;
    const char force = wgrad_force();
    if (dtype != PRIMIA_BF16 || g.stem || force) return 0;
    p.persample = 1;
    return (int64_t)wgrad_patch_keep_bytes(p);
}

extern "C" int primia_conv2d_wgrad_persample_sqnorm_keep(const primia_conv_desc* d, const void* x, const void* dy,
                                                         double* sqnorm, void* slabs, int64_t slab_bytes, int dtype,
                                                         primia_stream_t stream) {
    PRIMIA_REQUIRE(d && x && dy && sqnorm && slabs);
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_UNSUPPORTED;
    ConvGeom g;
    WgradParams p;
    PRIMIA_REQUIRE(fill_wgrad_params(d, p, g));
    if (g.stem) return PRIMIA_ERR_UNSUPPORTED;
    p.x = x; p.dy = dy; p.dw = nullptr;
    p.persample = 1; p.sqnorm = sqnorm;
    p.ws = (float*)slabs; p.ws_bytes = (size_t)slab_bytes;
    return wgrad_patch_keep_dispatch(p, (hipStream_t)stream);
}

extern "C" int primia_conv_wgrad_clipped_sum(const primia_conv_desc* d, const void* slabs, const float* clip, float* dw_acc,
                                             int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && slabs && clip && dw_acc);
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_UNSUPPORTED;
    ConvGeom g;
    WgradParams p;
    PRIMIA_REQUIRE(fill_wgrad_params(d, p, g));
    p.persample = 1; p.dw = dw_acc;
    return wgrad_patch_clipped_sum(p, (const float*)slabs, clip, (hipStream_t)stream);
}

extern "C" int primia_conv2d_wgrad_persample(const primia_conv_desc* d, const void* x, const void* dy,
                                             float* dw_ps, int dtype, primia_stream_t stream) {
    return conv2d_wgrad_impl(d, x, dy, dw_ps, 1, dtype, stream);
}

namespace primia {
int dp_ghost_sqnorm_dispatch(const void* x, const void* dy, double* sq, int N, int H, int W, int C, int K, int R, int S,
                             int stride, int pad, hipStream_t st);
}

extern "C" int primia_conv2d_wgrad_persample_sqnorm(const primia_conv_desc* d, const void* x, const void* dy,
                                                    double* sqnorm, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(sqnorm);
    if (d && x && dy && dtype == PRIMIA_BF16) {
        // 7x7 images, 3x3 / stride 1: the norms come out of two Gram matrices per sample (dp_ghost.hip), the per-sample
        // gradients are never formed
        const int rc = dp_ghost_sqnorm_dispatch(x, dy, sqnorm, d->N, d->H, d->W, d->C, d->K, d->R, d->S, d->stride, d->pad,
                                                (hipStream_t)stream);
        if (rc != PRIMIA_ERR_UNSUPPORTED) return rc;
    }
    return conv2d_wgrad_impl(d, x, dy, nullptr, 1, dtype, stream, sqnorm);
}

extern "C" int primia_conv_wgrad_persample_kernel_id(const primia_conv_desc* d, int dtype) {
    ConvGeom g;
    WgradParams p;
    if (!d || !fill_wgrad_params(d, p, g)) return PRIMIA_ERR_ARG;
    p.persample = 1;
    if (g.stem) return 15;
    if (dtype != PRIMIA_BF16) return 14;
    const int gh = dp_ghost_kernel_id(d->H, d->W, d->C, d->K, d->R, d->S, d->stride, d->pad);
    if (gh) return gh;
    const char force = wgrad_force();
    if (!force) {
        const int id = wgrad_patch_persample_kernel_id(p);
        if (id) return id;
        const int idt = wgrad_tap_persample_kernel_id(p);
        if (idt) return idt;
    }
    const bool dma = force == 'd' || (force != 'o' && (g.C >= 256 || (g.K >= 256 && g.C >= 128)));
    return dma ? 13 : 14;
}

extern "C" int primia_conv_wgrad_kernel_id(const primia_conv_desc* d, int dtype) {
    ConvGeom g;
    WgradParams p;
    if (!d || !fill_wgrad_params(d, p, g)) return PRIMIA_ERR_ARG;
    if (g.stem) return 15;
    if (dtype != PRIMIA_BF16) return 14;
    const char force = wgrad_force();
    if (!force) {
        const int id = wgrad_patch_kernel_id(p);
        if (id) return id;
        const int idt = wgrad_tap_kernel_id(p);
        if (idt) return idt;
    }
    const bool dma = force == 'd' || (force != 'o' && (g.C >= 256 || (g.K >= 256 && g.C >= 128)));
    return dma ? 13 : 14;
}

static int conv2d_wgrad_impl(const primia_conv_desc* d, const void* x, const void* dy, float* dw_acc, int persample,
                             int dtype, primia_stream_t stream, double* sqnorm, float* ws, size_t ws_bytes) {
    PRIMIA_REQUIRE(d && x && dy && (dw_acc || sqnorm));
    ConvGeom g;
    WgradParams p;
    PRIMIA_REQUIRE(fill_wgrad_params(d, p, g));
    p.x = x; p.dy = dy; p.dw = dw_acc;
    p.persample = persample;
    p.sqnorm = sqnorm;
    p.ws = ws; p.ws_bytes = ws ? ws_bytes : 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32) {
        if (g.stem) return launch_wgrad<float, 64, 32, true>(p, st);
        if (g.K % 128 == 0 && g.C % 128 == 0) return launch_wgrad<float, 128, 128, false>(p, st);
        return launch_wgrad<float, 64, 64, false>(p, st);
    } else if (dtype == PRIMIA_BF16) {
        // Measured per layer (profiles/r01_conv_layers_*): the halo-patch kernel (conv_wgrad_patch.hip) wins
        // on every 3x3 / stride-1 layer; of the per-tap kernels the LDS-DMA one wins on the wide, few-pixel
        // layers (layer3/4) and the register-staged one elsewhere.  option wgrad_kernel = 1 (old) | 2 (dma) | 3 (tap) forces one
        // (tap = per-tap kernels with the default old/dma choice).
        const char force = wgrad_force();
        const bool dma = force == 'd' || (force != 'o' && (g.C >= 256 || (g.K >= 256 && g.C >= 128)));
        if (!force && !g.stem) {
            const int rc = wgrad_patch_dispatch(p, st);
            if (rc != PRIMIA_ERR_UNSUPPORTED) return rc;
            const int rt = wgrad_tap_dispatch(p, st);     // stride-2 / 1x1 layers with a workspace
            if (rt != PRIMIA_ERR_UNSUPPORTED) return rt;
            const int rp = wgrad_tap_persample_dispatch(p, st);   // ... and their DP-SGD norm pass
            if (rp != PRIMIA_ERR_UNSUPPORTED) return rp;
        }
        if (!g.stem && dma) return wgrad_dma_dispatch(p, st);
        if (g.stem) return launch_wgrad<bf16, 64, 32, true>(p, st);
        if (g.K % 128 == 0 && g.C % 128 == 0) return launch_wgrad<bf16, 128, 128, false>(p, st);
        return launch_wgrad<bf16, 64, 64, false>(p, st);
    }
    return PRIMIA_ERR_ARG;
}

// =================================================================================================
// LDS-DMA weight gradient (bf16, every non-stem conv).  Same GEMM as above, but
//   * both tiles are staged with global_load_lds_dwordx4 into linear rows whose 32-byte pair-chunks
//     are XOR-permuted on the SOURCE side, so the ds_read_b64_tr_b16 transposing reads (4 pixel
//     rows x 32 B per 16-lane group, two groups per LDS cycle) fall on distinct banks;
//   * 8 waves (2 x 4) per block for latency hiding; per-step address math is one reciprocal
//     division per staged row instead of carried (n, ho, wo) counters.
// (A variant that shares one activation tile between the three taps of a kernel row was measured
//  slower — profiles/r01_wgrad_rowtap_negative_result.txt — and is not kept.)
// =================================================================================================
namespace primia {

__device__ __attribute__((aligned(16))) const unsigned char kWgZeroPage[16] = {0};

template <int ROWB>
__device__ __forceinline__ int wg_swz(int row) {
    // pair-chunk (32 B) permutation key: 8 slots per 256-B row, 4 per 128-B row
    return ROWB == 256 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}

// q = m / d, rem = m % d for 0 <= m < 2^24 (all ResNet-18 pixel counts at batch <= 1024)
__device__ __forceinline__ int fast_div(int m, int d, float rcp, int& rem) {
    int q = (int)((float)m * rcp);
    rem = m - q * d;
    if (rem >= d) {
        ++q;
        rem -= d;
    } else if (rem < 0) {
        --q;
        rem += d;
    }
    return q;
}

template <int BMK, int BNC>
__global__ __launch_bounds__(512) void conv_wgrad_dma_kernel(WgradParams p) {
    constexpr int KP = 64;  // pixels per k-step
    constexpr int ROW_A = BMK * 2, ROW_B = BNC * 2;
    constexpr int RPI_A = 1024 / ROW_A, RPI_B = 1024 / ROW_B;  // rows per DMA instruction
    constexpr int NA = KP / RPI_A, NX = KP / RPI_B;
    constexpr int STAGE = KP * (ROW_A + ROW_B);
    constexpr int FM = BMK / 2 / 16, FN = BNC / 4 / 16;  // waves: 2 (k) x 4 (c)
    static_assert(FM >= 1 && FN >= 1, "tile too small");
    typedef __attribute__((address_space(3))) bf16x4_t* lds4_t;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % p.nct; bid /= p.nct;
    const int kt = bid % p.nkt; bid /= p.nkt;
    const int tap = bid % p.ntaps;
    const int split = bid / p.ntaps;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int ms = (int)((long)split * p.pix_per_split);
    int me = ms + (int)p.pix_per_split;
    if (me > (int)p.Md) me = (int)p.Md;
    const int nsteps = (me - ms + KP - 1) / KP;
    const float rcp_wo = 1.0f / (float)p.Wo, rcp_ho = 1.0f / (float)p.Ho;

    const bf16* __restrict__ x = (const bf16*)p.x;
    const bf16* __restrict__ dy = (const bf16*)p.dy;

    auto stage = [&](int step, int buf) {
        char* base = smem + buf * STAGE;
        const int p0 = ms + step * KP;
        for (int idx = wid; idx < NA + NX; idx += 8) {
            if (idx < NA) {
                const int row = idx * RPI_A + lane / (ROW_A / 16);
                const int sl = lane % (ROW_A / 16);
                const int chunk = ((((sl >> 1) ^ wg_swz<ROW_A>(row)) << 1) | (sl & 1));
                const int m = p0 + row;
                const bf16* g = m < me ? dy + (long)m * p.K + kt * BMK + chunk * 8 : (const bf16*)kWgZeroPage;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(base + idx * 1024), 16, 0, 0);
            } else {
                const int j = idx - NA;
                const int row = j * RPI_B + lane / (ROW_B / 16);
                const int sl = lane % (ROW_B / 16);
                const int chunk = ((((sl >> 1) ^ wg_swz<ROW_B>(row)) << 1) | (sl & 1));
                const int m = p0 + row;
                int wo, ho;
                const int t = fast_div(m, p.Wo, rcp_wo, wo);
                const int n = fast_div(t, p.Ho, rcp_ho, ho);
                const int hs = ho * p.stride - p.pad + tr, ws = wo * p.stride - p.pad + ts;
                const bool ok = m < me && hs >= 0 && hs < p.H && ws >= 0 && ws < p.W;
                const bf16* g = ok ? x + ((long)(n * p.H + hs) * p.W + ws) * p.C + ct * BNC + chunk * 8
                                   : (const bf16*)kWgZeroPage;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(base + KP * ROW_A + j * 1024),
                                                 16, 0, 0);
            }
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int tp = fr >> 2, tc8 = (fr & 3) * 8;  // transpose-read: row offset / byte offset inside the 32-B block
    const int ka0 = wm * (BMK / 2), cb0 = wn * (BNC / 4);

    auto compute = [&](int buf) {
        const char* la = smem + buf * STAGE;
        const char* lb = la + KP * ROW_A;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t a[FM], b[FN];
            const int r0 = kk * 32 + fg * 8 + tp, r1 = r0 + 4;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int pc = (ka0 + 16 * i) >> 4;
                bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds4_t)(la + r0 * ROW_A + ((pc ^ wg_swz<ROW_A>(r0)) << 5) + tc8));
                bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds4_t)(la + r1 * ROW_A + ((pc ^ wg_swz<ROW_A>(r1)) << 5) + tc8));
                a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int pc = (cb0 + 16 * j) >> 4;
                bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds4_t)(lb + r0 * ROW_B + ((pc ^ wg_swz<ROW_B>(r0)) << 5) + tc8));
                bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (lds4_t)(lb + r1 * ROW_B + ((pc ^ wg_swz<ROW_B>(r1)) << 5) + tc8));
                b[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    if (nsteps > 0) {
        stage(0, 0);
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            if (step + 1 < nsteps) stage(step + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    }

    if (p.sqnorm) {
        double sq = 0.0;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) sq += (double)acc[i][j][t] * (double)acc[i][j][t];
        wave_sqnorm_add(sq, p.sqnorm + split);
        return;
    }
    if (p.ws) {   // atomic-free path, as conv_wgrad_kernel
        float* o = p.ws + ((long)(((tap * p.nkt + kt) * p.nct + ct)) * p.nsplit + split) * (BMK * BNC);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    o[(ka0 + 16 * i + fg * 4 + t) * BNC + cb0 + 16 * j + fr] = acc[i][j][t];
        return;
    }
    const int ebase = tap * p.C + ct * BNC;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = kt * BMK + ka0 + 16 * i + fg * 4 + t;
                const int e = ebase + cb0 + 16 * j + fr;
                unsafeAtomicAdd(p.dw + (long)split * p.split_stride + (long)k * p.klen + e, acc[i][j][t]);
            }
}

template <int BMK, int BNC>
static bool wgrad_dma_geometry(WgradParams& p) {
    constexpr int KP = 64;
    if (p.Md >= (1L << 24)) return false;  // fast_div range
    p.nkt = p.K / BMK;
    p.nct = p.C / BNC;
    const int combos = p.ntaps * p.nkt * p.nct;
    // Blocks of one launch: ONE round at two resident blocks per CU.  Every block ends with a BMK x BNC fp32 flush, so
    // fewer, longer blocks halve that traffic too (1024 -> 504: layer3.0.conv1 100 -> 84 us, layer4.0.conv1 97 -> 83).
    const int target_blocks = PRIMIA_OPT(wg_blocks) > 0 ? PRIMIA_OPT(wg_blocks) : 504;
    long want = (target_blocks + combos - 1) / combos;
    long max_split = (p.Md + 8 * KP - 1) / (8 * KP);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    long pps = (p.Md + want - 1) / want;
    pps = (pps + KP - 1) / KP * KP;
    if (p.persample) pps = (long)p.Ho * p.Wo;
    p.pix_per_split = pps;
    p.nsplit = (int)((p.Md + pps - 1) / pps);
    p.split_stride = p.persample ? (long)p.K * p.klen : 0;
    return true;
}

template <int BMK, int BNC>
static int launch_wgrad_dma(WgradParams p, hipStream_t st) {
    constexpr int KP = 64;
    if (!wgrad_dma_geometry<BMK, BNC>(p)) return PRIMIA_ERR_ARG;
    const int combos = p.ntaps * p.nkt * p.nct;
    const bool store = !p.persample && p.ws && p.ws_bytes >= (size_t)combos * p.nsplit * BMK * BNC * sizeof(float);
    if (!store) p.ws = nullptr;
    const int grid = combos * p.nsplit;
    const size_t lds = 2 * (size_t)KP * (BMK + BNC) * 2;
    auto kern = conv_wgrad_dma_kernel<BMK, BNC>;
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    kern<<<grid, 512, lds, st>>>(p);
    if (store) launch_tile_reduce(p, BMK, BNC, 0, st);
    return launch_status();
}

size_t wgrad_dma_ws_bytes(const WgradParams& w) {
    WgradParams p = w;
    if (p.K % 128 == 0 && p.C % 128 == 0) {
        if (!wgrad_dma_geometry<128, 128>(p)) return 0;
        return (size_t)p.ntaps * p.nkt * p.nct * p.nsplit * 128 * 128 * sizeof(float);
    }
    if (!wgrad_dma_geometry<64, 64>(p)) return 0;
    return (size_t)p.ntaps * p.nkt * p.nct * p.nsplit * 64 * 64 * sizeof(float);
}

int wgrad_dma_dispatch(const WgradParams& p, hipStream_t st) {
    if (p.K % 128 == 0 && p.C % 128 == 0) return launch_wgrad_dma<128, 128>(p, st);
    return launch_wgrad_dma<64, 64>(p, st);
}

}  // namespace primia
