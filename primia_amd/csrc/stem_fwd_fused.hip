// The stem's forward tail without its 411 MB activation: conv1 -> bn1 -> relu -> maxpool (torchlib/models.py:466-471) as
// two passes over the padded INPUT (108 MB at batch 256) instead of one over it and two over conv1's output:
//
//   pass 1  primia_stem_conv_stats     conv1 tiles -> bn1's per-block partial sums; nothing is stored
//                                      (stem_conv_fwd_kernel of stem_conv.hip with y == NULL)
//   pass 2  primia_stem_conv_pool      conv1 recomputed (60 GFLOP = 24 us of MFMA: cheaper than any one pass over its output)
//                                      -> scale / shift -> ReLU -> 3x3 / 2 max with first-maximum argmax codes
//                                      -> the pooled tensor (103 MB) + codes (51 MB); conv1's output only if asked for
//
// Pass 2: ONE 8-wave block per image walks the image's 8 x 16 output patches row band by row band (band = 8 conv rows,
// left to right).  The filter lives in registers and the 21 x 40 input patch is staged by LDS-DMA as in stem_conv.hip; the
// patch's activations z = relu(bn(y)) (rounded to bf16, as the unfused chain stores them) go to an LDS tile, and the 4 x 8
// pooled windows of the patch are formed from that tile plus a one-pixel halo that never leaves LDS: the row above comes
// from a carry buffer holding the previous band's last row (double-buffered by band parity), the column to the left from
// the previous patch's last column (both triple-buffered: the matrix phase of patch k + 1 runs beside the pooling of patch k).
// A ninth wave does nothing but stage input patches: the eight working waves store results every stage, and a wave that
// waits for its own LDS-DMA (vmcnt) would wait for its stores' acknowledgements as well.
// Windows use the packed (value, first position) keys of bn.hip's bn_relu_pool_fwd_key_kernel: pooled values and argmax
// codes are bit-identical to primia_bn_relu_maxpool_fwd_from_sums on a materialised y.
#include <stdlib.h>

#include "conv_wgrad.h"

namespace primia {

__device__ __attribute__((aligned(16))) const unsigned char kSffZeroPage[16] = {0};

__device__ __forceinline__ void sff_dma16(const void* g, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}

struct StemPoolParams {
    const bf16* xp;        // [N][Hp][Wp][4]
    const bf16* wt;        // [64][256] stem forward layout
    bf16* y;               // [N][Ho][Wo][64] or null
    bf16* pooled;          // [N][Hq][Wq][64]
    uint8_t* argmax;       // [N][Hq][Wq][64]
    const float *gamma, *beta, *mean, *invstd;
    int N, Hp, Wp, Ho, Wo, Hq, Wq;
    int PH, PW;            // 8 x 16 patches per image column / row
    int eval_mode;         // invstd[] holds the running VARIANCE (nn.BatchNorm2d in eval mode): 1 / sqrt(var + eps) here
    float eps;
};

// LDS: [z tiles 2 x 16 KiB][carry rows 3 x (1 + 8 * 16) slots][carry columns 3 x 8 slots][y tiles 2 x 16 KiB, only when
// conv1's output is stored][input ring: R patches of 7 KiB].  The ring is what hides HBM: a patch is 6.7 KB, and with two
// of them in flight per CU both passes ran at the memory LATENCY (214 us / 117 us at batch 256: 1.2 TB/s); R = 10 keeps
// 63 KB in flight (R = 6 when the y tiles take their 32 KiB).
constexpr int kSffPatch = 7 * 1024;                  // 21 rows x 320 B staged by 7 DMA instructions
constexpr int kSffTile = 128 * 128;                  // 8 x 16 pixels x 64 channels of bf16
constexpr int kSffOffZ = 0;                          // two z tiles
constexpr int kSffRowSlots = 1 + 8 * 16;             // slot 0 = column -1 (images up to 128 output columns: 256 x 256 inputs)
constexpr int kSffOffRow = kSffOffZ + 2 * kSffTile;  // carry rows, by band % 3
constexpr int kSffOffCol = kSffOffRow + 3 * kSffRowSlots * 128;   // carry columns, by stage % 3
constexpr int kSffOffY = kSffOffCol + 3 * 8 * 128;   // two y tiles (write-back of conv1's output, optional)
constexpr int kSffRing = 10, kSffRingY = 6;          // input ring depth without / with the y tiles
constexpr int kSffLdsNoY = kSffOffY + kSffRing * kSffPatch;                       // 157,056 B
constexpr int kSffLdsY = kSffOffY + 2 * kSffTile + kSffRingY * kSffPatch;         // 161,152 B
constexpr int kSffLds = kSffLdsY > kSffLdsNoY ? kSffLdsY : kSffLdsNoY;
static_assert(kSffLds <= 163840, "LDS budget");

__device__ __forceinline__ void sff_wait_patches(int k) {      // all but the k youngest patches (7 pieces each) have landed
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    }
}

// pixel slot of the tile: slots s and s + 2 (neighbouring windows' columns) on different halves of the 64 banks
__device__ __forceinline__ int sff_pslot(int slot) { return slot ^ ((slot >> 1) & 1); }

__global__ __launch_bounds__(576) void stem_conv_pool_kernel(StemPoolParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = (wave >> 2) & 1, pq = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int n = blockIdx.x;
    const int nstages = p.PH * p.PW;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int R = p.y ? kSffRingY : kSffRing;                          // ring depth; a patch is requested R - 1 stages ahead
    const int xoff = kSffOffY + (p.y ? 2 * kSffTile : 0);              // the ring starts behind the y tiles, if any

    if (wave == 8) {
        // =============================== loader wave: 7 DMA pieces per input patch ===============================
        auto stage = [&](int s, int buf) {
            const int ph = s / p.PW, pw = s - ph * p.PW;
            const long base = ((long)(n * p.Hp + ph * 16) * p.Wp + pw * 32) * 4;
#pragma unroll
            for (int pc = 0; pc < 7; ++pc) {
                const int G = pc * 64 + lane;
                const int row = G / 20, c16 = G - row * 20;
                const bf16* g = G < 420 ? p.xp + base + ((long)row * p.Wp + 2 * c16) * 4 : (const bf16*)kSffZeroPage;
                sff_dma16(g, __builtin_amdgcn_readfirstlane(lds0 + xoff + buf * kSffPatch + pc * 1024));
            }
        };
        const int D = (R - 1 < 9 ? R - 1 : 9);                        // (vmcnt counts to 63: at most 9 patches in flight)
        for (int s = 0; s < D && s < nstages; ++s) stage(s, s % R);
        for (int s = 0; s <= nstages; ++s) {
            // requested so far: patches 0 .. min(s - 1 + D, last); patch s must have landed
            int inflight = (s - 1 + D < nstages - 1 ? s - 1 + D : nstages - 1) - s;
            sff_wait_patches(inflight < 0 ? 0 : inflight);
            __builtin_amdgcn_s_barrier();
            if (s + D < nstages) stage(s + D, (s + D) % R);           // its slot was last read by the matrix phase of stage s - 1
        }
        return;
    }

    // ---- weights -> registers: A fragment (r, i): row 32*kh + 16*i + fr, elements r*32 + 8*fg .. +7 ----
    bf16x8_t wreg[7][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bf16* wrow = p.wt + (long)(32 * kh + 16 * i + fr) * 256 + 8 * fg;
#pragma unroll
        for (int r = 0; r < 7; ++r) wreg[r][i] = *(const bf16x8_t*)(wrow + r * 32);
    }
    // bn1's per-channel constants of this lane's 8 output channels 32 kh + 16 i + 4 fg + e
    float kmu[2][4], ksc[2][4], kbe[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 32 * kh + 16 * i + 4 * fg + e;
            kmu[i][e] = p.mean[c];
            const float is = p.eval_mode ? 1.f / sqrtf(p.invstd[c] + p.eps) : p.invstd[c];   // bn_apply_kernel's expression
            ksc[i][e] = is * p.gamma[c];
            kbe[i][e] = p.beta[c];
        }
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): (the builtin — the constants are known to have arrived before the loop)

    // B fragment of output row (2*pq + j), kernel row r: byte offset ((2*(2*pq + j) + r) * 20 + fr + fg) * 16
    const int offb0 = ((4 * pq) * 20 + fr + fg) * 16;  // + (2*j + r) * 320

    // where this lane's results go: pixel (row 2 pq + j, column fr); 8 bytes = channels 32 kh + 16 i + 4 fg .. +3 =
    // 16-byte chunk 4 kh + 2 i + (fg >> 1), half fg & 1; chunks XOR-swizzled by the pixel's column pair
    const int ckey = (fr >> 1) & 7;
    auto chunk_off = [&](int i) { return (((4 * kh + 2 * i + (fg >> 1)) ^ ckey) << 4) | ((fg & 1) << 3); };

    auto compute = [&](int s, int xbuf, int tb) {
        const int ph = s / p.PW, pw = s - ph * p.PW;
        const char* sb = smem + xoff + xbuf * kSffPatch;
        f32x4 acc[2][2];  // [row j][K fragment i]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            bf16x8_t b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8_t*)(sb + offb0 + (2 * j + r) * 320);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[r][i], b[j], acc[j][i], 0, 0, 0);
        }
        char* zt = smem + kSffOffZ + tb * kSffTile;
        char* yt = smem + kSffOffY + tb * kSffTile;
        char* crow = smem + kSffOffRow + ((ph + 1) % 3) * (kSffRowSlots * 128);      // read by the band below
        char* ccol = smem + kSffOffCol + ((s + 1) % 3) * (8 * 128);                  // read by the next stage (patch to the right)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 2 * pq + j;
            const int slot = sff_pslot(row * 16 + fr);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // y as stored (bf16, round to nearest even: v_cvt_pk_bf16_f32 takes two values), then the activation as the
                // unfused chain stores it: relu(bn(y as stored)), rounded to bf16, sign cleared
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                u32x2 yo, zo;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    yo[h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{acc[j][i][2 * h], acc[j][i][2 * h + 1]}, bf2));
                    const float y0 = __uint_as_float(yo[h] << 16), y1 = __uint_as_float(yo[h] & 0xffff0000u);
                    const float z0 = fmaxf(__builtin_fmaf(y0 - kmu[i][2 * h], ksc[i][2 * h], kbe[i][2 * h]), 0.f);
                    const float z1 = fmaxf(__builtin_fmaf(y1 - kmu[i][2 * h + 1], ksc[i][2 * h + 1], kbe[i][2 * h + 1]), 0.f);
                    zo[h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{z0, z1}, bf2)) & 0x7fff7fffu;
                }
                const int co = chunk_off(i);
                *(u32x2*)(zt + slot * 128 + co) = zo;
                if (p.y) *(u32x2*)(yt + slot * 128 + co) = yo;
                if (row == 7) *(u32x2*)(crow + (1 + 16 * pw + fr) * 128 + co) = zo;
                if (fr == 15) *(u32x2*)(ccol + row * 128 + co) = zo;
            }
        }
    };

    // ---- pooling of patch s from tile tb: thread -> (window (k, m) of 4 x 8, 4 channels q) ----
    // Branch-free: the nine tap addresses are a per-thread constant offset from one of three per-stage bases (the z
    // tile; the carry row for the taps above a window of the patch's first row; the carry column for the taps left of a
    // window of its first column), and a tap outside the IMAGE is masked out of the key comparison.
    const int pwin = tid >> 4, pq4 = tid & 15;
    const int wk = pwin >> 3, wm = pwin & 7;
    const bool top = wk == 0, left = wm == 0;      // (top: wave-uniform)
    int toff[9];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
            const int r = 2 * wk - 1 + dr, c = 2 * wm - 1 + dc;
            // 16-byte chunk pq4 >> 1 of the pixel, swizzled by its column pair (column -1 is the previous patch's 15: key 7)
            const int key7 = c >= 0 ? (c >> 1) & 7 : 7;
            const int co = (((pq4 >> 1) ^ key7) << 4) | ((pq4 & 1) << 3);
            toff[dr * 3 + dc] = (r < 0 ? (1 + c) * 128 : (c < 0 ? r * 128 : sff_pslot(r * 16 + c) * 128)) + co;
        }
    auto pool = [&](int s, int tb) {
        const int ph = s / p.PW, pw = s - ph * p.PW;
        const unsigned bz = lds0 + kSffOffZ + tb * kSffTile;
        const unsigned brow = lds0 + kSffOffRow + (ph % 3) * (kSffRowSlots * 128) + 16 * pw * 128;
        const unsigned bcol = lds0 + kSffOffCol + (s % 3) * (8 * 128);
        const unsigned mrow = (top && ph == 0) ? 0u : 0xffffffffu;       // taps above the image
        const unsigned mcol = (left && pw == 0) ? 0u : 0xffffffffu;      // taps left of the image
        unsigned key[4] = {0u, 0u, 0u, 0u};       // below every real key (15 - tap >= 7)
        typedef __attribute__((address_space(3))) const u32x2* lds2_t;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dr = t / 3, dc = t - 3 * dr;
            unsigned base = bz;
            if (dc == 0) base = left ? bcol : base;
            if (dr == 0) base = top ? brow : base;
            const u32x2 v = *(lds2_t)(size_t)(base + (unsigned)toff[t]);
            const unsigned tail = 15u - (unsigned)t;
            unsigned m = 0xffffffffu;
            if (dr == 0) m &= mrow;
            if (dc == 0) m &= mcol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned zb = (e & 1) ? v[e >> 1] >> 16 : v[e >> 1] & 0xffffu;
                unsigned cand = (zb << 4) | tail;
                if (dr == 0 || dc == 0) cand &= m;
                key[e] = cand > key[e] ? cand : key[e];
            }
        }
        const long o = (((long)n * p.Hq + ph * 4 + wk) * p.Wq + pw * 8 + wm) * 64 + pq4 * 4;
        u32x2 pv;
        pv[0] = (key[0] >> 4) | ((key[1] >> 4) << 16);
        pv[1] = (key[2] >> 4) | ((key[3] >> 4) << 16);
        const uint32_t pk = (15u - (key[0] & 15u)) | ((15u - (key[1] & 15u)) << 8) | ((15u - (key[2] & 15u)) << 16) |
                            ((15u - (key[3] & 15u)) << 24);
        *(u32x2*)(p.pooled + o) = pv;
        *(uint32_t*)(p.argmax + o) = pk;
        if (p.y) {
            // conv1's output rows of the patch: wave w stores output row w (16 pixels x 128 B), two 1-KiB instructions
            const char* yt = smem + kSffOffY + tb * kSffTile;
            bf16* rowp = p.y + ((long)(n * p.Ho + ph * 8 + wave) * p.Wo + pw * 16) * 64;
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int px = hlf * 8 + (lane >> 3), c16 = lane & 7;
                const u32x4 v = *(const u32x4*)(yt + sff_pslot(wave * 16 + px) * 128 + ((c16 ^ ((px >> 1) & 7)) << 4));
                *(u32x4*)(rowp + px * 64 + c16 * 8) = v;
            }
        }
    };

    // 3-deep input ring, ONE barrier per stage: after the barrier of stage s the working waves pool patch s - 1 (tile
    // (s - 1) & 1) and multiply patch s into tile s & 1.  A wave's ds_writes of stage s have landed before it passes the
    // barrier of stage s + 1 (lgkmcnt(0)); the loader's DMA pieces for patch s + 1 likewise (its vmcnt).
    for (int s = 0; s <= nstages; ++s) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s > 0) pool(s - 1, (s - 1) & 1);
        if (s < nstages) compute(s, s % R, s & 1);
    }
}

}  // namespace primia

using namespace primia;

extern "C" {

// 1: primia_stem_conv_pool serves this shape
int primia_stem_conv_pool_ok(int N, int H, int W, int dtype) {
    if (dtype != PRIMIA_BF16 || N <= 0 || H <= 0 || W <= 0 || H % 32 != 0 || W % 32 != 0 || W / 2 > 128) return 0;
    if ((long)N * (H + 6) * (W + 8) * 4 >= (1L << 31) || (long)N * (H / 2) * (W / 2) * 64 >= (1L << 31)) return 0;
    return 1;
}

static int stem_conv_pool_launch(const void* x_padded, const void* w_fwd, void* y, void* pooled, uint8_t* argmax,
                                 const float* gamma, const float* beta, const float* mean, const float* invstd_or_var,
                                 int eval_mode, float eps, int N, int H, int W, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(x_padded && w_fwd && pooled && argmax && gamma && beta && mean && invstd_or_var);
    if (!primia_stem_conv_pool_ok(N, H, W, dtype)) return PRIMIA_ERR_UNSUPPORTED;
    StemPoolParams p;
    p.eval_mode = eval_mode; p.eps = eps;
    p.xp = (const bf16*)x_padded; p.wt = (const bf16*)w_fwd; p.y = (bf16*)y; p.pooled = (bf16*)pooled; p.argmax = argmax;
    p.gamma = gamma; p.beta = beta; p.mean = mean; p.invstd = invstd_or_var;
    p.N = N; p.Hp = H + 6; p.Wp = W + 8; p.Ho = H / 2; p.Wo = W / 2;
    p.Hq = p.Ho / 2; p.Wq = p.Wo / 2;
    p.PH = p.Ho / 8; p.PW = p.Wo / 16;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)stem_conv_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSffLds) !=
            hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    stem_conv_pool_kernel<<<N, 576, kSffLds, (hipStream_t)stream>>>(p);
    return launch_status();
}

int primia_stem_conv_pool(const void* x_padded, const void* w_fwd, void* y, void* pooled, uint8_t* argmax,
                          const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, int N,
                          int H, int W, int dtype, primia_stream_t stream) {
    return stem_conv_pool_launch(x_padded, w_fwd, y, pooled, argmax, gamma, beta, save_mean, save_invstd, 0, 0.f, N, H, W,
                                 dtype, stream);
}

// Eval mode (running statistics): conv1 -> bn1 -> relu -> maxpool in ONE pass over the input, nothing else read or written
// — the chain primia_stem_conv_fwd -> primia_bn_fwd_eval -> primia_maxpool3x3s2_fwd, bit for bit, without its two tensors.
int primia_stem_conv_pool_eval(const void* x_padded, const void* w_fwd, void* pooled, uint8_t* argmax, const float* gamma,
                               const float* beta, const float* running_mean, const float* running_var, float eps, int N,
                               int H, int W, int dtype, primia_stream_t stream) {
    return stem_conv_pool_launch(x_padded, w_fwd, nullptr, pooled, argmax, gamma, beta, running_mean, running_var, 1, eps,
                                 N, H, W, dtype, stream);
}

}  // extern "C"
