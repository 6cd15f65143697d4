// DP-SGD: per-sample squared gradient norms of a 3x3 / stride-1 / pad-1 convolution on 7x7 images WITHOUT forming the
// per-sample gradients ("ghost" norms), bf16, gfx950.
//
// The per-sample weight gradient of sample n is  g = sum_p dy_p (x) u_p  (dy_p: the K output-gradient channels of pixel p,
// u_p: its 9 C unfolded input values), so
//     ||g||^2 = sum_{p,q} (dy_p . dy_q) (u_p . u_q)
// and with XX = the Gram matrix of the 81 pixels of the zero-padded 9x9 input halo,
//     u_p . u_q = sum over the 9 taps t of XX[h(p) + t][h(q) + t].
// For layer4 of ResNet-18 (49 pixels, C = K = 512) that is 49^2 (512 + 512 (81/49)^2 ...) ~ 14 MFLOP per sample on
// the matrix cores instead of the 231 MFLOP of the per-sample gradient itself (and none of its 8x8-sub-patch padding):
// the norm pass of the three such layers took 127 us each (profiles/r03_dp_step_timeline.txt), the same FLOPs as their
// batch pass at half the speed.
//
// One 4-wave block per sample, everything in LDS: x halo rows and dy rows as [row][C or K] bf16 with a 16-byte pad per
// row (row pitch = 65 sixteen-byte chunks: the 16 rows of a fragment land in 16 different bank groups), both Gram
// matrices by v_mfma_f32_16x16x32_bf16 (A and B fragments are the same 16-byte row reads: a Gram matrix is X X^T), the
// tap sum and the final contraction on the vector ALU, the result added to sqnorm[n] in fp64.
//
// Replaces pytorch-dp's per-sample gradient norm for these layers (train.py:325-334 PrivacyEngine; clip rule in
// primia_dp_clip_factors); same quantity as primia_conv2d_wgrad_persample + primia_persample_sqnorm to fp32 rounding.
#include <stdlib.h>

#include "common.h"

namespace primia {

struct GhostParams {
    const bf16* x;    // [N][49][C]
    const bf16* dy;   // [N][49][K]
    double* sq;       // [N], accumulated into
    int C, K;
};

constexpr int kGhHalo = 81;   // 9 x 9 halo pixels; row 81 = zeros
constexpr int kGhPix = 49;    // 7 x 7 pixels; dy row 49 = zeros

// Stage `rows` rows of `cpr` 16-byte chunks into LDS rows of `pitch` bytes, EIGHT chunks per thread in flight (a block is alone
// on its CU: one chunk at a time was a chain of ~20 memory round trips, 29 us per launch for 14 MFLOP).  src(row, ch)
// returns the global address of a chunk or nullptr for a zero chunk.
template <typename F>
__device__ __forceinline__ void ghost_stage(char* dst, int pitch, int rows, int cpr, F src) {
    const int total = rows * cpr;
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
        u32x4 v[8];
        int off[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u;
            v[u] = u32x4{0u, 0u, 0u, 0u};
            off[u] = -1;
            if (i < total) {
                const int row = i / cpr, ch = i - row * cpr;
                off[u] = row * pitch + ch * 16;
                const u32x4* g = src(row, ch);
                if (g) v[u] = *g;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (off[u] >= 0) *(u32x4*)(dst + off[u]) = v[u];
    }
}

// Gram tile row block `ti` (16 rows) against NTJ column blocks, rows / columns beyond `nrows` read the zero row
template <int NTJ>
__device__ __forceinline__ void gram_rows(const char* base, int pitch, int nrows, int klen, int ti, int lane, f32x4 (&acc)[NTJ]) {
    const int fr = lane & 15, fg = lane >> 4;
    const int ra = 16 * ti + fr;
    const char* pa = base + (ra < nrows ? ra : nrows) * pitch + fg * 16;
    const char* pb[NTJ];
#pragma unroll
    for (int tj = 0; tj < NTJ; ++tj) {
        const int rb = 16 * tj + fr;
        pb[tj] = base + (rb < nrows ? rb : nrows) * pitch + fg * 16;
        acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int k = 0; k < klen; k += 32) {
        const bf16x8_t a = *(const bf16x8_t*)(pa + k * 2);
#pragma unroll
        for (int tj = 0; tj < NTJ; ++tj) {
            const bf16x8_t b = *(const bf16x8_t*)(pb[tj] + k * 2);
            acc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[tj], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(256) void dp_ghost_sqnorm7_kernel(GhostParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int C = p.C, K = p.K;
    const int px = C * 2 + 16, pd = K * 2 + 16;            // row pitches in bytes
    char* const sx = smem;                                   // 82 rows
    char* const sd = sx + (kGhHalo + 1) * px;               // 50 rows; later: XX [81][84] fp32
    constexpr int G1P = 52, XXP = 84;
    const int sdb = (kGhPix + 1) * pd > kGhHalo * XXP * 4 ? (kGhPix + 1) * pd : kGhHalo * XXP * 4;   // dy rows, later XX
    float* const g1 = (float*)(sd + sdb);                    // [49][52] fp32

    // ---- stage: halo rows of x (border and row 81: zeros), rows of dy (row 49: zeros) ---------------------------------
    {
        const bf16* xs = p.x + (long)n * kGhPix * C;
        ghost_stage(sx, px, kGhHalo + 1, C / 8, [&](int row, int ch) -> const u32x4* {
            const int hy = row / 9, hx = row - hy * 9;
            if (row < kGhHalo && hy >= 1 && hy <= 7 && hx >= 1 && hx <= 7)
                return (const u32x4*)(xs + ((hy - 1) * 7 + hx - 1) * C + ch * 8);
            return nullptr;
        });
        const bf16* ds = p.dy + (long)n * kGhPix * K;
        ghost_stage(sd, pd, kGhPix + 1, K / 8, [&](int row, int ch) -> const u32x4* {
            return row < kGhPix ? (const u32x4*)(ds + row * K + ch * 8) : nullptr;
        });
    }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    // ---- G1 = dy dy^T (49 x 49 in a 64 x 64 grid of 16 tiles: wave w = tile row w) -------------------------------------
    {
        f32x4 acc[4];
        gram_rows<4>(sd, pd, kGhPix, K, wave, lane, acc);
        // lane (fr, fg) holds rows 4 fg .. +3 of column fr of each tile
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 16 * wave + 4 * fg + e, c = 16 * tj + fr;
                if (r < kGhPix && c < kGhPix) g1[r * G1P + c] = acc[tj][e];
            }
    }
    // ---- XX = X X^T (81 x 81 in a 96 x 96 grid: wave w = tile rows w and w + 4), kept in registers until dy is dead ------
    f32x4 xa[6], xb[6];
    gram_rows<6>(sx, px, kGhHalo, C, wave, lane, xa);
    if (wave < 2) gram_rows<6>(sx, px, kGhHalo, C, wave + 4, lane, xb);
    __syncthreads();                                         // every wave has read its dy rows: the region becomes XX
    float* const xx = (float*)sd;
#pragma unroll
    for (int tj = 0; tj < 6; ++tj)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 16 * tj + fr;
            const int r0 = 16 * wave + 4 * fg + e, r1 = r0 + 64;
            if (r0 < kGhHalo && c < kGhHalo) xx[r0 * XXP + c] = xa[tj][e];
            if (wave < 2 && r1 < kGhHalo && c < kGhHalo) xx[r1 * XXP + c] = xb[tj][e];
        }
    __syncthreads();
    // ---- sum_{p,q} G1[p][q] * sum_taps XX[h(p) + t][h(q) + t] -------------------------------------------------------------
    double part = 0.0;
    for (int i = tid; i < kGhPix * kGhPix; i += 256) {
        const int pp = i / kGhPix, qq = i - pp * kGhPix;
        const int hp = (pp / 7) * 9 + pp % 7, hq = (qq / 7) * 9 + qq % 7;
        float g2 = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) g2 += xx[(hp + 9 * r + s) * XXP + hq + 9 * r + s];
        part += (double)g1[pp * G1P + qq] * (double)g2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    __shared__ double wsum[4];
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (tid == 0) p.sq[n] += wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---- the stride-2 sibling: 14x14 input -> 7x7 output (a transition block's conv1) -------------------------------------
// u_p . u_q = sum over the 9 taps t of x[2p + t] . x[2q + t]: nine Gram matrices of the 49 pixels a tap samples, summed in
// the accumulators (a 225 x 225 Gram matrix of the whole 15 x 15 halo would not fit LDS).  dy is staged first, its Gram
// tiles stay in registers (wave w: tile row w, 16 registers), then the input halo takes dy's place in LDS; the contraction
// sum G1 * G2 happens in the accumulator layout, no Gram matrix is ever stored.
__global__ __launch_bounds__(256) void dp_ghost_sqnorm7s2_kernel(GhostParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n = blockIdx.x;
    const int C = p.C, K = p.K;
    const int px = C * 2 + 16, pd = K * 2 + 16;
    constexpr int HALO = 225;                               // 15 x 15: halo (hy, hx) = input (hy - 1, hx - 1); row 225 = zeros
    {
        const bf16* ds = p.dy + (long)n * kGhPix * K;
        ghost_stage(smem, pd, kGhPix + 1, K / 8, [&](int row, int ch) -> const u32x4* {
            return row < kGhPix ? (const u32x4*)(ds + row * K + ch * 8) : nullptr;
        });
    }
    __syncthreads();
    f32x4 g1[4];
    gram_rows<4>(smem, pd, kGhPix, K, wave, lane, g1);
    __syncthreads();                                         // dy is dead: the input halo takes its place
    {
        const bf16* xs = p.x + (long)n * 196 * C;
        ghost_stage(smem, px, HALO + 1, C / 8, [&](int row, int ch) -> const u32x4* {
            const int hy = row / 15, hx = row - hy * 15;
            if (row < HALO && hy >= 1 && hx >= 1) return (const u32x4*)(xs + ((hy - 1) * 14 + hx - 1) * C + ch * 8);
            return nullptr;
        });
    }
    __syncthreads();
    // halo row of output pixel q at tap (0, 0): (2 qy) * 15 + 2 qx; tap (r, s): + 15 r + s; pixels >= 49: the zero row
    auto hrow = [&](int q) { return q < kGhPix ? (2 * (q / 7)) * 15 + 2 * (q % 7) : -1; };
    const int ha = hrow(16 * wave + fr);
    int hb[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) hb[tj] = hrow(16 * tj + fr);
    f32x4 g2[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) g2[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 9; ++t) {
        const int off = 15 * (t / 3) + t % 3;
        const char* pa = smem + (ha < 0 ? HALO : ha + off) * px + fg * 16;
        const char* pb[4];
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) pb[tj] = smem + (hb[tj] < 0 ? HALO : hb[tj] + off) * px + fg * 16;
        for (int k = 0; k < C; k += 32) {
            const bf16x8_t a = *(const bf16x8_t*)(pa + k * 2);
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const bf16x8_t b = *(const bf16x8_t*)(pb[tj] + k * 2);
                g2[tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, g2[tj], 0, 0, 0);
            }
        }
    }
    double part = 0.0;      // (rows / columns beyond the 49 pixels are Gram entries of the zero row: zero)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int e = 0; e < 4; ++e) part += (double)g1[tj][e] * (double)g2[tj][e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    __shared__ double wsum[4];
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (tid == 0) p.sq[n] += wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// PRIMIA_ERR_UNSUPPORTED where the form does not apply (the caller keeps its per-sample weight-gradient pass)
static size_t ghost7_lds_bytes(int C, int K) {
    size_t sdb = (size_t)(kGhPix + 1) * (K * 2 + 16);
    if (sdb < (size_t)kGhHalo * 84 * 4) sdb = (size_t)kGhHalo * 84 * 4;
    return (size_t)(kGhHalo + 1) * (C * 2 + 16) + sdb + (size_t)kGhPix * 52 * 4;
}

// which Gram-matrix kernel serves the norm pass of this layer: 0 none, 21 dp_ghost_sqnorm7_kernel (7x7, stride 1),
// 22 dp_ghost_sqnorm7s2_kernel (14x14 -> 7x7, stride 2)
int dp_ghost_kernel_id(int H, int W, int C, int K, int R, int S, int stride, int pad) {
    if (!PRIMIA_OPT(dp_ghost) || R != 3 || S != 3 || pad != 1 || C % 32 || K % 32 || C < 32 || K < 32) return 0;
    if (H == 14 && W == 14 && stride == 2) {
        size_t a = (size_t)226 * (C * 2 + 16), b = (size_t)(kGhPix + 1) * (K * 2 + 16);
        return (a > b ? a : b) > 160 * 1024 ? 0 : 22;
    }
    // (14 x 14 / stride 1 as Gram matrices was built and measured SLOWER than the per-sample weight-gradient pass it would
    // replace — DP-SGD step 6.42 -> 6.56 ms, profiles/r03_negative_results.txt — and is not kept)
    if (H != 7 || W != 7 || stride != 1) return 0;
    return ghost7_lds_bytes(C, K) > 160 * 1024 ? 0 : 21;
}

int dp_ghost_sqnorm_dispatch(const void* x, const void* dy, double* sq, int N, int H, int W, int C, int K, int R, int S,
                             int stride, int pad, hipStream_t st) {
    const int id = dp_ghost_kernel_id(H, W, C, K, R, S, stride, pad);
    if (!id) return PRIMIA_ERR_UNSUPPORTED;
    GhostParams p{(const bf16*)x, (const bf16*)dy, sq, C, K};
    if (id == 22) {
        size_t a = (size_t)226 * (C * 2 + 16), b = (size_t)(kGhPix + 1) * (K * 2 + 16);
        const size_t lds2 = a > b ? a : b;
        static size_t lds2_set = 0;
        if (lds2 > lds2_set) {
            if (hipFuncSetAttribute((const void*)dp_ghost_sqnorm7s2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds2) != hipSuccess)
                return PRIMIA_ERR_LAUNCH;
            lds2_set = lds2;
        }
        dp_ghost_sqnorm7s2_kernel<<<N, 256, lds2, st>>>(p);
        return launch_status();
    }
    const size_t lds = ghost7_lds_bytes(C, K);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute((const void*)dp_ghost_sqnorm7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        lds_set = lds;
    }
    dp_ghost_sqnorm7_kernel<<<N, 256, lds, st>>>(p);
    return launch_status();
}

}  // namespace primia
