// Interface of the linear-halo 3x3 kernel (conv3x3_lh2.hip) towards the dispatch code in conv_igemm.hip.
#pragma once
#include "conv_common.h"

namespace primia {

// Linear-halo 3x3 / stride-1 kernel (conv3x3_lh2.hip): persistent 392- / 196-pixel tiles.  (Its first generation,
// conv3x3_lh.hip — one tile per block, 74.8 us per launch against 55.6 — was superseded in round 3 and removed.)
// the BatchNorm whose backward sums a plain data-gradient launch forms in its write-back (lh_tile_writeback below)
struct LhBnBwd {
    const bf16* y;          // [M][Nd], the BatchNorm's input (null: forward statistics)
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
};

// ... and of an ACCUMULATING 64 -> 64 data gradient (conv3x3_c64.hip): mode 2 = a residual BatchNorm, ReLU from its stored mask
// bytes (aux = its input y, c0 = mean, c1 = invstd); mode 3 = the stem's BatchNorm seen through the max-pool (aux = the pooled
// activation p, [p > 0], xhat = (p - beta) / gamma: c0 = beta, c1 = gamma)
struct C64AccBnb {
    int mode;
    const bf16* aux;
    const uint8_t* mask;
    const float* c0;
    const float* c1;
};

// pixel tiles (= partial slots) if the shape is served by the kernel, else PRIMIA_ERR_UNSUPPORTED
int conv3x3_lh2_tiles_m(int N, int H, int W, int Cs, int Nd);
int conv3x3_lh_kernel_of(int N, int H, int W, int Cs, int Nd);   // 4 conv3x3_lh2_kernel | 6 conv3x3_lh4_kernel | 0 neither
int conv3x3_lh2_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int Cs, int Nd, int flip,
                         int accumulate, hipStream_t st, float* stat_partials = nullptr,
                         const uint8_t* acc_mask = nullptr, const LhBnBwd* bnb = nullptr);

// Transition blocks on the parity planes (conv_s2lh.hip): 3x3 / stride 2 / pad 1 and 1x1 / stride 2 forward (one launch, either
// filter may be null) and data gradient (the downsample's optional).  PRIMIA_ERR_UNSUPPORTED where conv_s2lh_ok() is false.
bool conv_s2lh_ok(int N, int H, int W, int C, int K);
int conv_s2lh_tiles_m(int N, int H, int W);
int conv_s2lh_fwd(const bf16* x, const bf16* w, bf16* y, float* stat, const bf16* w_ds, bf16* y_ds, float* stat_ds, int N,
                  int H, int W, int C, int K, hipStream_t st);
// (bnb: the residual BatchNorm in front of the block whose backward sums the write-back forms — 64-channel dx only; `sums`
// [conv_s2lh_tiles_m * 2][2][64] partials)
struct S2BnBwd {
    const bf16* y;
    const uint8_t* mask;
    const float* mean;
    const float* invstd;
    float* sums;
};
int conv_s2lh_dgrad(const bf16* dy, const bf16* wd, const bf16* dy_ds, const bf16* wd_ds, bf16* dx, int N, int H, int W, int C,
                    int K, hipStream_t st, const S2BnBwd* bnb = nullptr);


// =====================================================================================================================
// Device code shared by the linear-halo kernels (conv3x3_lh2.hip, conv3x3_lh4.hip; conv_s2lh.hip uses the helpers): the
// per-tile write-back from the accumulators — plain, accumulating and masked-accumulating forms — and the BatchNorm partial
// sums.  ONE copy: conv3x3_lh2_kernel and conv3x3_lh4_kernel must produce the same bits on the same tiles
// (tests/test_gpu_ops.py::test_loader_wave_kernel_is_bit_identical_to_the_linear_halo_kernel), and until round 5 they did
// so with two hand-synchronised copies of this code.
// =====================================================================================================================
#if defined(__HIPCC__)

// Chunk swizzle of every 64-byte LDS row (halo slots, weight rows): 16-byte chunk c of row r sits at c ^ lh_key(r >> 2)
// (conflict-free under the real ds_read_b128 lane groups at every tap shift: conv3x3_lh4.hip)
__device__ __forceinline__ int lh_key(int quad) { return (quad & 1) << 1; }

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane, fixed order
__device__ __forceinline__ float lh_row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}

// v_permlane16_swap: lanes 16-31 (48-63) of `a` trade places with lanes 0-15 (32-47) of `b`
__device__ __forceinline__ void lh_swap16(uint32_t& a, uint32_t& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// Write-back of one wave's share of a BM-pixel x 128-channel tile, straight from the accumulator registers.
// A lane (fr, fg) holds channels 16i + 4fg .. +3 (i = 0..3) of pixel 16 (F0 + j) + fr of the wave's 64 channels (half wn).
//  (1) per pair of fragments (i = 2b, 2b+1) a v_permlane16_swap between the lane rows fg = 2a and 2a+1 leaves an
//      even row with channels 16 (2b) + 8a .. +7 and an odd row with channels 16 (2b+1) + 8a .. +7: 16 contiguous
//      bytes per lane, i.e. pieces P0 (channels 0-31 of the wave's 64) and P1 (32-63) of the lane's pixel;
//  (2) the two 8-lane halves of every row trade pieces (DPP row_ror:8), so that ONE store instruction carries
//      complete 128-byte lines.  Stores go FIRST, the BatchNorm sums (of the values AS STORED) are formed while they drain:
//      one lane per fg parks the pixel group's partial in `scr_base` (LDS, [4 groups][2][128] floats), the caller's
//      stat_combine adds the four groups in a fixed order after the next barrier.
// ACC: dst += tile (old rows requested before the first store), `acc_mask` (or null): ReLU mask bits applied to the OLD values.
// NT: non-temporal stores in the plain form.  SKIP: measurement builds only (accumulators kept alive, nothing written).
// BNB (data-gradient launches, plain form): the tile just written is dz, the gradient w.r.t. z = relu(bn(y)) of the layer BEFORE
// this convolution; with `bnb.y` set the partial sums are the two sums that layer's BatchNorm backward needs —
//     sum g   and   sum g * xhat,   g = dz AS STORED * [bn(y) > 0],   xhat = (y - mean) * invstd
// — instead of the forward statistics, in the same [4 groups][2][128] scratch and the same per-tile partial table: the separate
// reduction pass over (y, dz) is dropped (primia_conv2d_dgrad_bnsums + primia_bn_relu_bwd_from_sums; round 5).  The mask is the
// expression bn_bwd_apply_kernel recomputes (one fma on y - mean with scale = invstd * gamma), xhat is BwdFn's.

template <int BM, int JW, int F0, bool ACC, bool NT, bool SKIP, bool BNB = false>
__device__ __forceinline__ void lh_tile_writeback(f32x4 (&acc)[4][JW], bf16* dst, const uint8_t* acc_mask, const bool stats,
                                                  float* scr_base, const int M, const int Nd, const int m0, const int n0,
                                                  const int wn, const int fr, const int fg, const LhBnBwd bnb = LhBnBwd{}) {
        if (SKIP) {
#pragma unroll
            for (int j = 0; j < JW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        unsigned row0 = (unsigned)(m0 + 16 * F0 + (fr & 7));
        const int hi8 = fr >> 3;                            // 0: this lane stores P0 pieces, 1: P1 pieces
        // this lane's 16-byte piece: fragment 2 hi8 + (fg & 1), channels 8 (fg >> 1) .. +7 of it
        const unsigned col0 = (unsigned)(n0 + wn * 64 + 32 * hi8 + 16 * (fg & 1) + 8 * (fg >> 1));
        // (m0 / n0 are known when the tile starts: without this the store addresses are computed there and stay in
        // registers through the whole main loop)
        asm volatile("" : "+v"(row0));
        auto ror8 = [](uint32_t v) {
            return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);   // row_ror:8
        };
        // accumulate form: EVERY old row (and mask byte) of the tile is requested before the first store — a load that
        // is waited for after a store also waits for that store (one vmcnt for both), and the compiler cannot move a
        // load of dst above a store to dst: fragment by fragment, the tile paid seven store -> load round trips (the
        // fragment registers of the main loop are dead here: 70 registers are free)
        u32x4 oldA_[JW], oldB_[JW];
        unsigned mkA_[JW], mkB_[JW];
        if constexpr (ACC) {
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const int plA = 16 * (F0 + j) + (fr & 7), plB = plA + 8;
                const bool okA = plA < BM && m0 + plA < M, okB = plB < BM && m0 + plB < M;
                const unsigned eoA = (row0 + 16 * j) * (unsigned)Nd + col0, eoB = eoA + 8u * (unsigned)Nd;
                oldA_[j] = oldB_[j] = u32x4{0u, 0u, 0u, 0u};
                mkA_[j] = mkB_[j] = 0xffu;
                if (okA) {
                    oldA_[j] = *(const u32x4*)((const char*)dst + (size_t)(eoA * 2u));
                    if (acc_mask) mkA_[j] = acc_mask[eoA >> 3];
                }
                if (okB) {
                    oldB_[j] = *(const u32x4*)((const char*)dst + (size_t)(eoB * 2u));
                    if (acc_mask) mkB_[j] = acc_mask[eoB >> 3];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            // rows of this lane in the two stores of fragment j: pixels (fr & 7) and 8 + (fr & 7)
            const int plA = 16 * (F0 + j) + (fr & 7), plB = plA + 8;
            const bool okA = plA < BM && m0 + plA < M, okB = plB < BM && m0 + plB < M;
            const unsigned eoA = (row0 + 16 * j) * (unsigned)Nd + col0, eoB = eoA + 8u * (unsigned)Nd;
            u32x4 oldA = {0u, 0u, 0u, 0u}, oldB = {0u, 0u, 0u, 0u};
            unsigned mkA = 0xffu, mkB = 0xffu;
            if constexpr (ACC) {
                oldA = oldA_[j]; oldB = oldB_[j];
                mkA = mkA_[j]; mkB = mkB_[j];
            }
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[i][j];
            if constexpr (ACC) {
                // the old values travel the two exchanges backwards into the accumulators' lane layout, are added in
                // fp32 and the sum is rounded once
                auto masked = [](u32x4 o, unsigned mk) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] &= ((mk >> (2 * e)) & 1u ? 0x0000ffffu : 0u) | ((mk >> (2 * e + 1)) & 1u ? 0xffff0000u : 0u);
                    return o;
                };
                const u32x4 a_ = masked(oldA, mkA), b_ = masked(oldB, mkB);
                // lanes fr < 8 hold (own P0 of pixel fr in A, P0 of pixel fr + 8 in B); lanes fr >= 8 hold (P1 of pixel
                // fr - 8 in A, own P1 in B): what is not the lane's own pixel goes back across the row halves
                u32x4 p0, p1;       // this lane's pixel: pieces P0 and P1 (post-swap layout)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t give = hi8 ? a_[e] : b_[e];
                    const uint32_t got = ror8(give);
                    p0[e] = hi8 ? got : a_[e];
                    p1[e] = hi8 ? b_[e] : got;
                }
#pragma unroll
                for (int bq = 0; bq < 2; ++bq) {
                    const u32x4 ov = bq ? p1 : p0;
                    uint32_t x0 = ov[0], x1 = ov[1], y0 = ov[2], y1 = ov[3];   // (x | y) = quarters (2a | 2a+1) of a fragment
                    lh_swap16(x0, y0);
                    lh_swap16(x1, y1);
                    // now x = fragment 2bq, y = fragment 2bq+1, both this lane's own quarter fg
                    const uint32_t ox[2][2] = {{x0, x1}, {y0, y1}};
#pragma unroll
                    for (int s_ = 0; s_ < 2; ++s_) {
                        const int i = 2 * bq + s_;
                        v[i][0] += __uint_as_float(ox[s_][0] << 16);
                        v[i][1] += __uint_as_float(ox[s_][0] & 0xffff0000u);
                        v[i][2] += __uint_as_float(ox[s_][1] << 16);
                        v[i][3] += __uint_as_float(ox[s_][1] & 0xffff0000u);
                    }
                }
            }
            u32x4 pc[2];        // P0, P1 of this lane's pixel
#pragma unroll
            for (int bq = 0; bq < 2; ++bq) {
                uint32_t x0 = (uint32_t)f32_to_bf16(v[2 * bq][0]) | ((uint32_t)f32_to_bf16(v[2 * bq][1]) << 16);
                uint32_t x1 = (uint32_t)f32_to_bf16(v[2 * bq][2]) | ((uint32_t)f32_to_bf16(v[2 * bq][3]) << 16);
                uint32_t y0 = (uint32_t)f32_to_bf16(v[2 * bq + 1][0]) | ((uint32_t)f32_to_bf16(v[2 * bq + 1][1]) << 16);
                uint32_t y1 = (uint32_t)f32_to_bf16(v[2 * bq + 1][2]) | ((uint32_t)f32_to_bf16(v[2 * bq + 1][3]) << 16);
                lh_swap16(x0, y0);
                lh_swap16(x1, y1);
                pc[bq] = u32x4{x0, x1, y0, y1};
            }
            u32x4 stA, stB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t give = hi8 ? pc[0][e] : pc[1][e];       // the piece the other half of the row stores
                const uint32_t got = ror8(give);
                stA[e] = hi8 ? got : pc[0][e];                         // pixel fr & 7:       P0 from lanes < 8, P1 via lanes >= 8
                stB[e] = hi8 ? pc[1][e] : got;                         // pixel 8 + (fr & 7)
            }
            // non-temporal stores for the plain forms (the accumulate form re-reads what it wrote a launch earlier): the
            // tile's 100 KB do not displace the weight and halo lines the next tile reads — same-box A/B, 20 + 200 steps:
            // family 1002 -> 1030 TFLOP/s, step 5.04 -> 5.00 ms (profiles/r04_lh3_experiments.txt)
            if (NT && !ACC) {
                if (okA) __builtin_nontemporal_store(stA, (u32x4*)((char*)dst + (size_t)(eoA * 2u)));
                if (okB) __builtin_nontemporal_store(stB, (u32x4*)((char*)dst + (size_t)(eoB * 2u)));
            } else {
                if (okA) *(u32x4*)((char*)dst + (size_t)(eoA * 2u)) = stA;
                if (okB) *(u32x4*)((char*)dst + (size_t)(eoB * 2u)) = stB;
            }
        }
        if (BNB && !ACC && stats && bnb.y) {
            float* scr = scr_base + (F0 == 0 ? 0 : (F0 - 1) / (BM == 392 ? 6 : 3)) * 256;   // pixel group 0..3
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c0 = n0 + wn * 64 + 16 * i + 4 * fg;       // this lane's four channels of fragment row i
                const f32x4 mu = *(const f32x4*)(bnb.mean + c0), is = *(const f32x4*)(bnb.invstd + c0);
                const f32x4 ga = *(const f32x4*)(bnb.gamma + c0), be = *(const f32x4*)(bnb.beta + c0);
                u32x2 yr[JW];
#pragma unroll
                for (int j = 0; j < JW; ++j) {
                    const int pl = 16 * (F0 + j) + fr;
                    const bool ok = pl < BM && m0 + pl < M;
                    yr[j] = u32x2{0u, 0u};
                    if (ok) yr[j] = *(const u32x2*)(bnb.y + ((size_t)(m0 + pl) * Nd + c0));
                }
                // ~9 vector instructions per element (the write-back is bound by their issue): the stored dz re-made two at a time
                // (one v_cvt_pk_bf16_f32), sum g * (y - mean) accumulated and scaled by invstd ONCE per channel below
                const f32x4 sc = {is[0] * ga[0], is[1] * ga[1], is[2] * ga[2], is[3] * ga[3]};
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < JW; ++j) {
                    const int pl = 16 * (F0 + j) + fr;
                    const bool ok = pl < BM && m0 + pl < M;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t pk = (uint32_t)f32_to_bf16(acc[i][j][2 * h]) | ((uint32_t)f32_to_bf16(acc[i][j][2 * h + 1]) << 16);
                        const float dzv[2] = {__uint_as_float(pk << 16), __uint_as_float(pk & 0xffff0000u)};
                        const float yv[2] = {__uint_as_float(yr[j][h] << 16), __uint_as_float(yr[j][h] & 0xffff0000u)};
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int e = 2 * h + u;
                            const float t = yv[u] - mu[e];
                            const float zz = __builtin_fmaf(t, sc[e], be[e]);
                            const float g = (ok && zz > 0.f) ? dzv[u] : 0.f;
                            s1[e] += g;
                            s2[e] = __builtin_fmaf(g, t, s2[e]);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = lh_row_sum(s1[e]), t2 = lh_row_sum(s2[e]) * is[e];
                    if (fr == 0) {
                        const int ch = wn * 64 + 16 * i + 4 * fg + e;
                        scr[ch] = t1;
                        scr[128 + ch] = t2;
                    }
                }
            }
        } else if (!ACC && stats) {
            // BatchNorm partial sums of the values AS STORED, formed while the stores drain: the accumulators are still
            // intact, rounding them again gives the stored bits.  Fold the 16 pixels of a row (same fg), then one lane per
            // fg parks the pixel group's partial in LDS; the B half adds the four groups in a fixed order after the next
            // barrier (stat_combine).
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 s1[4][2], s2[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) s1[i][h] = s2[i][h] = f32x2{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const int pl = 16 * (F0 + j) + fr;
                const bool ok = pl < BM && m0 + pl < M;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x2 r = {bf16_to_f32(f32_to_bf16(acc[i][j][2 * h])), bf16_to_f32(f32_to_bf16(acc[i][j][2 * h + 1]))};
                        if (!ok) r = f32x2{0.f, 0.f};
                        s1[i][h] += r;
                        s2[i][h] += r * r;
                    }
            }
            float* scr = scr_base + (F0 == 0 ? 0 : (F0 - 1) / (BM == 392 ? 6 : 3)) * 256;   // pixel group 0..3
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t1 = lh_row_sum(s1[i][e >> 1][e & 1]), t2 = lh_row_sum(s2[i][e >> 1][e & 1]);
                    if (fr == 0) {
                        const int ch = wn * 64 + 16 * i + 4 * fg + e;
                        scr[ch] = t1;
                        scr[128 + ch] = t2;
                    }
                }
        }
    }

#endif  // __HIPCC__

}  // namespace primia
