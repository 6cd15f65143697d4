// Implicit-GEMM convolution forward and data-gradient for NHWC tensors on gfx950 MFMA.
//
//   dst[m, n] = sum_{tap=(r,s)} sum_c  src[pix(m, tap), c] * wt[n, tap, c]
//
//   forward : dst = y [N*Ho*Wo, K], src = x,  pix = (ho*stride - pad + r, wo*stride - pad + s)
//   dgrad   : dst = dx [N*H*W, C],  src = dy, pix = ((h + pad - r)/stride, (w + pad - s)/stride)
//             taken only where the division is exact (wt is the [C][R][S][K] copy).
//
// Tiling: 256 threads = 4 waves (2 x 2); block tile BM pixels x BN channels; one k-step is 128
// bytes of the reduction axis (64 bf16 / 32 f32).  Both operand tiles are staged global -> VGPR
// -> LDS as rows of 128 B made of eight 16-B chunks; chunk c of row r lives at chunk slot
// c ^ ((r >> 1) & 7), which makes the ds_read_b128 fragment reads (row = lane & 15, chunk =
// 4*kk + (lane >> 4)) conflict-free for the hardware's 16-lane read groups.  The MFMA "A" operand
// is the weight tile and "B" the pixel tile, so each lane ends up with 4 consecutive output
// channels of one pixel and stores them with one 8-B (bf16) / 16-B (f32) store.
// LDS is double buffered: the global loads of step t+1 are in flight while step t's MFMAs run.
//
// f32 uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain) — the parity path; bf16 uses
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation — the throughput path.
#include <stdlib.h>

#include "conv3x3_lh.h"

namespace primia {

constexpr bool kUpfront = true;

struct IgemmParams {
    const void* src;
    const void* wt;
    void* dst;
    int Nb;          // batch
    int Hd, Wd, Nd;  // dst spatial size and channels
    int Hs, Ws, Cs;  // src spatial size and channels (Cs = 4 for the stem)
    int R, S, stride, pad;
    int klen;        // weight row length in elements
    int nsteps;      // k-steps
    long Md;         // dst pixels
    int accumulate;
    int ntile_n;
    float* stat_sums;  // optional [slots][2][Nd]: per-channel sum and sum of squares of the stored output
    int stat_tiles;    // 1: stat_sums holds one deterministic partial per pixel tile (written); 0: kStatSlots atomic slots
    int s2_classes;    // data-gradient of a stride-2 conv: dst pixels are processed in 4 parity classes
    int cls_inner;     // ... the class index is the fastest tile coordinate (option dgrad_cls_inner)
    int ntm_class;     // pixel tiles per class
    // Transition block (3x3/2 conv1 beside a 1x1/2 downsample, both reading the same x): the downsample's data
    // gradient lands on the even/even pixels only, where conv1's only tap is the centre one and reads the SAME dy
    // pixel — so it is that class's reduction axis made longer: K more elements from dy2 against wt2 [C][K].
    const void* src2;
    const void* wt2;
    // pixel index -> (n, h, w) without 64-bit division (set by launch_igemm): q = umulhi(m, magic), exact for
    // m * d < 2^32; fastdiv = 0 falls back to the long division
    unsigned magicW, magicH;
    int fastdiv;
    // data gradient (round 5): with bnb_y the write-back also forms the backward sums of the RESIDUAL BatchNorm whose output
    // gradient this launch writes (dst = dz): sum g, sum g * xhat with g = dz AS STORED * mask bit, one partial per tile into
    // stat_sums [tiles][2][Nd] (tile = pixel tile x parity class)
    const void* bnb_y;
    const unsigned char* bnb_mask;
    const float* bnb_mean;
    const float* bnb_invstd;
};

template <typename T>
struct MmaTraits;
template <>
struct MmaTraits<bf16> {
    static constexpr int KE = 64;  // elements per k-step
};
template <>
struct MmaTraits<float> {
    static constexpr int KE = 32;
};

// Fused BatchNorm statistics are spread over this many partial slots ([kStatSlots][2][C] floats).
constexpr int kStatSlots = 64;

// 16 zero bytes every padded / out-of-range DMA lane reads from.
__device__ __attribute__((aligned(16))) const unsigned char kZeroPage[16] = {0};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

typedef int ig_i32x4 __attribute__((ext_vector_type(4)));

// buffer-addressed LDS-DMA (bf16 launches): a 32-bit lane offset instead of a 64-bit pointer — one add and one select
// per staged row and k-step instead of seven vector instructions; an offset beyond num_records reads zeros
__device__ __forceinline__ void ig_bdma16(unsigned voff, ig_i32x4 rsrc, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ ig_i32x4 ig_rsrc(const void* base, long bytes) {
    const unsigned long long a = (unsigned long long)base;
    ig_i32x4 r;
    r[0] = (int)(unsigned)a;
    r[1] = (int)(unsigned)(a >> 32) & 0xffff;       // stride 0: raw buffer
    r[2] = (int)(unsigned)(bytes > 0xfffffff0L ? 0xfffffff0L : bytes);
    r[3] = 0x00020000;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_readfirstlane(r[j]);
    return r;
}

// WM x WN waves per block (pixels x channels); STAGES LDS buffers (STAGES-1 k-steps of DMA in flight).
template <typename T, int BM, int BN, int WM, int WN, int STAGES, bool DGRAD, bool STEM>
__device__ __forceinline__ void conv_igemm_body(const IgemmParams& p, int bid, int nblk) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int KE = MmaTraits<T>::KE;
    constexpr int CH = Elem<T>::kPerChunk;  // elements per 16-B chunk
    constexpr int PR = BM * 8 / NT;         // pixel rows (= 16-B chunks) staged per thread
    constexpr int WR = BN * 8 / NT;         // weight rows staged per thread
    constexpr int FM = BN / WN / 16;        // 16-channel fragments per wave
    constexpr int FN = BM / WM / 16;        // 16-pixel fragments per wave
    constexpr int TILE_P = BM * 128, TILE_W = BN * 128;
    static_assert(PR >= 1 && WR >= 1 && FM >= 1 && FN >= 1, "tile too small for the wave grid");
    static_assert(!STEM || (NT == 256 && STAGES == 2), "stem path is written for 256 threads, 2 stages");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // buffer b: pixel tile at smem + b*(TILE_P+TILE_W), weight tile right behind it.

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;  // wave position: pixels, channels

    const int tile = xcd_remap(bid, nblk);
    const int tn = tile % p.ntile_n;
    int tm = tile / p.ntile_n;
    const int n0 = tn * BN;
    // Stride-2 data-gradient: a dst pixel (h, w) only receives taps with r = (h + pad) mod 2, s = (w + pad)
    // mod 2, so the pixels are split into 4 parity classes, each a dense problem over its own taps
    // (1 + 2 + 2 + 4 = 9 tap-GEMMs on quarter-size pixel sets instead of 9 on the full set).
    int cls_ph = 0, cls_pw = 0;
    int slot_id = tm;
    if (DGRAD && p.s2_classes) {
        // class-major (all tiles of class 0, then class 1, ...) or, option dgrad_cls_inner, position-major: the four classes of
        // a pixel tile are dispatched back to back, so that the dy rows they share are fetched into the XCD's L2 once
        slot_id = tm;          // unique per (pixel tile, class): the row of a BatchNorm-sums partial table
        int cls;
        if (p.cls_inner) {
            cls = tm & 3;
            tm >>= 2;
        } else {
            cls = tm / p.ntm_class;
            tm -= cls * p.ntm_class;
        }
        cls_ph = cls >> 1;
        cls_pw = cls & 1;
        // a class without taps (1x1 / stride 2: three of the four) adds nothing: when accumulating, leave its
        // pixels alone instead of reading and rewriting them
        if (p.accumulate && (cls_ph >= p.R || cls_pw >= p.S)) return;
    }
    const long m0 = (long)tm * BM;
    const int Hc = (DGRAD && p.s2_classes) ? p.Hd >> 1 : p.Hd, Wc = (DGRAD && p.s2_classes) ? p.Wd >> 1 : p.Wd;
    const long Mc = (DGRAD && p.s2_classes) ? (long)p.Nb * Hc * Wc : p.Md;
    // row index inside the (class) pixel set -> dst pixel (n, hd, wd)
    // (the 64-bit division this replaces cost ~100 vector instructions a piece; with 2 + 2..4 calls per thread it was
    // most of the 900 vector instructions per wave and tile of the short-reduction launches — the stride-2 data-gradient
    // classes, the 1x1 layers — next to 64 MFMAs: profiles/r03_inst_mix.txt)
    auto dst_pixel = [&](long m, int& n, int& hd, int& wd) {
        int w2, h2;
        if (p.fastdiv) {
            const unsigned mu = (unsigned)m;
            const unsigned t = __umulhi(mu, p.magicW);
            w2 = (int)(mu - t * (unsigned)Wc);
            const unsigned nn = __umulhi(t, p.magicH);
            h2 = (int)(t - nn * (unsigned)Hc);
            n = (int)nn;
        } else {
            w2 = (int)(m % Wc);
            const long t = m / Wc;
            h2 = (int)(t % Hc);
            n = (int)(t / Hc);
        }
        if (DGRAD && p.s2_classes) {
            hd = 2 * h2 + ((cls_ph + p.pad) & 1);
            wd = 2 * w2 + ((cls_pw + p.pad) & 1);
        } else {
            hd = h2;
            wd = w2;
        }
    };

    const T* __restrict__ src = (const T*)p.src;
    const T* __restrict__ wt = (const T*)p.wt;
    const bool pair = DGRAD && !STEM && p.s2_classes && p.src2 && cls_ph == 1 && cls_pw == 1;  // block-uniform

    // ---- per-thread staging coordinates -----------------------------------------------------
    // Register staging (stem): thread t owns rows t/8 + 32j, chunk t%8.
    // LDS-DMA staging (all other convs): one global_load_lds_dwordx4 moves 8 rows x 128 B per
    // wave; wave w issues groups w*PR + j, lane l lands in row 8*(w*PR+j) + l/8, chunk SLOT l%8,
    // and — the destination being lane-linear — fetches the global chunk slot ^ ((row>>1)&7), so
    // the LDS image carries the same XOR swizzle the fragment reads expect.
    constexpr bool GLDS = !STEM;
    const int srow = GLDS ? 0 : (tid >> 3), schunk = tid & 7;
    int nb[PR], hb[PR], wb[PR];
    bool mval[PR];
#pragma unroll
    for (int j = 0; j < PR; ++j) {
        const int row = GLDS ? ((wid * PR + j) * 8 + (lane >> 3)) : (srow + 32 * j);
        long m = m0 + row;
        mval[j] = m < Mc;
        if (!mval[j]) m = 0;
        int n, hd, wd;
        dst_pixel(m, n, hd, wd);
        nb[j] = n * p.Hs * p.Ws;
        if (DGRAD) {
            hb[j] = hd + p.pad;
            wb[j] = wd + p.pad;
        } else {
            hb[j] = hd * p.stride - p.pad;
            wb[j] = wd * p.stride - p.pad;
        }
    }
    const T* wrow[WR];
#pragma unroll
    for (int j = 0; j < WR; ++j) {
        if (GLDS) {
            const int row = (wid * WR + j) * 8 + (lane >> 3);
            wrow[j] = wt + (long)(n0 + row) * p.klen + ((lane & 7) ^ ((row >> 1) & 7)) * CH;
        } else {
            wrow[j] = wt + (long)(n0 + srow + 32 * j) * p.klen + schunk * CH;
        }
    }

    const T* wrow2[WR];
#pragma unroll
    for (int j = 0; j < WR; ++j) {
        const int row = (wid * WR + j) * 8 + (lane >> 3);
        wrow2[j] = (const T*)p.wt2 + (long)(n0 + row) * p.Cs + ((lane & 7) ^ ((row >> 1) & 7)) * CH;
    }

    u32x4 rp[PR], rw[WR];

    // ---- LDS-DMA stage ------------------------------------------------------------------------
    // Everything that depends on the lane is folded ONCE into (poff, pmask) per staged row:
    //   source element offset of tap (r,s), channel c0  =  poff[j] + tapoff(r,s) + c0
    //   row j takes part in tap t                        =  bit t of pmask[j]
    // with a wave-uniform tapoff: forward (r*Ws + s)*Cs; data-gradient -( (r/stride)*Ws + s/stride )*Cs
    // (for stride 2 the taps of the right parity are exactly those with (hb - r) even, and then
    // (hb - r)/2 = (hb >> 1) - (r >> 1)).  The per-step work is one add, one bit test and a select.
    int poff[PR];
    unsigned pmask[PR];
    if constexpr (GLDS) {
#pragma unroll
        for (int j = 0; j < PR; ++j) {
            const int row = (wid * PR + j) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            // tap (r, s) is valid iff its row is and its column is: R + S tests instead of R * S compound ones
            unsigned colm = 0, mask = 0;
            for (int s2 = 0; s2 < p.S; ++s2) {
                bool ok;
                if (DGRAD) {
                    const int tw = wb[j] - s2;
                    ok = tw >= 0 && (p.stride != 2 || (tw & 1) == 0) && (p.stride == 2 ? tw >> 1 : tw) < p.Ws;
                } else {
                    const int ws = wb[j] + s2;
                    ok = ws >= 0 && ws < p.Ws;
                }
                colm |= (ok ? 1u : 0u) << s2;
            }
            if (!mval[j]) colm = 0;
            for (int r = 0; r < p.R; ++r) {
                bool ok;
                if (DGRAD) {
                    const int th = hb[j] - r;
                    ok = th >= 0 && (p.stride != 2 || (th & 1) == 0) && (p.stride == 2 ? th >> 1 : th) < p.Hs;
                } else {
                    const int hs = hb[j] + r;
                    ok = hs >= 0 && hs < p.Hs;
                }
                if (ok) mask |= colm << (r * p.S);
            }
            pmask[j] = mask;
            const int h0 = (DGRAD && p.stride == 2) ? hb[j] >> 1 : hb[j];
            const int w0 = (DGRAD && p.stride == 2) ? wb[j] >> 1 : wb[j];
            poff[j] = (nb[j] + h0 * p.Ws + w0) * p.Cs + chunk * CH;
        }
    }
    // wave-uniform walk over the reduction axis: (tap, c0) and the tap's source offset
    const int tstep = (DGRAD && p.s2_classes) ? 2 : 1;  // class taps: r = ph, ph+2, ... ; s = pw, pw+2, ...
    int st_r = cls_ph, st_s = cls_pw, st_c0 = 0;
    int st_tap = st_r * p.S + st_s;
    int st_tapoff = DGRAD ? -(((p.stride == 2 ? st_r >> 1 : st_r) * p.Ws + (p.stride == 2 ? st_s >> 1 : st_s)) * p.Cs)
                          : (st_r * p.Ws + st_s) * p.Cs;
    int st_regular = 0;  // regular steps left before the paired tensor's steps (set with nsteps below)
    // bf16: buffer resources + lane-constant weight-row offsets (bytes)
    constexpr bool BUFDMA = GLDS && sizeof(T) == 2;
    constexpr unsigned kOob = 0xfffffff0u;
    ig_i32x4 rs_src = {0, 0, 0, 0}, rs_wt = {0, 0, 0, 0}, rs_src2 = {0, 0, 0, 0}, rs_wt2 = {0, 0, 0, 0};
    unsigned wbyte[WR], wbyte2[WR];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    if constexpr (BUFDMA) {
        rs_src = ig_rsrc(p.src, (long)p.Nb * p.Hs * p.Ws * p.Cs * 2);
        rs_wt = ig_rsrc(p.wt, (long)p.Nd * p.klen * 2);
        if (p.src2) {
            rs_src2 = ig_rsrc(p.src2, (long)p.Nb * p.Hs * p.Ws * p.Cs * 2);
            rs_wt2 = ig_rsrc(p.wt2, (long)p.Nd * p.Cs * 2);
        }
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            const int row = (wid * WR + j) * 8 + (lane >> 3);
            wbyte[j] = (unsigned)(((n0 + row) * p.klen + ((lane & 7) ^ ((row >> 1) & 7)) * CH) * 2);
            wbyte2[j] = (unsigned)(((n0 + row) * p.Cs + ((lane & 7) ^ ((row >> 1) & 7)) * CH) * 2);
        }
    }
    auto stage_glds = [&](int step, int buf) {
        (void)step;  // stages are issued in order; the walk state below IS the step
        if constexpr (BUFDMA) {
            const unsigned lpa = lds0 + buf * (TILE_P + TILE_W), lwa = lpa + TILE_P;
            if (DGRAD && pair && st_regular == 0) {
                const unsigned cbit = 1u << (p.S + 1);
#pragma unroll
                for (int j = 0; j < PR; ++j)
                    ig_bdma16((pmask[j] & cbit) ? (unsigned)((poff[j] + st_c0) * 2) : kOob, rs_src2,
                              __builtin_amdgcn_readfirstlane(lpa + (wid * PR + j) * 1024));
#pragma unroll
                for (int j = 0; j < WR; ++j)
                    ig_bdma16(wbyte2[j] + (unsigned)(st_c0 * 2), rs_wt2,
                              __builtin_amdgcn_readfirstlane(lwa + (wid * WR + j) * 1024));
                st_c0 += KE;
                return;
            }
            --st_regular;
            const int uoff = st_tapoff + st_c0;
            const unsigned tbit = 1u << st_tap;
#pragma unroll
            for (int j = 0; j < PR; ++j)
                ig_bdma16((pmask[j] & tbit) ? (unsigned)((poff[j] + uoff) * 2) : kOob, rs_src,
                          __builtin_amdgcn_readfirstlane(lpa + (wid * PR + j) * 1024));
            const unsigned woffb = (unsigned)((st_tap * p.Cs + st_c0) * 2);
#pragma unroll
            for (int j = 0; j < WR; ++j)
                ig_bdma16(wbyte[j] + woffb, rs_wt, __builtin_amdgcn_readfirstlane(lwa + (wid * WR + j) * 1024));
            // advance
            st_c0 += KE;
            if (st_c0 == p.Cs) {
                st_c0 = 0;
                st_s += tstep;
                if (st_s >= p.S) {
                    st_s = cls_pw;
                    st_r += tstep;
                }
                st_tap = st_r * p.S + st_s;
                if (DGRAD)
                    st_tapoff = -(((p.stride == 2 ? st_r >> 1 : st_r) * p.Ws + (p.stride == 2 ? st_s >> 1 : st_s)) * p.Cs);
                else
                    st_tapoff = (st_r * p.Ws + st_s) * p.Cs;
            }
            return;
        }
        char* lp = smem + buf * (TILE_P + TILE_W);
        char* lw = lp + TILE_P;
        if (DGRAD && pair && st_regular == 0) {
            // paired 1x1/2 tensor: the centre tap's pixel (tap offset 0 for stride 2), rows of wt2
            const T* __restrict__ s2p = (const T*)p.src2;
            const unsigned cbit = 1u << (p.S + 1);
#pragma unroll
            for (int j = 0; j < PR; ++j) {
                const T* g = (pmask[j] & cbit) ? s2p + (poff[j] + st_c0) : (const T*)kZeroPage;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lp + (wid * PR + j) * 1024),
                                                 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < WR; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow2[j] + st_c0),
                                                 (__attribute__((address_space(3))) void*)(lw + (wid * WR + j) * 1024),
                                                 16, 0, 0);
            st_c0 += KE;
            return;
        }
        --st_regular;
        const int uoff = st_tapoff + st_c0;
        const unsigned tbit = 1u << st_tap;
#pragma unroll
        for (int j = 0; j < PR; ++j) {
            const T* g = (pmask[j] & tbit) ? src + (poff[j] + uoff) : (const T*)kZeroPage;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lp + (wid * PR + j) * 1024),
                                             16, 0, 0);
        }
        const int woff = st_tap * p.Cs + st_c0;
#pragma unroll
        for (int j = 0; j < WR; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow[j] + woff),
                                             (__attribute__((address_space(3))) void*)(lw + (wid * WR + j) * 1024),
                                             16, 0, 0);
        // advance
        st_c0 += KE;
        if (st_c0 == p.Cs) {
            st_c0 = 0;
            st_s += tstep;
            if (st_s >= p.S) {
                st_s = cls_pw;
                st_r += tstep;
            }
            st_tap = st_r * p.S + st_s;
            if (DGRAD)
                st_tapoff = -(((p.stride == 2 ? st_r >> 1 : st_r) * p.Ws + (p.stride == 2 ? st_s >> 1 : st_s)) * p.Cs);
            else
                st_tapoff = (st_r * p.Ws + st_s) * p.Cs;
        }
    };

    auto load_global = [&](int step) {
        // register staging, stem only: element e0 of the reduction axis -> kernel row r, first
        // pixel sx inside the row; each 8-byte pixel is predicated on its own.
        const int e0 = step * KE + schunk * CH;
        const int r = e0 >> 5, sx = (e0 & 31) >> 2;
#pragma unroll
        for (int j = 0; j < PR; ++j) {
            const int hs = hb[j] + r;
            const bool rowok = mval[j] && r < p.R && hs >= 0 && hs < p.Hs;
            const int ws = wb[j] + sx;
            const long base = ((long)nb[j] + (long)hs * p.Ws + ws) * 4;
            if (sizeof(T) == 4) {
                u32x4 v = {0, 0, 0, 0};
                if (rowok && ws >= 0 && ws < p.Ws) v = *(const u32x4*)(src + base);
                rp[j] = v;
            } else {
                u32x2 a = {0, 0}, b = {0, 0};
                if (rowok && ws >= 0 && ws < p.Ws) a = *(const u32x2*)(src + base);
                if (rowok && ws + 1 >= 0 && ws + 1 < p.Ws) b = *(const u32x2*)(src + base + 4);
                rp[j] = u32x4{a[0], a[1], b[0], b[1]};
            }
        }
#pragma unroll
        for (int j = 0; j < WR; ++j) rw[j] = *(const u32x4*)(wrow[j] + (long)step * KE);
    };

    auto store_lds = [&](int buf) {
        char* lp = smem + buf * (TILE_P + TILE_W);
        char* lw = lp + TILE_P;
#pragma unroll
        for (int j = 0; j < PR; ++j) *(u32x4*)(lp + lds_off(srow + 32 * j, schunk)) = rp[j];
#pragma unroll
        for (int j = 0; j < WR; ++j) *(u32x4*)(lw + lds_off(srow + 32 * j, schunk)) = rw[j];
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int prow0 = wm * (BM / WM) + fr;  // + 16*fn
    const int crow0 = wn * (BN / WN) + fr;  // + 16*fm

    auto compute = [&](int buf) {
        const char* lp = smem + buf * (TILE_P + TILE_W);
        const char* lw = lp + TILE_P;
        if constexpr (sizeof(T) == 2) {
            // all fragment reads of the k-step (both 32-element halves) are issued before the first MFMA: the
            // LDS latency is paid once per k-step instead of once per half
            bf16x8_t a[2][FM], b[2][FN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    a[kk][i] = *(const bf16x8_t*)(lw + lds_off(crow0 + 16 * i, kk * 4 + fg));
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    b[kk][j] = *(const bf16x8_t*)(lp + lds_off(prow0 + 16 * j, kk * 4 + fg));
            }
            if (kUpfront) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        } else {
            // lane's k-set = floats of chunk fg and chunk fg+4 (any k order works as long as
            // both operands agree); MFMA t consumes float t of every lane.
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 a[FM], b[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    a[i] = *(const f32x4*)(lw + lds_off(crow0 + 16 * i, h * 4 + fg));
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    b[j] = *(const f32x4*)(lp + lds_off(prow0 + 16 * j, h * 4 + fg));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int j = 0; j < FN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop -----------------------------------------------------------------------------
    int nsteps = p.nsteps;
    if (DGRAD && p.s2_classes) {
        const int nr = cls_ph < p.R ? (p.R - cls_ph + 1) / 2 : 0, ns = cls_pw < p.S ? (p.S - cls_pw + 1) / 2 : 0;
        nsteps = nr * ns * (p.Cs / KE);
    }
    st_regular = nsteps;
    if (pair) nsteps += p.Cs / KE;
    if constexpr (GLDS) {
        // STAGES-deep LDS ring fed by LDS-DMA.  Per k-step ONE raw barrier: a counted vmcnt leaves the
        // newer stages' DMA in flight across it (a plain __syncthreads() would drain them).
        constexpr int NLOAD = PR + WR;  // DMA instructions per wave per stage
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nsteps) stage_glds(s, s);
        int cur = 0, nxt = STAGES - 1;
        for (int step = 0; step < nsteps; ++step) {
            // stages step+1 .. step+STAGES-2 may still be in flight
            int ahead = nsteps - 1 - step;
            if (ahead > STAGES - 2) ahead = STAGES - 2;
            if (STAGES >= 4 && ahead >= 2)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLOAD) : "memory");
            else if (STAGES >= 3 && ahead >= 1)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // stage `step` visible to all; buffer `nxt` no longer read
            if (step + STAGES - 1 < nsteps) stage_glds(step + STAGES - 1, nxt);
            compute(cur);
            cur = cur + 1 == STAGES ? 0 : cur + 1;
            nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
        }
    } else {
        load_global(0);
        store_lds(0);
        __syncthreads();
        for (int step = 0; step < p.nsteps; ++step) {
            const int cur = step & 1;
            const bool more = step + 1 < p.nsteps;
            if (more) load_global(step + 1);
            compute(cur);
            if (more) store_lds(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: lane holds 4 consecutive channels of one pixel per fragment -------------------
    // Optionally (stat_sums != null) the per-channel sum / sum of squares of the values AS STORED are
    // accumulated for the BatchNorm that follows: 16-lane shuffle reduction over the pixels a wave owns,
    // then one atomic per channel per wave — this replaces a full read pass over the conv output.
    T* __restrict__ dst = (T*)p.dst;
    if constexpr (sizeof(T) == 2 && GLDS) {
        if ((!p.stat_sums || p.stat_tiles || p.bnb_y) && !(p.Nd & 7) &&
            BM * (p.accumulate ? BN * 4 : BN * 2) <= STAGES * (BM + BN) * 128) {
            // bf16, no fused statistics: the tile leaves through LDS as whole 16-byte chunks of its pixel rows (the
            // 8-byte-per-lane stores below are store-issue bound: T21 of the programming guide; the stride-2 classes
            // scatter their rows two pixels apart, which makes it worse).  fp32 staging + ONE rounding when accumulating.
            __syncthreads();                       // every wave is done reading the last stage
            const int rb = p.accumulate ? BN * 4 : BN * 2;   // staged row bytes
            constexpr int CPR = BN / 8;            // 16-byte bf16 output chunks per row
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wm * (BM / WM) + 16 * j + fr;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int ch = wn * (BN / WN) + 16 * i + fg * 4;
                    if (p.accumulate) {
                        *(f32x4*)(smem + row * rb + (((ch >> 2) ^ (row & (2 * CPR - 1))) << 4)) = acc[i][j];
                    } else {
                        u32x2 o;
                        o[0] = (uint32_t)f32_to_bf16(acc[i][j][0]) | ((uint32_t)f32_to_bf16(acc[i][j][1]) << 16);
                        o[1] = (uint32_t)f32_to_bf16(acc[i][j][2]) | ((uint32_t)f32_to_bf16(acc[i][j][3]) << 16);
                        *(u32x2*)(smem + row * rb + (((ch >> 3) ^ (row & (CPR - 1))) << 4) + ((ch & 4) << 1)) = o;
                    }
                }
            }
            __syncthreads();
            // (forward, stat_tiles): BatchNorm partial sums of the values AS STORED, one deterministic partial per
            // pixel tile [tm][2][Nd] as the linear-halo kernel emits them — NT is a multiple of CPR, so a thread's 8
            // channels are the same in every trip
            float st1[8], st2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) st1[e] = st2[e] = 0.f;
            float bmu[8];
            if (DGRAD && p.bnb_y) {
#pragma unroll
                for (int e = 0; e < 8; ++e) bmu[e] = p.bnb_mean[n0 + (tid % CPR) * 8 + e];
            }
            for (int q = tid; q < BM * CPR; q += NT) {
                const int row = q / CPR, c8 = q - row * CPR;
                const long mc = m0 + row;
                if (mc >= Mc) continue;
                long m = mc;
                if (DGRAD && p.s2_classes) {
                    int n, hd, wd;
                    dst_pixel(mc, n, hd, wd);
                    m = ((long)n * p.Hd + hd) * p.Wd + wd;
                }
                T* gq = dst + m * p.Nd + n0 + c8 * 8;
                u32x4 v;
                if (p.accumulate) {
                    const u32x4 old = *(const u32x4*)gq;
                    const f32x4 lo = *(const f32x4*)(smem + row * rb + (((2 * c8) ^ (row & (2 * CPR - 1))) << 4));
                    const f32x4 hi = *(const f32x4*)(smem + row * rb + (((2 * c8 + 1) ^ (row & (2 * CPR - 1))) << 4));
                    float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[2 * e] += __uint_as_float(old[e] << 16);
                        f[2 * e + 1] += __uint_as_float(old[e] & 0xffff0000u);
                        v[e] = (uint32_t)f32_to_bf16(f[2 * e]) | ((uint32_t)f32_to_bf16(f[2 * e + 1]) << 16);
                    }
                } else {
                    v = *(const u32x4*)(smem + row * rb + ((c8 ^ (row & (CPR - 1))) << 4));
                }
                *(u32x4*)gq = v;
                if (DGRAD && p.bnb_y) {
                    const long eo = m * p.Nd + n0 + c8 * 8;
                    const u32x4 yv = *(const u32x4*)((const T*)p.bnb_y + eo);
                    const unsigned mk = p.bnb_mask[eo >> 3];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float yk = __uint_as_float((e & 1) ? (yv[e >> 1] & 0xffff0000u) : (yv[e >> 1] << 16));
                        const float dk = __uint_as_float((e & 1) ? (v[e >> 1] & 0xffff0000u) : (v[e >> 1] << 16));
                        const float gk = ((mk >> e) & 1u) ? dk : 0.f;
                        st1[e] += gk;
                        st2[e] = __builtin_fmaf(gk, yk - bmu[e], st2[e]);     // (x invstd below)
                    }
                } else if (p.stat_sums) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
                        st1[2 * e] += lo;
                        st2[2 * e] += lo * lo;
                        st1[2 * e + 1] += hi;
                        st2[2 * e + 1] += hi * hi;
                    }
                }
            }
            if (p.stat_sums) {
                __syncthreads();                   // the staged rows are dead
                float* red = (float*)smem;         // [NT / CPR][2][BN]
                const int grp = tid / CPR, cb = (tid % CPR) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(grp * 2 + 0) * BN + cb + e] = st1[e];
                    red[(grp * 2 + 1) * BN + cb + e] = st2[e];
                }
                __syncthreads();
                for (int c = tid; c < 2 * BN; c += NT) {
                    const int qq = c / BN, cl = c - qq * BN;
                    float a = 0.f;
                    for (int g = 0; g < NT / CPR; ++g) a += red[(g * 2 + qq) * BN + cl];
                    if (DGRAD && p.bnb_y) {
                        if (qq) a *= p.bnb_invstd[n0 + cl];
                        p.stat_sums[((long)slot_id * 2 + qq) * p.Nd + n0 + cl] = a;
                    } else {
                        p.stat_sums[((long)tm * 2 + qq) * p.Nd + n0 + cl] = a;
                    }
                }
            }
            return;
        }
    }
    float s1[FM][4], s2[FM][4];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) s1[i][t] = s2[i][t] = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const long mc = m0 + wm * (BM / WM) + 16 * j + fr;
        if (mc >= Mc) continue;
        long m = mc;
        if (DGRAD && p.s2_classes) {
            int n, hd, wd;
            dst_pixel(mc, n, hd, wd);
            m = ((long)n * p.Hd + hd) * p.Wd + wd;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int ch = n0 + wn * (BN / WN) + 16 * i + fg * 4;
            T* q = dst + m * p.Nd + ch;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (sizeof(T) == 4) {
                if (p.accumulate) {
                    const f32x4 o = *(const f32x4*)q;
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] += o[t];
                }
                *(f32x4*)q = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                if (p.accumulate) {
                    const u32x2 o = *(const u32x2*)q;
                    v[0] += __uint_as_float(o[0] << 16);
                    v[1] += __uint_as_float(o[0] & 0xffff0000u);
                    v[2] += __uint_as_float(o[1] << 16);
                    v[3] += __uint_as_float(o[1] & 0xffff0000u);
                }
                uint16_t h[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    h[t] = f32_to_bf16(v[t]);
                    v[t] = bf16_to_f32(h[t]);
                }
                u32x2 o;
                o[0] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
                o[1] = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                *(u32x2*)q = o;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s1[i][t] += v[t];
                s2[i][t] += v[t] * v[t];
            }
        }
    }
    if (p.stat_sums) {
        // block-level combine in LDS (the staging buffers are dead by now), then ONE global atomic per
        // channel per block into partial slot (tile % kStatSlots): keeps same-address contention low.
        float* red = (float*)smem;  // [2][BN]
        __syncthreads();
        for (int c = tid; c < 2 * BN; c += NT) red[c] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float a = s1[i][t], b = s2[i][t];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    a += __shfl_xor(a, o, 64);
                    b += __shfl_xor(b, o, 64);
                }
                if (fr == 0) {
                    const int cl = wn * (BN / WN) + 16 * i + fg * 4 + t;
                    atomicAdd(red + cl, a);
                    atomicAdd(red + BN + cl, b);
                }
            }
        __syncthreads();
        float* slot = p.stat_sums + (long)(tile % kStatSlots) * 2 * p.Nd;
        for (int c = tid; c < 2 * BN; c += NT) {
            const int q = c / BN, cl = c - q * BN;
            unsafeAtomicAdd(slot + q * p.Nd + n0 + cl, red[c]);
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES, bool DGRAD, bool STEM>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(IgemmParams p) {
    conv_igemm_body<T, BM, BN, WM, WN, STAGES, DGRAD, STEM>(p, blockIdx.x, gridDim.x);
}

// Two forward convolutions that read the same x in ONE launch (a transition block's 3x3 / 2 conv1 and its 1x1 / 2
// downsample): blocks [0, na) walk the tiles of a, the rest those of b.  The downsample alone is a launch of short
// tiles (one or a few k-steps) that never fills the chip for long — 24 / 15 / 11 us for 3.3 GFLOP each — behind the
// last, 77 %-full round of conv1's tiles; its tiles now run in that tail.
struct IgemmPair {
    IgemmParams a, b;
    int na;
};
template <typename T, int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_pair_kernel(IgemmPair pp) {
    const bool first = (int)blockIdx.x < pp.na;
    conv_igemm_body<T, BM, BN, WM, WN, STAGES, false, false>(first ? pp.a : pp.b, first ? (int)blockIdx.x : (int)blockIdx.x - pp.na,
                                                             first ? pp.na : (int)gridDim.x - pp.na);
}

// per-launch fields derived from the tile configuration; returns the grid
template <int BM, int BN, bool DGRAD>
static int igemm_finish_params(IgemmParams& q) {
    q.ntile_n = q.Nd / BN;
    int ntm = ceil_div(q.Md, BM);
    if (DGRAD && q.s2_classes) {
        q.ntm_class = ceil_div((long)q.Nb * (q.Hd / 2) * (q.Wd / 2), BM);
        ntm = 4 * q.ntm_class;
    }
    const int Hc = (DGRAD && q.s2_classes) ? q.Hd >> 1 : q.Hd, Wc = (DGRAD && q.s2_classes) ? q.Wd >> 1 : q.Wd;
    const long Mc = (DGRAD && q.s2_classes) ? (long)q.Nb * Hc * Wc : q.Md;
    const long dmax = Hc > Wc ? Hc : Wc;
    q.fastdiv = (Hc > 1 && Wc > 1 && (Mc + BM) * dmax < (1L << 32)) ? 1 : 0;
    q.magicW = q.fastdiv ? (unsigned)(((1ULL << 32) + Wc - 1) / Wc) : 0u;
    q.magicH = q.fastdiv ? (unsigned)(((1ULL << 32) + Hc - 1) / Hc) : 0u;
    return ntm * q.ntile_n;
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES>
static int launch_igemm_pair(const IgemmParams& pa, const IgemmParams& pb, hipStream_t st) {
    IgemmPair pp;
    pp.a = pa;
    pp.b = pb;
    pp.na = igemm_finish_params<BM, BN, false>(pp.a);
    const int nb = igemm_finish_params<BM, BN, false>(pp.b);
    const size_t lds = (size_t)STAGES * (BM + BN) * 128;
    auto kern = conv_igemm_pair_kernel<T, BM, BN, WM, WN, STAGES>;
    static bool attr_set = false;
    if (lds > 48 * 1024 && !attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    kern<<<pp.na + nb, 64 * WM * WN, lds, st>>>(pp);
    return launch_status();
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES, bool DGRAD, bool STEM>
static int launch_igemm(const IgemmParams& p, hipStream_t st) {
    IgemmParams q = p;
    const int grid = igemm_finish_params<BM, BN, DGRAD>(q);
    const size_t lds = (size_t)STAGES * (BM + BN) * 128;
    auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, STAGES, DGRAD, STEM>;
    static bool attr_set = false;  // once per instantiation (not per launch: keeps graph capture legal)
    if (lds > 48 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return PRIMIA_ERR_LAUNCH;
        attr_set = true;
    }
    kern<<<grid, 64 * WM * WN, lds, st>>>(q);
    return launch_status();
}

// Tile choice.  Default 'e': 128-pixel tile, 8 waves (4 x 2), double-buffered LDS-DMA, 2 blocks/CU —
// measured fastest on the ResNet-18 shapes (profiles/r01_conv_layers_*.txt): the kernel is bound by
// the L2 -> LDS fill rate per CU, so waves in flight beat pipeline depth (the 3-stage variants lose
// the second resident block).  option conv_cfg = 0..5 (a..f) selects an alternative for A/B runs:
//   a 128 px 2x2 waves 2 stages | b 128 px 2x2 3 stages | c 256 px 4x2 2 stages | d 256 px 4x2 3 stages
//   e 128 px 4x2 waves 2 stages | f 128 px 4x2 3 stages
template <typename T, bool DGRAD>
static int dispatch_igemm(const IgemmParams& p, bool stem, hipStream_t st) {
    if (stem) {
        if (DGRAD) return PRIMIA_ERR_UNSUPPORTED;
        return launch_igemm<T, 128, 64, 2, 2, 2, false, true>(p, st);
    }
    if ((long)p.Nb * p.Hs * p.Ws * p.Cs >= (1L << 31)) return PRIMIA_ERR_ARG;  // 32-bit element offsets
    const char cfg_env = (char)('a' + PRIMIA_OPT(conv_cfg));      // option conv_cfg 0..5 = a..f
    const char cfg = (p.stat_tiles || p.bnb_y) ? 'e' : cfg_env;   // per-tile statistics assume the 128-pixel tile
    const bool wide = p.Nd % 128 == 0;
#define PRIMIA_IGEMM_CASE(L, BM, WM_, WN_, ST)                                                   \
    case L:                                                                                      \
        return wide ? launch_igemm<T, BM, 128, WM_, WN_, ST, DGRAD, false>(p, st)                \
                    : launch_igemm<T, BM, 64, WM_, WN_, ST, DGRAD, false>(p, st);
    switch (cfg) {
        PRIMIA_IGEMM_CASE('a', 128, 2, 2, 2)
        PRIMIA_IGEMM_CASE('b', 128, 2, 2, 3)
        PRIMIA_IGEMM_CASE('c', 256, 4, 2, 2)
        PRIMIA_IGEMM_CASE('d', 256, 4, 2, 3)
        PRIMIA_IGEMM_CASE('e', 128, 4, 2, 2)
        PRIMIA_IGEMM_CASE('f', 128, 4, 2, 3)
        default:
            break;
    }
#undef PRIMIA_IGEMM_CASE
    return wide ? launch_igemm<T, 128, 128, 4, 2, 2, DGRAD, false>(p, st)
                : launch_igemm<T, 128, 64, 4, 2, 2, DGRAD, false>(p, st);
}

}  // namespace primia

using namespace primia;

namespace primia {
int conv3x3_c64_dispatch(const bf16* src, const bf16* wt, bf16* dst, int N, int H, int W, int flip, int accumulate,
                         hipStream_t st, float* stat_partials = nullptr, const uint8_t* acc_mask = nullptr,
                         const LhBnBwd* bnb = nullptr, const C64AccBnb* abnb = nullptr);
int conv3x3_c64_grid(int N, int H, int W);
}

// wide 3x3 / stride-1 layers (layer2-4): linear-halo kernel (conv3x3_lh2.hip); option lh2 = 0 keeps the implicit GEMM
static int lh_fwd_maxw() {   // forward only: widest image the linear-halo kernel takes (option lh_fwd_maxw)
    return PRIMIA_OPT(lh_fwd_maxw);
}
static bool lh_shape(const ConvGeom& g) {
    return !g.stem && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1;
}

// layer1 shape (3x3, stride 1, pad 1, 64 -> 64 channels, bf16): weight-stationary halo kernel (conv3x3_c64.hip);
// option c64 = 0 keeps the implicit GEMM (A/B measurements)
static bool use_c64(const ConvGeom& g) {
    return PRIMIA_OPT(c64) && !g.stem && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.C == 64 && g.K == 64;
}

// transition-block shapes served by conv_s2lh_kernel (bf16): 3x3 / 2 / pad 1 or 1x1 / 2 / pad 0 on an even-sized input
// option s2lh (bits): 1 data gradient where it wins in the training step (dx with <= s2lh_dx_max = 64 channels: layer2.0 —
// in-step medians at batch 256: 100 / 74 / 69 us on the implicit GEMM, 84 / 70 / 72 on conv_s2lh_kernel), 2 forward too
// (74 / 58 / 47 vs 85 / 85 / 95 us: off by default), 4 data gradient at every width.  Default 1.
static bool s2_pass_on(int pass, int dx_channels = 0) {
    const int o = PRIMIA_OPT(s2lh);
    if (pass == 0) return (o & 2) != 0;
    return (o & 4) != 0 || ((o & 1) != 0 && dx_channels <= PRIMIA_OPT(s2lh_dx_max));
}
static bool s2_conv1_shape(const ConvGeom& g) {
    return !g.stem && g.R == 3 && g.S == 3 && g.stride == 2 && g.pad == 1 && conv_s2lh_ok(g.N, g.H, g.W, g.C, g.K);
}
static bool s2_ds_shape(const ConvGeom& g) {
    return !g.stem && g.R == 1 && g.S == 1 && g.stride == 2 && g.pad == 0 && conv_s2lh_ok(g.N, g.H, g.W, g.C, g.K);
}

extern "C" {

static int conv2d_fwd_impl(const primia_conv_desc* d, const void* x, const void* w_fwd, void* y, float* stat_sums,
                           int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && x && w_fwd && y);
    ConvGeom g;
    PRIMIA_REQUIRE(g.init(*d));
    IgemmParams p;
    p.src = x; p.wt = w_fwd; p.dst = y;
    p.Nb = g.N; p.Hd = g.Ho; p.Wd = g.Wo; p.Nd = g.K;
    p.Hs = g.H; p.Ws = g.W; p.Cs = g.C;
    p.R = g.R; p.S = g.S; p.stride = g.stride; p.pad = g.pad;
    p.klen = g.klen;
    p.Md = (long)g.N * g.Ho * g.Wo;
    p.accumulate = 0;
    p.stat_sums = stat_sums;
    p.stat_tiles = (stat_sums && dtype == PRIMIA_BF16 && !g.stem) ? 1 : 0;
    p.s2_classes = 0;
    p.bnb_y = nullptr; p.bnb_mask = nullptr; p.bnb_mean = nullptr; p.bnb_invstd = nullptr;
    p.cls_inner = 0;
    p.ntm_class = 0;
    p.src2 = nullptr; p.wt2 = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32) {
        p.nsteps = g.stem ? 7 : g.klen / 32;
        return dispatch_igemm<float, false>(p, g.stem, st);
    } else if (dtype == PRIMIA_BF16) {
        if (use_c64(g)) {  // (its statistics are per-block partials, see primia_conv_stat_slots_for)
            const int rc = conv3x3_c64_dispatch((const bf16*)x, (const bf16*)w_fwd, (bf16*)y, g.N, g.H, g.W, 0, 0, st,
                                                stat_sums);
            if (rc != PRIMIA_ERR_UNSUPPORTED) return rc;
        } else if (lh_shape(g) && g.W <= lh_fwd_maxw()) {   // (statistics: per-block partials as well)
            const int rc2 = conv3x3_lh2_dispatch((const bf16*)x, (const bf16*)w_fwd, (bf16*)y, g.N, g.H, g.W, g.C, g.K, 0,
                                                 0, st, stat_sums);
            if (rc2 != PRIMIA_ERR_UNSUPPORTED) return rc2;
        } else if (s2_pass_on(0) && s2_conv1_shape(g)) {
            const int rc3 = conv_s2lh_fwd((const bf16*)x, (const bf16*)w_fwd, (bf16*)y, stat_sums, nullptr, nullptr, nullptr, g.N,
                                          g.H, g.W, g.C, g.K, st);
            if (rc3 != PRIMIA_ERR_UNSUPPORTED) return rc3;
        } else if (s2_pass_on(0) && s2_ds_shape(g)) {
            const int rc3 = conv_s2lh_fwd((const bf16*)x, nullptr, nullptr, nullptr, (const bf16*)w_fwd, (bf16*)y, stat_sums, g.N,
                                          g.H, g.W, g.C, g.K, st);
            if (rc3 != PRIMIA_ERR_UNSUPPORTED) return rc3;
        }
        p.nsteps = g.klen / 64;
        return dispatch_igemm<bf16, false>(p, g.stem, st);
    }
    return PRIMIA_ERR_ARG;
}

int primia_conv2d_fwd(const primia_conv_desc* d, const void* x, const void* w_fwd, void* y, int dtype,
                      primia_stream_t stream) {
    return conv2d_fwd_impl(d, x, w_fwd, y, nullptr, dtype, stream);
}

// a transition block's conv1 (3x3 / 2) and downsample (1x1 / 2) forward in one launch: both on the bf16 implicit GEMM,
// same input, output channels a multiple of 128
static bool fwd_pair_shape(const ConvGeom& g, const ConvGeom& gd, int dtype) {
    if (!PRIMIA_OPT(fwd_pair) || dtype != PRIMIA_BF16 || g.stem || gd.stem) return false;
    if (gd.N != g.N || gd.H != g.H || gd.W != g.W || gd.C != g.C || gd.Ho != g.Ho || gd.Wo != g.Wo) return false;
    if (g.stride != 2 || gd.stride != 2 || g.R != 3 || g.S != 3 || gd.R != 1 || gd.S != 1 || gd.pad != 0) return false;
    if (g.K % 128 || gd.K % 128) return false;
    return (long)g.N * g.H * g.W * g.C < (1L << 31);
}

int primia_conv_fwd_pair_ok(const primia_conv_desc* d, const primia_conv_desc* d_ds, int dtype) {
    ConvGeom g, gd;
    if (!d || !d_ds || !g.init(*d) || !gd.init(*d_ds)) return PRIMIA_ERR_ARG;
    return fwd_pair_shape(g, gd, dtype) ? 1 : 0;
}

int primia_conv2d_fwd_stats_pair(const primia_conv_desc* d, const void* x, const void* w_fwd, void* y, float* stat_sums,
                                 const primia_conv_desc* d_ds, const void* w_fwd_ds, void* y_ds, float* stat_sums_ds,
                                 int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && d_ds && x && w_fwd && y && w_fwd_ds && y_ds);
    ConvGeom g, gd;
    PRIMIA_REQUIRE(g.init(*d) && gd.init(*d_ds));
    if (!fwd_pair_shape(g, gd, dtype)) return PRIMIA_ERR_UNSUPPORTED;
    if (s2_pass_on(0) && s2_conv1_shape(g) && s2_ds_shape(gd) && g.pad == 1 && (stat_sums == nullptr) == (stat_sums_ds == nullptr)) {
        const int rc = conv_s2lh_fwd((const bf16*)x, (const bf16*)w_fwd, (bf16*)y, stat_sums, (const bf16*)w_fwd_ds,
                                     (bf16*)y_ds, stat_sums_ds, g.N, g.H, g.W, g.C, g.K, (hipStream_t)stream);
        if (rc != PRIMIA_ERR_UNSUPPORTED) return rc;
    }
    auto fill = [&](IgemmParams& p, const ConvGeom& c, const void* w, void* out, float* sums) {
        p.src = x; p.wt = w; p.dst = out;
        p.Nb = c.N; p.Hd = c.Ho; p.Wd = c.Wo; p.Nd = c.K;
        p.Hs = c.H; p.Ws = c.W; p.Cs = c.C;
        p.R = c.R; p.S = c.S; p.stride = c.stride; p.pad = c.pad;
        p.klen = c.klen;
        p.nsteps = c.klen / 64;
        p.Md = (long)c.N * c.Ho * c.Wo;
        p.accumulate = 0;
        p.stat_sums = sums;
        p.stat_tiles = sums ? 1 : 0;
        p.s2_classes = 0;
        p.bnb_y = nullptr; p.bnb_mask = nullptr; p.bnb_mean = nullptr; p.bnb_invstd = nullptr;
        p.cls_inner = 0;
        p.ntm_class = 0;
        p.src2 = nullptr; p.wt2 = nullptr;
    };
    IgemmParams pa, pb;
    fill(pa, g, w_fwd, y, stat_sums);
    fill(pb, gd, w_fwd_ds, y_ds, stat_sums_ds);
    return launch_igemm_pair<bf16, 128, 128, 4, 2, 2>(pa, pb, (hipStream_t)stream);
}

int primia_conv_kernel_id(const primia_conv_desc* d, int pass, int dtype) {
    ConvGeom g;
    if (!d || !g.init(*d) || (pass != 0 && pass != 1)) return PRIMIA_ERR_ARG;
    if (dtype != PRIMIA_BF16) return 1;
    if (g.stem) return 1;     // (the engine's bf16 stem runs primia_stem_conv_fwd on the padded input: stem_conv_fwd_kernel)
    if (use_c64(g) && (long)g.N * g.H * g.W * 64 < (1L << 31)) return 2;
    if (lh_shape(g) && (pass == 1 || g.W <= lh_fwd_maxw())) {
        const int cs = pass == 0 ? g.C : g.K, nd = pass == 0 ? g.K : g.C;
        const int lk = conv3x3_lh_kernel_of(g.N, g.H, g.W, cs, nd);
        if (lk) return lk;
    }
    if (s2_pass_on(pass, g.C) && (s2_conv1_shape(g) || (pass == 0 && s2_ds_shape(g)))) return 5;
    return 1;
}

int primia_conv_stat_slots(void) { return kStatSlots; }

// slots of the kernel serving this conv; *per_tile: 1 = deterministic per-tile partials WRITTEN by its write-back (nothing
// to zero, no extra pass), 0 = the kStatSlots atomic slots
static int conv_stat_slots_impl(const primia_conv_desc* d, int dtype, int* per_tile) {
    ConvGeom g;
    *per_tile = 0;
    if (!d || !g.init(*d)) return PRIMIA_ERR_ARG;
    *per_tile = 1;
    if (dtype == PRIMIA_BF16 && use_c64(g) && (long)g.N * g.H * g.W * 64 < (1L << 31)) return conv3x3_c64_grid(g.N, g.H, g.W);
    if (dtype == PRIMIA_BF16 && !use_c64(g) && lh_shape(g) && g.W <= lh_fwd_maxw()) {
        const int t2 = conv3x3_lh2_tiles_m(g.N, g.H, g.W, g.C, g.K);
        if (t2 > 0) return t2;
    }
    if (dtype == PRIMIA_BF16 && s2_pass_on(0) && (s2_conv1_shape(g) || s2_ds_shape(g))) return conv_s2lh_tiles_m(g.N, g.H, g.W);
    // bf16 implicit GEMM: one partial per 128-pixel tile out of its write-back (every tile config in use has BM = 128)
    if (dtype == PRIMIA_BF16 && !g.stem && g.K % 8 == 0) return (int)(((long)g.N * g.Ho * g.Wo + 127) / 128);
    *per_tile = 0;
    return kStatSlots;
}

int primia_conv_stat_slots_for(const primia_conv_desc* d, int dtype) {
    int per_tile;
    return conv_stat_slots_impl(d, dtype, &per_tile);
}

// 1: this conv's forward kernel writes the BatchNorm partial sums for free (per-tile, deterministic); 0: atomic slots.
// (The slot COUNT does not tell: layer4's 3x3 convs at batch 256 have 64 tiles, which is also kStatSlots.)
int primia_conv_stats_per_tile(const primia_conv_desc* d, int dtype) {
    int per_tile;
    const int rc = conv_stat_slots_impl(d, dtype, &per_tile);
    return rc < 0 ? rc : per_tile;
}

int primia_conv2d_fwd_stats(const primia_conv_desc* d, const void* x, const void* w_fwd, void* y, float* stat_sums,
                            int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(stat_sums);
    return conv2d_fwd_impl(d, x, w_fwd, y, stat_sums, dtype, stream);
}

static int conv2d_dgrad_impl(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                             int accumulate, const void* dy2, const void* w_dgrad2, int dtype, primia_stream_t stream,
                             const uint8_t* acc_mask = nullptr, const S2BnBwd* pair_bnb = nullptr,
                             const C64AccBnb* acc_bnb = nullptr, float* acc_sums = nullptr) {
    PRIMIA_REQUIRE(d && dy && w_dgrad && dx);
    ConvGeom g;
    PRIMIA_REQUIRE(g.init(*d));
    PRIMIA_REQUIRE(!g.stem && (g.stride == 1 || g.stride == 2));
    IgemmParams p;
    p.src2 = dy2; p.wt2 = w_dgrad2;
    p.src = dy; p.wt = w_dgrad; p.dst = dx;
    p.Nb = g.N; p.Hd = g.H; p.Wd = g.W; p.Nd = g.C;
    p.Hs = g.Ho; p.Ws = g.Wo; p.Cs = g.K;
    p.R = g.R; p.S = g.S; p.stride = g.stride; p.pad = g.pad;
    p.klen = g.R * g.S * g.K;
    p.Md = (long)g.N * g.H * g.W;
    p.accumulate = accumulate;
    p.stat_sums = nullptr;
    p.stat_tiles = 0;
    p.bnb_y = nullptr; p.bnb_mask = nullptr; p.bnb_mean = nullptr; p.bnb_invstd = nullptr;
    const bool no_classes = !PRIMIA_OPT(dgrad_classes);
    p.s2_classes = (g.stride == 2 && g.H % 2 == 0 && g.W % 2 == 0 && !no_classes) ? 1 : 0;
    p.cls_inner = PRIMIA_OPT(dgrad_cls_inner) ? 1 : 0;
    p.ntm_class = 0;
    if (p.src2 && !p.s2_classes) return PRIMIA_ERR_UNSUPPORTED;   // the pairing lives in the parity-class walk
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PRIMIA_F32) {
        p.nsteps = p.klen / 32;
        return dispatch_igemm<float, true>(p, false, st);
    } else if (dtype == PRIMIA_BF16) {
        if (use_c64(g)) {
            const int rc = conv3x3_c64_dispatch((const bf16*)dy, (const bf16*)w_dgrad, (bf16*)dx, g.N, g.H, g.W, 1,
                                                accumulate, st, acc_sums, acc_mask, nullptr, acc_bnb);
            if (rc != PRIMIA_ERR_UNSUPPORTED) return rc;
        }
        if (acc_bnb) return PRIMIA_ERR_UNSUPPORTED;     // only the 64 -> 64 accumulate form carries these sums
        if (!use_c64(g) && lh_shape(g) && !p.src2) {
            const int rc2 = conv3x3_lh2_dispatch((const bf16*)dy, (const bf16*)w_dgrad, (bf16*)dx, g.N, g.H, g.W, g.K, g.C,
                                                 1, accumulate, st, nullptr, acc_mask);
            if (rc2 != PRIMIA_ERR_UNSUPPORTED) return rc2;
        }
        if (!accumulate && !acc_mask && s2_pass_on(1, g.C) && s2_conv1_shape(g)) {     // transition block: conv1 (+ the downsample) on the parity planes
            const int rc3 = conv_s2lh_dgrad((const bf16*)dy, (const bf16*)w_dgrad, (const bf16*)dy2, (const bf16*)w_dgrad2,
                                            (bf16*)dx, g.N, g.H, g.W, g.C, g.K, st);
            if (rc3 != PRIMIA_ERR_UNSUPPORTED) return rc3;
        }
        if (acc_mask) return PRIMIA_ERR_UNSUPPORTED;   // only the 64->64 and linear-halo write-backs mask the old values
        if (pair_bnb) {
            p.bnb_y = pair_bnb->y; p.bnb_mask = pair_bnb->mask; p.bnb_mean = pair_bnb->mean; p.bnb_invstd = pair_bnb->invstd;
            p.stat_sums = pair_bnb->sums;
        }
        p.nsteps = p.klen / 64;
        return dispatch_igemm<bf16, true>(p, false, st);
    }
    return PRIMIA_ERR_ARG;
}

// The plain data gradient of a wide 3x3 / stride-1 layer that ALSO forms, in its write-back, the two sums the BatchNorm backward
// of the layer in front of it needs (dx = dz of that layer: sum g and sum g * xhat per channel, g = dz * [bn(y) > 0]) as per-tile
// partials — the separate reduction pass over (y, dz) is dropped (conv3x3_lh.h: LhBnBwd; primia_bn_relu_bwd_from_sums consumes
// them).  Slots of the partial table [slots][2][C], or 0 where the linear-halo kernels do not serve the shape.
int primia_conv_dgrad_bnsums_slots(const primia_conv_desc* d, int dtype) {
    ConvGeom g;
    if (!d || !g.init(*d)) return PRIMIA_ERR_ARG;
    if (dtype != PRIMIA_BF16 || !lh_shape(g)) return 0;
    if (use_c64(g)) return (PRIMIA_OPT(c64_bnsums) && (long)g.N * g.H * g.W * 64 < (1L << 31)) ? conv3x3_c64_grid(g.N, g.H, g.W) : 0;
    const int t = conv3x3_lh2_tiles_m(g.N, g.H, g.W, g.K, g.C);
    return t > 0 ? t : 0;
}

int primia_conv2d_dgrad_bnsums(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const void* bn_y,
                               const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                               float* sums, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && dy && w_dgrad && dx && bn_y && bn_mean && bn_invstd && bn_gamma && bn_beta && sums);
    if (primia_conv_dgrad_bnsums_slots(d, dtype) <= 0) return PRIMIA_ERR_UNSUPPORTED;
    ConvGeom g;
    PRIMIA_REQUIRE(g.init(*d));
    const LhBnBwd bnb{(const bf16*)bn_y, bn_mean, bn_invstd, bn_gamma, bn_beta};
    if (use_c64(g))
        return conv3x3_c64_dispatch((const bf16*)dy, (const bf16*)w_dgrad, (bf16*)dx, g.N, g.H, g.W, 1, 0, (hipStream_t)stream, sums,
                                    nullptr, &bnb);
    return conv3x3_lh2_dispatch((const bf16*)dy, (const bf16*)w_dgrad, (bf16*)dx, g.N, g.H, g.W, g.K, g.C, 1, 0,
                                (hipStream_t)stream, sums, nullptr, &bnb);
}

int primia_conv2d_dgrad(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                        int accumulate, int dtype, primia_stream_t stream) {
    return conv2d_dgrad_impl(d, dy, w_dgrad, dx, accumulate, nullptr, nullptr, dtype, stream);
}

int primia_conv_dgrad_masked_acc_ok(const primia_conv_desc* d, int dtype) {
    ConvGeom g;
    if (!d || !g.init(*d)) return PRIMIA_ERR_ARG;
    if (dtype != PRIMIA_BF16) return 0;
    if (use_c64(g)) return (long)g.N * g.H * g.W * 64 < (1L << 31) ? 1 : 0;
    return lh_shape(g) && conv3x3_lh2_tiles_m(g.N, g.H, g.W, g.K, g.C) > 0 ? 1 : 0;
}

int primia_conv2d_dgrad_masked_acc(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                                   const uint8_t* relu_mask, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(relu_mask);
    if (dtype != PRIMIA_BF16) return PRIMIA_ERR_UNSUPPORTED;
    return conv2d_dgrad_impl(d, dy, w_dgrad, dx, 1, nullptr, nullptr, dtype, stream, relu_mask);
}

// primia_conv2d_dgrad_masked_acc whose write-back also forms the backward sums of the BatchNorm whose OUTPUT gradient the call
// completes (dx after the call = that gradient): sum g, sum g * xhat over g = dx AS STORED where the layer's ReLU passed, as
// per-block partials [slots][2][64].  mode 2: a residual BatchNorm (aux = its input y, aux_mask = the ReLU-mask bytes of its
// forward pass, c0 = saved mean, c1 = saved invstd) -> primia_bn_bwd_mask_from_sums; mode 3: the stem's BatchNorm seen through
// the 3x3 / 2 max-pool (aux = the pooled activation p: ReLU = [p > 0], xhat = (p - beta) / gamma; c0 = beta, c1 = gamma)
// -> primia_bn_relu_maxpool_bwd_from_sums.  64 -> 64 layers (conv3x3_c64_kernel<true, 3, mode>); slots = 0 elsewhere.
int primia_conv_dgrad_masked_acc_bnsums_slots(const primia_conv_desc* d, int dtype) {
    ConvGeom g;
    if (!d || !g.init(*d)) return PRIMIA_ERR_ARG;
    if (dtype != PRIMIA_BF16 || !use_c64(g) || (long)g.N * g.H * g.W * 64 >= (1L << 31)) return 0;
    return PRIMIA_OPT(c64_bnsums) ? conv3x3_c64_grid(g.N, g.H, g.W) : 0;
}

int primia_conv2d_dgrad_masked_acc_bnsums(const primia_conv_desc* d, const void* dy, const void* w_dgrad, void* dx,
                                          const uint8_t* relu_mask, int mode, const void* aux, const uint8_t* aux_mask,
                                          const float* c0, const float* c1, float* sums, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(relu_mask && aux && c0 && c1 && sums && (mode == 2 || mode == 3) && (mode == 3 || aux_mask));
    if (primia_conv_dgrad_masked_acc_bnsums_slots(d, dtype) <= 0) return PRIMIA_ERR_UNSUPPORTED;
    const C64AccBnb ab{mode, (const bf16*)aux, aux_mask, c0, c1};
    return conv2d_dgrad_impl(d, dy, w_dgrad, dx, 1, nullptr, nullptr, dtype, stream, relu_mask, nullptr, &ab, sums);
}

// primia_conv2d_dgrad_pair whose write-back also forms the backward sums of the residual BatchNorm in FRONT of the transition
// block (dx = the gradient w.r.t. that layer's output z = relu(bn(y) + identity); its forward pass left one ReLU-mask byte per 8
// channels): sum g, sum g * xhat with g = dx AS STORED * mask bit, as partials [slots][2][C] — primia_bn_bwd_mask_from_sums
// consumes them.  Served by conv_s2lh_kernel for 64-channel dx (layer2.0) and by conv_igemm_kernel's LDS write-back loop (which
// already walks whole 16-byte chunks of the scattered pixel rows) for the wider ones; slots = 0 elsewhere.
int primia_conv_dgrad_pair_bnsums_slots(const primia_conv_desc* d, int dtype) {
    ConvGeom g;
    if (!d || !g.init(*d)) return PRIMIA_ERR_ARG;
    if (dtype != PRIMIA_BF16 || !s2_conv1_shape(g)) return 0;
    if (s2_pass_on(1, g.C)) return g.C == 64 ? 2 * conv_s2lh_tiles_m(g.N, g.H, g.W) : 0;
    // conv_igemm_kernel's parity-class walk: one partial per (128-pixel tile, class)
    if (!PRIMIA_OPT(dgrad_classes) || g.C % 8) return 0;
    return 4 * ceil_div((long)g.N * (g.H / 2) * (g.W / 2), 128);
}

int primia_conv2d_dgrad_pair_bnsums(const primia_conv_desc* d, const void* dy, const void* w_dgrad,
                                    const primia_conv_desc* d_ds, const void* dy_ds, const void* w_dgrad_ds, void* dx,
                                    const void* bn_y, const uint8_t* relu_mask, const float* bn_mean, const float* bn_invstd,
                                    float* sums, int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && d_ds && dy && w_dgrad && dy_ds && w_dgrad_ds && dx && bn_y && relu_mask && bn_mean && bn_invstd && sums);
    PRIMIA_REQUIRE(d_ds->R == 1 && d_ds->S == 1 && d_ds->stride == 2 && d_ds->pad == 0);
    PRIMIA_REQUIRE(d_ds->N == d->N && d_ds->H == d->H && d_ds->W == d->W && d_ds->C == d->C && d_ds->K == d->K);
    if (primia_conv_dgrad_pair_bnsums_slots(d, dtype) <= 0) return PRIMIA_ERR_UNSUPPORTED;
    ConvGeom g;
    PRIMIA_REQUIRE(g.init(*d));
    const S2BnBwd bnb{(const bf16*)bn_y, relu_mask, bn_mean, bn_invstd, sums};
    if (!s2_pass_on(1, g.C)) return conv2d_dgrad_impl(d, dy, w_dgrad, dx, 0, dy_ds, w_dgrad_ds, dtype, stream, nullptr, &bnb);
    return conv_s2lh_dgrad((const bf16*)dy, (const bf16*)w_dgrad, (const bf16*)dy_ds, (const bf16*)w_dgrad_ds, (bf16*)dx, g.N, g.H,
                           g.W, g.C, g.K, (hipStream_t)stream, &bnb);
}

int primia_conv2d_dgrad_pair(const primia_conv_desc* d, const void* dy, const void* w_dgrad,
                             const primia_conv_desc* d_ds, const void* dy_ds, const void* w_dgrad_ds, void* dx,
                             int dtype, primia_stream_t stream) {
    PRIMIA_REQUIRE(d && d_ds && dy_ds && w_dgrad_ds);
    // conv1 3x3 / stride 2 / pad 1 and downsample 1x1 / stride 2 / pad 0 on the same even-sized input
    PRIMIA_REQUIRE(d->R == 3 && d->S == 3 && d->stride == 2 && d->pad == 1 && d->H % 2 == 0 && d->W % 2 == 0);
    PRIMIA_REQUIRE(d_ds->R == 1 && d_ds->S == 1 && d_ds->stride == 2 && d_ds->pad == 0);
    PRIMIA_REQUIRE(d_ds->N == d->N && d_ds->H == d->H && d_ds->W == d->W && d_ds->C == d->C && d_ds->K == d->K &&
                   d_ds->Ho == d->Ho && d_ds->Wo == d->Wo);
    return conv2d_dgrad_impl(d, dy, w_dgrad, dx, 0, dy_ds, w_dgrad_ds, dtype, stream);
}

}  // extern "C"
