// Z_2^64 ring kernels for the encrypted-inference path: element-wise share arithmetic, per-share
// truncation, the PySyft im2col / pool-unroll layouts, the int64 ring GEMM and the Beaver
// (SPDZ) combine step.  Integer work: results are bit-exact by construction (unsigned wrap-around
// arithmetic on the int64 bit patterns).  Everything except the GEMM is an HBM streaming pass.
#include "common.h"

namespace primia {

typedef unsigned long long u64;

template <int OP>
__global__ __launch_bounds__(256) void ring_ew_kernel(const u64* __restrict__ a, const u64* __restrict__ b,
                                                      u64* __restrict__ out, long n, long nb) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const u64 x = a[i], y = b[nb == n ? i : i % nb];
        out[i] = OP == 0 ? x + y : (OP == 1 ? x - y : x * y);
    }
}

__global__ __launch_bounds__(256) void ring_sub_inplace_kernel(u64* __restrict__ x, const u64* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] -= y[i];
}

__global__ __launch_bounds__(256) void ring_scale_kernel(const u64* __restrict__ a, u64 k, u64* __restrict__ out,
                                                         long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = a[i] * k;
}

__global__ __launch_bounds__(256) void trunc_div_kernel(const int64_t* __restrict__ x, int64_t d,
                                                        int64_t* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        // C++ integer division truncates toward zero; do it on magnitudes so INT64_MIN is defined.
        const int64_t v = x[i];
        const u64 mag = v < 0 ? (u64)0 - (u64)v : (u64)v;
        const u64 q = mag / (u64)d;
        out[i] = v < 0 ? (int64_t)((u64)0 - q) : (int64_t)q;
    }
}

__global__ __launch_bounds__(256) void ring_rowsum_kernel(const u64* __restrict__ x, u64* __restrict__ out,
                                                          long rows, long w) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    u64 s = 0;
    for (long k = 0; k < w; ++k) s += x[r * w + k];
    out[r] = s;
}

__global__ __launch_bounds__(256) void ring_slice_cols_kernel(const u64* __restrict__ x, u64* __restrict__ out,
                                                              long rows, long w, long start, long len) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * len) return;
    const long r = i / len, j = i - r * len;
    out[i] = x[r * w + start + j];
}

// ---- layouts -------------------------------------------------------------------------------------
// im[b][ho*Wo + wo][(c*R + r)*S + s] = xpad[b][c][ho*stride + r][wo*stride + s]
__global__ __launch_bounds__(256) void im2col_syft_kernel(const u64* __restrict__ x, u64* __restrict__ im, int B,
                                                          int C, int H, int W, int R, int S, int stride,
                                                          int pad, int Ho, int Wo) {
    const long K = (long)C * R * S;
    const long total = (long)B * Ho * Wo * K;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % K);
    long t = i / K;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int s = k % S, r = (k / S) % R, c = k / (S * R);
    const int h = ho * stride - pad + r, w = wo * stride - pad + s;
    u64 v = 0;
    if (h >= 0 && h < H && w >= 0 && w < W) v = x[(((long)b * C + c) * H + h) * W + w];
    im[i] = v;
}

// out[b][o][p] = res[b][p][o] + bias[o]
__global__ __launch_bounds__(256) void col2out_syft_kernel(const u64* __restrict__ res, const u64* __restrict__ bias,
                                                           u64* __restrict__ out, int B, int P, int O) {
    const long total = (long)B * P * O;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % P);
    long t = i / P;
    const int o = (int)(t % O);
    const int b = (int)(t / O);
    out[i] = res[((long)b * P + p) * O + o] + (bias ? bias[o] : (u64)0);
}

// out[b][c][ho*Wo + wo][r*k + s] = xpad[b][c][ho*stride + r][wo*stride + s]
__global__ __launch_bounds__(256) void pool_unroll_kernel(const u64* __restrict__ x, u64* __restrict__ out, int B,
                                                          int C, int H, int W, int k, int stride, int pad,
                                                          int Ho, int Wo) {
    const long total = (long)B * C * Ho * Wo * k * k;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i % (k * k));
    long t = i / (k * k);
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    t /= Ho;  // t = b*C + c
    const int r = e / k, s = e % k;
    const int h = ho * stride - pad + r, w = wo * stride - pad + s;
    u64 v = 0;
    if (h >= 0 && h < H && w >= 0 && w < W) v = x[(t * H + h) * W + w];
    out[i] = v;
}

// ---- Beaver element-wise combine -------------------------------------------------------------------
__global__ __launch_bounds__(256) void beaver_mul_kernel(int j, const u64* __restrict__ delta,
                                                         const u64* __restrict__ eps, const u64* __restrict__ a,
                                                         const u64* __restrict__ b, const u64* __restrict__ c,
                                                         u64* __restrict__ z, long n, long nb) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const long ib = nb == n ? i : i % nb;
        const u64 d = delta[i], e = eps[ib];
        u64 v = d * b[ib] + a[i] * e + c[i];
        if (j == 0) v += d * e;
        z[i] = v;
    }
}

// ... followed by the party's truncation of ITS share (precision.py:309-316: FPT * FPT = Beaver product, then each share
// divided by the scale toward zero): the product share is not written and read back in between
__global__ __launch_bounds__(256) void beaver_mul_trunc_kernel(int j, const u64* __restrict__ delta,
                                                               const u64* __restrict__ eps, const u64* __restrict__ a,
                                                               const u64* __restrict__ b, const u64* __restrict__ c,
                                                               int64_t* __restrict__ z, long n, long nb, u64 div) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const long ib = nb == n ? i : i % nb;
        const u64 d = delta[i], e = eps[ib];
        u64 v = d * b[ib] + a[i] * e + c[i];
        if (j == 0) v += d * e;
        const int64_t sv = (int64_t)v;
        const u64 mag = sv < 0 ? (u64)0 - v : v;
        const u64 q = mag / div;
        z[i] = sv < 0 ? (int64_t)((u64)0 - q) : (int64_t)q;
    }
}

// spdz_mask of both operands in one launch: d = x - a (n elements), e = y - b (nb elements)
__global__ __launch_bounds__(256) void beaver_mask_kernel(const u64* __restrict__ x, const u64* __restrict__ a,
                                                          u64* __restrict__ d, long n, const u64* __restrict__ y,
                                                          const u64* __restrict__ b, u64* __restrict__ e, long nb) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n + nb; i += stride) {
        if (i < n)
            d[i] = x[i] - a[i];
        else
            e[i - n] = y[i - n] - b[i - n];
    }
}

// ---- FixedPrecisionTensor.reciprocal(method="newton") for two parties hosted in ONE process -------------------------
// precision.py:507-518:   x = (C + 1 - v) / C;  79 x { y = C + 1 - v * (x * x);  x = y * x / C },  C = 20,
// every product a Beaver multiplication followed by each party's truncation of ITS share, every `C + 1 - t` the
// re-sharing of the public constant with a fresh mask (additive_shared.py:453-527).  Unfused that is ~38 launches per
// iteration on a few thousand elements (half of all launches of an encrypted forward); the iteration is element-wise,
// and with both parties' shares on this GPU an "open" is an addition, so one thread can carry one element of BOTH
// parties through all 80 steps.  The arithmetic, and the order in which the crypto provider's primitives are consumed,
// are exactly those of the unfused chain (tests: bit-identical); the three-role deployment keeps the unfused form,
// there the opens are messages.
//   prim: device array of pointers, in consumption order:  mask0 | 79 x { T1 (a0,b0,c0,a1,b1,c1), T2 (..6..), mask, T3 (..6..) }
__device__ __forceinline__ int64_t trunc_div1(int64_t v, u64 d) {
    const u64 mag = v < 0 ? (u64)0 - (u64)v : (u64)v;
    const u64 q = mag / d;
    return v < 0 ? (int64_t)((u64)0 - q) : (int64_t)q;
}

__global__ __launch_bounds__(64) void newton_local_kernel(const u64* __restrict__ v0, const u64* __restrict__ v1,
                                                          const u64* const* __restrict__ prim, u64 scale,
                                                          u64* __restrict__ x0o, u64* __restrict__ x1o, long n) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    constexpr u64 C = 20;
    const u64 c21 = 21 * scale;                    // (C + 1) encoded
    const u64 va = v0[i], vb = v1[i];
    // Beaver product of (xa, xb) and (ya, yb) with triple t (6 pointers), then each party truncates by `scale`
    auto fpt_mul = [&](u64 xa, u64 xb, u64 ya, u64 yb, const u64* const* t, u64& za, u64& zb) {
        const u64 a0 = t[0][i], b0 = t[1][i], c0 = t[2][i], a1 = t[3][i], b1 = t[4][i], c1 = t[5][i];
        const u64 delta = (xa - a0) + (xb - a1), eps = (ya - b0) + (yb - b1);      // spdz_mask + open
        const u64 w0 = delta * b0 + a0 * eps + c0 + delta * eps;                   // spdz_compute, j = 0
        const u64 w1 = delta * b1 + a1 * eps + c1;                                 // j = 1
        za = (u64)trunc_div1((int64_t)w0, scale);
        zb = (u64)trunc_div1((int64_t)w1, scale);
    };
    // -(t - [21*scale]) with the constant re-shared as (r, 21*scale - r)
    auto c21_minus = [&](u64 ta, u64 tb, const u64* mask, u64& ya, u64& yb) {
        const u64 r = mask[0];
        ya = (u64)0 - (ta - r);
        yb = (u64)0 - (tb - (c21 - r));
    };
    u64 ya, yb, xa, xb;
    c21_minus(va, vb, prim[0], ya, yb);
    xa = (u64)trunc_div1((int64_t)ya, C);
    xb = (u64)trunc_div1((int64_t)yb, C);
    const u64* const* pp = prim + 1;
#pragma unroll 1
    for (int it = 0; it < 79; ++it, pp += 19) {
        u64 qa, qb, ta, tb;
        fpt_mul(xa, xb, xa, xb, pp, qa, qb);            // x * x
        fpt_mul(va, vb, qa, qb, pp + 6, ta, tb);        // v * (x * x)
        c21_minus(ta, tb, pp[12], ya, yb);              // C + 1 - ...
        fpt_mul(ya, yb, xa, xb, pp + 13, qa, qb);       // y * x
        xa = (u64)trunc_div1((int64_t)qa, C);           // / C
        xb = (u64)trunc_div1((int64_t)qb, C);
    }
    x0o[i] = xa;
    x1o[i] = xb;
}

// ---- int64 ring GEMM: C = C0 + A1@B1 + A2@B2 ---------------------------------------------------------
// 64x64 output tile per 256-thread block, 4x4 outputs per thread, k-step 16 staged through LDS.
// There is no 64-bit integer MFMA; the products run on the vector ALU (v_mad_u64_u32 chains).
struct GemmPair {
    const u64* A;
    const u64* B;
};

__global__ __launch_bounds__(256) void ring_gemm_kernel(GemmPair p1, GemmPair p2, const u64* __restrict__ C0,
                                                        u64* __restrict__ C, int M, int K, int N, int ksplit) {
    __shared__ u64 sa[16][64 + 1];
    __shared__ u64 sb[16][64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    // split-K: slice z covers k in [kb, ke); slices are combined with 64-bit atomics (ring addition is
    // associative, so the result is bit-identical whatever the order)
    const int kchunk = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
    const int kb = blockIdx.z * kchunk;
    const int ke = kb + kchunk < K ? kb + kchunk : K;
    u64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    for (int pair = 0; pair < 2; ++pair) {
        const u64* __restrict__ A = pair == 0 ? p1.A : p2.A;
        const u64* __restrict__ Bm = pair == 0 ? p1.B : p2.B;
        if (!A) continue;
        for (int k0 = kb; k0 < ke; k0 += 16) {
            // A tile: 64 rows x 16 k (transposed into sa[k][m]); B tile: 16 k x 64 cols
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int idx = threadIdx.x + 256 * l;
                const int am = idx >> 4, ak = idx & 15;
                const int gm = m0 + am, gk = k0 + ak;
                sa[ak][am] = (gm < M && gk < ke) ? A[(long)gm * K + gk] : (u64)0;
                const int bk = idx >> 6, bn = idx & 63;
                const int gn = n0 + bn, gk2 = k0 + bk;
                sb[bk][bn] = (gn < N && gk2 < ke) ? Bm[(long)gk2 * N + gn] : (u64)0;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                u64 av[4], bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = sa[kk][ty * 4 + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = sb[kk][tx + 16 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx + 16 * j;
            if (gn >= N) continue;
            const long o = (long)gm * N + gn;
            if (ksplit == 1)
                C[o] = acc[i][j] + (C0 ? C0[o] : (u64)0);
            else
                atomicAdd(C + o, acc[i][j]);  // C was initialised with C0 (or zero) by the launcher
        }
    }
}

// The crypto provider's side of a Beaver triple whose input shares were drawn directly (a = a0 + a1, b = b0 + b1, all four
// uniform; c0 uniform): the second share of the product, c1 = a * b - c0 (mpc/beaver.py:7-63 splits c the same way).
__global__ __launch_bounds__(256) void triple_mul_c1_kernel(const u64* __restrict__ x0, const u64* __restrict__ x1,
                                                            const u64* __restrict__ y0, const u64* __restrict__ y1,
                                                            const u64* __restrict__ c0, u64* __restrict__ c1, long n, long nb) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const long j = nb == n ? i : i % nb;
        c1[i] = (x0[i] + x1[i]) * (y0[j] + y1[j]) - c0[i];
    }
}

static inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

static int launch_gemm(GemmPair p1, GemmPair p2, const u64* C0, u64* C, int M, int K, int N, hipStream_t st) {
    const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
    // few output tiles (the N = 1 image has M = 49..784 rows in layer3/4): split K to fill the chip
    int ksplit = 1;
    if (tiles < 256) {
        ksplit = (512 + tiles - 1) / tiles;
        const int kmax = (K + 63) / 64;  // >= 64 k per slice
        if (ksplit > kmax) ksplit = kmax;
        if (ksplit < 1) ksplit = 1;
    }
    if (ksplit > 1) {
        const size_t bytes = (size_t)M * N * sizeof(u64);
        hipError_t e = hipSuccess;
        if (!C0)
            e = hipMemsetAsync(C, 0, bytes, st);
        else if (C0 != C)
            e = hipMemcpyAsync(C, C0, bytes, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return PRIMIA_ERR_LAUNCH;
    }
    dim3 grid((N + 63) / 64, (M + 63) / 64, ksplit);
    ring_gemm_kernel<<<grid, 256, 0, st>>>(p1, p2, C0, C, M, K, N, ksplit);
    return launch_status();
}

}  // namespace primia

using namespace primia;

extern "C" {

#define RING_EW(NAME, OP)                                                                                   \
    int NAME(const int64_t* a, const int64_t* b, int64_t* out, int64_t n, int64_t nb, primia_stream_t st) { \
        if (n == 0) return PRIMIA_OK; /* empty input: no-op */                                             \
        PRIMIA_REQUIRE(a && b && out && n >= 0 && nb > 0 && nb <= n && (n == 0 || n % nb == 0));           \
        if (n == 0) return PRIMIA_OK;                                                                       \
        ring_ew_kernel<OP><<<ew_blocks(n), 256, 0, (hipStream_t)st>>>((const u64*)a, (const u64*)b,         \
                                                                      (u64*)out, n, nb);                    \
        return launch_status();                                                                             \
    }
RING_EW(primia_ring_add, 0)
RING_EW(primia_ring_sub, 1)
RING_EW(primia_ring_mul, 2)

int primia_ring_scale(const int64_t* a, int64_t k, int64_t* out, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(a && out && n >= 0);
    if (n == 0) return PRIMIA_OK;
    ring_scale_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)st>>>((const u64*)a, (u64)k, (u64*)out, n);
    return launch_status();
}

int primia_trunc_div(const int64_t* x, int64_t d, int64_t* out, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(x && out && n >= 0 && d > 0);
    if (n == 0) return PRIMIA_OK;
    trunc_div_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)st>>>(x, d, out, n);
    return launch_status();
}

int primia_ring_rowsum(const int64_t* x, int64_t* out, int64_t rows, int64_t w, primia_stream_t st) {
    PRIMIA_REQUIRE(x && out && rows > 0 && w > 0);
    ring_rowsum_kernel<<<ceil_div(rows, 256), 256, 0, (hipStream_t)st>>>((const u64*)x, (u64*)out, rows, w);
    return launch_status();
}

int primia_ring_slice_cols(const int64_t* x, int64_t* out, int64_t rows, int64_t w, int64_t start, int64_t len,
                           primia_stream_t st) {
    PRIMIA_REQUIRE(x && out && rows > 0 && w > 0 && start >= 0 && len > 0 && start + len <= w);
    ring_slice_cols_kernel<<<ceil_div(rows * len, 256), 256, 0, (hipStream_t)st>>>((const u64*)x, (u64*)out, rows, w,
                                                                                   start, len);
    return launch_status();
}

int primia_ring_matmul(const int64_t* A, const int64_t* B, int64_t* C, int M, int K, int N, int accumulate,
                       primia_stream_t st) {
    PRIMIA_REQUIRE(A && B && C && M > 0 && K > 0 && N > 0);
    return launch_gemm(GemmPair{(const u64*)A, (const u64*)B}, GemmPair{nullptr, nullptr},
                       accumulate ? (const u64*)C : nullptr, (u64*)C, M, K, N, (hipStream_t)st);
}

int primia_im2col_syft(const int64_t* x, int64_t* im, int B, int C, int H, int W, int R, int S, int stride,
                       int pad, primia_stream_t st) {
    PRIMIA_REQUIRE(x && im && B > 0 && C > 0 && H > 0 && W > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0);
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    PRIMIA_REQUIRE(Ho > 0 && Wo > 0);
    const long total = (long)B * Ho * Wo * C * R * S;
    im2col_syft_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)st>>>((const u64*)x, (u64*)im, B, C, H, W, R, S,
                                                                           stride, pad, Ho, Wo);
    return launch_status();
}

int primia_col2out_syft(const int64_t* res, const int64_t* bias, int64_t* out, int B, int HoWo, int O,
                        primia_stream_t st) {
    PRIMIA_REQUIRE(res && out && B > 0 && HoWo > 0 && O > 0);
    const long total = (long)B * HoWo * O;
    col2out_syft_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)st>>>((const u64*)res, (const u64*)bias,
                                                                            (u64*)out, B, HoWo, O);
    return launch_status();
}

int primia_pool_unroll_syft(const int64_t* x, int64_t* out, int B, int C, int H, int W, int k, int stride,
                            int pad, primia_stream_t st) {
    PRIMIA_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0);
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    PRIMIA_REQUIRE(Ho > 0 && Wo > 0);
    const long total = (long)B * C * Ho * Wo * k * k;
    pool_unroll_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)st>>>((const u64*)x, (u64*)out, B, C, H, W, k,
                                                                           stride, pad, Ho, Wo);
    return launch_status();
}

int primia_newton_reciprocal_local(const int64_t* v0, const int64_t* v1, const int64_t* const* prim, int64_t scale,
                                   int64_t* x0, int64_t* x1, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(v0 && v1 && prim && x0 && x1 && n > 0 && scale > 0);
    newton_local_kernel<<<ceil_div(n, 64), 64, 0, (hipStream_t)st>>>((const u64*)v0, (const u64*)v1,
                                                                      (const u64* const*)prim, (u64)scale, (u64*)x0,
                                                                      (u64*)x1, n);
    return launch_status();
}

int primia_beaver_mask(const int64_t* x, const int64_t* a, int64_t* d, int64_t n, const int64_t* y, const int64_t* b,
                       int64_t* e, int64_t nb, primia_stream_t st) {
    if (n + nb == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(n >= 0 && nb >= 0 && (n == 0 || (x && a && d)) && (nb == 0 || (y && b && e)));
    beaver_mask_kernel<<<ew_blocks(n + nb), 256, 0, (hipStream_t)st>>>((const u64*)x, (const u64*)a, (u64*)d, n,
                                                                       (const u64*)y, (const u64*)b, (u64*)e, nb);
    return launch_status();
}

int primia_beaver_combine_mul_trunc(int j, const int64_t* delta, const int64_t* eps, const int64_t* a, const int64_t* b,
                                    const int64_t* c, int64_t* z, int64_t n, int64_t nb, int64_t div,
                                    primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE((j == 0 || j == 1) && delta && eps && a && b && c && z && n >= 0 && nb > 0 && nb <= n && div > 0);
    PRIMIA_REQUIRE(n % nb == 0);
    beaver_mul_trunc_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)st>>>(j, (const u64*)delta, (const u64*)eps, (const u64*)a,
                                                                        (const u64*)b, (const u64*)c, z, n, nb, (u64)div);
    return launch_status();
}

int primia_beaver_combine_mul(int j, const int64_t* delta, const int64_t* eps, const int64_t* a,
                              const int64_t* b, const int64_t* c, int64_t* z, int64_t n, int64_t nb,
                              primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE((j == 0 || j == 1) && delta && eps && a && b && c && z && n >= 0 && nb > 0 && nb <= n);
    PRIMIA_REQUIRE(n == 0 || n % nb == 0);
    if (n == 0) return PRIMIA_OK;
    beaver_mul_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)st>>>(j, (const u64*)delta, (const u64*)eps,
                                                                  (const u64*)a, (const u64*)b, (const u64*)c,
                                                                  (u64*)z, n, nb);
    return launch_status();
}

int primia_beaver_combine_matmul(int j, const int64_t* delta, const int64_t* eps, const int64_t* a,
                                 const int64_t* b, const int64_t* c, int64_t* z, int64_t* scratch, int M, int K,
                                 int N, primia_stream_t st) {
    PRIMIA_REQUIRE((j == 0 || j == 1) && delta && eps && a && b && c && z && M > 0 && K > 0 && N > 0);
    hipStream_t s = (hipStream_t)st;
    // z = c + delta @ (b [+ eps]) + a @ eps  — ring-identical to the reference's three products
    const u64* b2 = (const u64*)b;
    if (j == 0) {
        PRIMIA_REQUIRE(scratch);
        const long kn = (long)K * N;
        ring_ew_kernel<0><<<ew_blocks(kn), 256, 0, s>>>((const u64*)b, (const u64*)eps, (u64*)scratch, kn, kn);
        b2 = (const u64*)scratch;
    }
    return launch_gemm(GemmPair{(const u64*)delta, b2}, GemmPair{(const u64*)a, (const u64*)eps}, (const u64*)c,
                       (u64*)z, M, K, N, s);
}

int primia_triple_mul_c1(const int64_t* x0, const int64_t* x1, const int64_t* y0, const int64_t* y1, const int64_t* c0,
                         int64_t* c1, int64_t n, int64_t nb, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(x0 && x1 && y0 && y1 && c0 && c1 && n > 0 && nb > 0 && nb <= n && n % nb == 0);
    triple_mul_c1_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)st>>>((const u64*)x0, (const u64*)x1, (const u64*)y0,
                                                                     (const u64*)y1, (const u64*)c0, (u64*)c1, n, nb);
    return launch_status();
}

int primia_triple_matmul_c1(const int64_t* a0, const int64_t* a1, const int64_t* b0, const int64_t* b1, const int64_t* c0,
                            int64_t* c1, int64_t* scratch, int M, int K, int N, primia_stream_t st) {
    PRIMIA_REQUIRE(a0 && a1 && b0 && b1 && c0 && c1 && scratch && M > 0 && K > 0 && N > 0);
    hipStream_t s = (hipStream_t)st;
    const long kn = (long)K * N, mn = (long)M * N, mk = (long)M * K;
    // a = a0 + a1, b = b0 + b1 (scratch: [M K | K N]); c1 = a @ b; c1 -= c0
    u64* sa = (u64*)scratch;
    u64* sb = sa + mk;
    ring_ew_kernel<0><<<ew_blocks(mk), 256, 0, s>>>((const u64*)a0, (const u64*)a1, sa, mk, mk);
    ring_ew_kernel<0><<<ew_blocks(kn), 256, 0, s>>>((const u64*)b0, (const u64*)b1, sb, kn, kn);
    const int rc = launch_gemm(GemmPair{sa, sb}, GemmPair{nullptr, nullptr}, nullptr, (u64*)c1, M, K, N, s);
    if (rc != PRIMIA_OK) return rc;
    ring_sub_inplace_kernel<<<ew_blocks(mn), 256, 0, s>>>((u64*)c1, (const u64*)c0, mn);
    return launch_status();
}

}  // extern "C"
