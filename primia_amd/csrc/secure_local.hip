// Encrypted inference with BOTH parties' shares on this GPU (the in-process deployment that inference.py's
// VirtualWorker run and tools/bench_secure.py time): per-layer kernels that carry the shares of party 0 AND party 1
// through a whole protocol step.
//
// In this deployment an "open" (mpc/spdz.py:162-176: each party sends its masked value, both add) is an addition of two
// values that already sit side by side, so the chain  mask -> open -> combine -> truncate -> re-layout  of a layer — six
// to twenty-odd launches of a few microseconds each on tensors of a few hundred kilobytes — is one pass in which a
// thread computes exactly what each party computes, in the party's own arithmetic (unsigned wrap-around on the int64
// patterns, truncation toward zero per share).  The crypto provider's primitives are requested by the host in the
// reference's order and handed in by pointer: results are bit-identical to the step-by-step chain of csrc/ring.hip
// (tests/test_gpu_secure_local.py) and to the reference-minted fixtures (tests/test_gpu_secure_ref.py).  The three-role
// deployment (one rank per party, opens are messages) keeps the step-by-step form.
//
// Reference: syft/frameworks/torch/mpc/spdz.py:125-197 (spdz_mul), tensors/interpreters/precision.py:309-316,419-463
// (FPT mul / matmul + truncation), nn/functional.py:44-75 (batch_norm), :204-308 (conv2d), :460-508 (_pool2d).
#include "common.h"

namespace primia {

typedef unsigned long long u64;

__device__ __forceinline__ u64 sl_trunc(u64 v, u64 d) {        // a party's truncation of ITS share toward zero
    const int64_t sv = (int64_t)v;
    const u64 mag = sv < 0 ? (u64)0 - v : v;
    const u64 q = mag / d;
    return sv < 0 ? (u64)0 - q : q;
}

struct Pair {
    const u64* p0;
    const u64* p1;
};
struct OutPair {
    u64* p0;
    u64* p1;
};
struct Triple {          // a pairs with the first operand, b with the second, c = a * b (shares of both parties)
    const u64 *a0, *b0, *c0, *a1, *b1, *c1;
};

// Beaver product of (x0, x1) and (y0, y1): delta = open(x - a), eps = open(y - b); z_j = delta b_j + a_j eps + c_j
// (+ delta eps for j = 0)
__device__ __forceinline__ void sl_beaver(u64 x0, u64 x1, u64 y0, u64 y1, u64 a0, u64 b0, u64 c0, u64 a1, u64 b1, u64 c1,
                                          u64& z0, u64& z1) {
    const u64 delta = (x0 - a0) + (x1 - a1), eps = (y0 - b0) + (y1 - b1);
    z0 = delta * b0 + a0 * eps + c0 + delta * eps;
    z1 = delta * b1 + a1 * eps + c1;
}

// ---- element-wise on both parties' shares: out_j = a_j (op) b_j[i % nb] --------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void ew2p_kernel(Pair a, Pair b, OutPair o, long n, long nb) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += stride) {
        const int j = i >= n;
        const long e = j ? i - n : i;
        const u64 x = (j ? a.p1 : a.p0)[e], y = (j ? b.p1 : b.p0)[nb == n ? e : e % nb];
        (j ? o.p1 : o.p0)[e] = OP == 0 ? x + y : x - y;
    }
}

// ---- FPT * FPT for both parties (spdz_mul "mul" + per-share truncation); x: n elements, y: nb (broadcast) ----------
__global__ __launch_bounds__(256) void fpt_mul_local_kernel(Pair x, Pair y, Triple t, Pair addend, OutPair z, long n,
                                                            long nb, u64 div) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const long ib = nb == n ? i : i % nb;
        u64 z0, z1;
        sl_beaver(x.p0[i], x.p1[i], y.p0[ib], y.p1[ib], t.a0[i], t.b0[ib], t.c0[i], t.a1[i], t.b1[ib], t.c1[i], z0, z1);
        if (div) {
            z0 = sl_trunc(z0, div);
            z1 = sl_trunc(z1, div);
        }
        if (addend.p0) {
            z0 += addend.p0[i];
            z1 += addend.p1[i];
        }
        z.p0[i] = z0;
        z.p1[i] = z1;
    }
}

// ---- left + (right >= left) * (right - left)  (nn/functional.py:494) ---------------------------------------------------
// bit: shares of the comparison [rows * len]; left / right: column ranges [start, start + len) of [rows][w] matrices
struct ColOperand {
    const u64 *p0, *p1;
    int w, start;
};
__global__ __launch_bounds__(256) void max_combine_local_kernel(Pair bit, ColOperand left, ColOperand right, Triple t,
                                                                OutPair out, long rows, int len) {
    const long n = rows * len;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const long r = i / len, c = i - r * len;
        const long il = r * left.w + left.start + c, ir = r * right.w + right.start + c;
        const u64 l0 = left.p0[il], l1 = left.p1[il];
        const u64 d0 = right.p0[ir] - l0, d1 = right.p1[ir] - l1;
        u64 z0, z1;
        sl_beaver(bit.p0[i], bit.p1[i], d0, d1, t.a0[i], t.b0[i], t.c0[i], t.a1[i], t.b1[i], t.c1[i], z0, z1);
        out.p0[i] = l0 + z0;
        out.p1[i] = l1 + z1;
    }
}

// ---- batch_norm in eval mode (nn/functional.py:44-75) on one image, both parties -------------------------------------
//   rows = x.permute(1,0,2,3).reshape(C,-1).t()                         [HW, C]
//   normalized = inv * (rows - mean)     (FPT mul: Beaver + truncation; triple t1: a ~ inv [C], b ~ rows, c ~ rows)
//   result = normalized * weight + bias  (triple t2: a ~ rows, b ~ weight [C], c ~ rows)
//   out = result.t().reshape(C, 1, H, W).permute(1,0,2,3)               [1, C, H, W]
// x / out are NCHW ([C][HW]), the triples are in the ROWS layout: a 32 x 32 tile goes through LDS so that both sides
// are read and written in whole lines.
struct BnVec {
    const u64 *mean0, *mean1, *inv0, *inv1, *w0, *w1, *bias0, *bias1;
};
__global__ __launch_bounds__(256) void bn_eval_local_kernel(Pair x, BnVec v, Triple t1, Triple t2, OutPair out, int C, int HW,
                                                            u64 div) {
    __shared__ u64 tile[2][32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    // load x[c][p] (coalesced along p)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, p = p0 + tx;
        if (c < C && p < HW) {
            tile[0][ty + 8 * k][tx] = x.p0[(long)c * HW + p];
            tile[1][ty + 8 * k][tx] = x.p1[(long)c * HW + p];
        }
    }
    __syncthreads();
    // rows layout: thread (p, c) with c fastest
    u64 r0[4], r1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + ty + 8 * k, c = c0 + tx;
        r0[k] = r1[k] = 0;
        if (c < C && p < HW) {
            const long i = (long)p * C + c;
            const u64 y0 = tile[0][tx][ty + 8 * k] - v.mean0[c], y1 = tile[1][tx][ty + 8 * k] - v.mean1[c];
            u64 n0, n1;
            // fpt_mul(inv, rows - mean): the small operand (inv) is the FIRST one -> its triple side is `a`
            {
                const u64 delta = (v.inv0[c] - t1.a0[c]) + (v.inv1[c] - t1.a1[c]);
                const u64 eps = (y0 - t1.b0[i]) + (y1 - t1.b1[i]);
                n0 = sl_trunc(delta * t1.b0[i] + t1.a0[c] * eps + t1.c0[i] + delta * eps, div);
                n1 = sl_trunc(delta * t1.b1[i] + t1.a1[c] * eps + t1.c1[i], div);
            }
            u64 z0, z1;
            sl_beaver(n0, n1, v.w0[c], v.w1[c], t2.a0[i], t2.b0[c], t2.c0[i], t2.a1[i], t2.b1[c], t2.c1[i], z0, z1);
            r0[k] = sl_trunc(z0, div) + v.bias0[c];
            r1[k] = sl_trunc(z1, div) + v.bias1[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        tile[0][tx][ty + 8 * k] = r0[k];
        tile[1][tx][ty + 8 * k] = r1[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, p = p0 + tx;
        if (c < C && p < HW) {
            out.p0[(long)c * HW + p] = tile[0][ty + 8 * k][tx];
            out.p1[(long)c * HW + p] = tile[1][ty + 8 * k][tx];
        }
    }
}

// ---- conv2d / linear pieces ------------------------------------------------------------------------------------------
// im2col of both parties' shares (the layout of im2col_syft_kernel, ring.hip)
__global__ __launch_bounds__(256) void im2col_2p_kernel(Pair x, OutPair im, int B, int C, int H, int W, int R, int S, int stride,
                                                        int pad, int Ho, int Wo) {
    const long K = (long)C * R * S;
    const long total = (long)B * Ho * Wo * K;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % K);
    long tt = i / K;
    const int wo = (int)(tt % Wo);
    tt /= Wo;
    const int ho = (int)(tt % Ho);
    const int b = (int)(tt / Ho);
    const int s = k % S, r = (k / S) % R, c = k / (S * R);
    const int h = ho * stride - pad + r, w = wo * stride - pad + s;
    u64 v0 = 0, v1 = 0;
    if (h >= 0 && h < H && w >= 0 && w < W) {
        const long src = (((long)b * C + c) * H + h) * W + w;
        v0 = x.p0[src];
        v1 = x.p1[src];
    }
    im.p0[i] = v0;
    im.p1[i] = v1;
}

// spdz_mask + open of a matrix product's operands, and the accumulators' initial values, in one pass:
//   delta = (x0 - a0) + (x1 - a1)  [M K]     eps = (y0 - b0) + (y1 - b1)  [K N]     b0e = b0 + eps  [K N]
//   z0 = c0, z1 = c1  [M N]   (the ring GEMMs then add  delta @ b0e + a0 @ eps  and  delta @ b1 + a1 @ eps)
__global__ __launch_bounds__(256) void matmul_open_local_kernel(Pair x, Pair y, Triple t, u64* __restrict__ delta,
                                                                u64* __restrict__ eps, u64* __restrict__ b0e, OutPair z,
                                                                long mk, long kn, long mn) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < mk + kn + mn; i += stride) {
        if (i < mk) {
            delta[i] = (x.p0[i] - t.a0[i]) + (x.p1[i] - t.a1[i]);
        } else if (i < mk + kn) {
            const long e = i - mk;
            const u64 ev = (y.p0[e] - t.b0[e]) + (y.p1[e] - t.b1[e]);
            eps[e] = ev;
            b0e[e] = t.b0[e] + ev;
        } else {
            const long e = i - mk - kn;
            z.p0[e] = t.c0[e];
            z.p1[e] = t.c1[e];
        }
    }
}

// both parties' products in one grid: C_j += A1 @ B1_j + A2_j @ B2 (ring_gemm_kernel's tiling; gridDim.z = 2 * ksplit)
__global__ __launch_bounds__(256) void ring_gemm2_kernel(const u64* __restrict__ delta, const u64* __restrict__ eps,
                                                         const u64* __restrict__ b0e, const u64* __restrict__ b1,
                                                         const u64* __restrict__ a0, const u64* __restrict__ a1, OutPair z, int M,
                                                         int K, int N, int ksplit) {
    __shared__ u64 sa[16][64 + 1];
    __shared__ u64 sb[16][64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int party = blockIdx.z / ksplit, slice = blockIdx.z - party * ksplit;
    const int kchunk = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
    const int kb = slice * kchunk;
    const int ke = kb + kchunk < K ? kb + kchunk : K;
    u64* __restrict__ Cm = party ? z.p1 : z.p0;
    u64 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    for (int pair = 0; pair < 2; ++pair) {
        const u64* __restrict__ A = pair == 0 ? delta : (party ? a1 : a0);
        const u64* __restrict__ Bm = pair == 0 ? (party ? b1 : b0e) : eps;
        for (int k0 = kb; k0 < ke; k0 += 16) {
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
                Cm[o] += acc[i][j];               // (this thread owns the element: c_j is already there)
            else
                atomicAdd(Cm + o, acc[i][j]);     // ring addition is associative: any order gives the same bits
        }
    }
}

// each party's truncation of its product share, then out[b][o][p] = res[b][p][o] (+ bias[o])
__global__ __launch_bounds__(256) void trunc_col2out_2p_kernel(Pair res, Pair bias, OutPair out, int B, int P, int O, u64 div) {
    const long total = (long)B * P * O;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % P);
    long t = i / P;
    const int o = (int)(t % O);
    const int b = (int)(t / O);
    const long src = ((long)b * P + p) * O + o;
    out.p0[i] = sl_trunc(res.p0[src], div) + (bias.p0 ? bias.p0[o] : (u64)0);
    out.p1[i] = sl_trunc(res.p1[src], div) + (bias.p1 ? bias.p1[o] : (u64)0);
}

// pool unroll of both parties (pool_unroll_kernel, ring.hip)
__global__ __launch_bounds__(256) void pool_unroll_2p_kernel(Pair x, OutPair out, int B, int C, int H, int W, int k, int stride,
                                                             int pad, int Ho, int Wo) {
    const long total = (long)B * C * Ho * Wo * k * k;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i % (k * k));
    long t = i / (k * k);
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    t /= Ho;
    const int r = e / k, s = e % k;
    const int h = ho * stride - pad + r, w = wo * stride - pad + s;
    u64 v0 = 0, v1 = 0;
    if (h >= 0 && h < H && w >= 0 && w < W) {
        v0 = x.p0[(t * H + h) * W + w];
        v1 = x.p1[(t * H + h) * W + w];
    }
    out.p0[i] = v0;
    out.p1[i] = v1;
}

static inline int sl_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace primia

using namespace primia;

#define U(p) ((const u64*)(p))

extern "C" {

int primia_ring_ew_2p(int op, const int64_t* a0, const int64_t* a1, const int64_t* b0, const int64_t* b1, int64_t* o0,
                      int64_t* o1, int64_t n, int64_t nb, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE((op == 0 || op == 1) && a0 && a1 && b0 && b1 && o0 && o1 && n > 0 && nb > 0 && nb <= n && n % nb == 0);
    const Pair a{U(a0), U(a1)}, b{U(b0), U(b1)};
    const OutPair o{(u64*)o0, (u64*)o1};
    if (op == 0)
        ew2p_kernel<0><<<sl_blocks(2 * n), 256, 0, (hipStream_t)st>>>(a, b, o, n, nb);
    else
        ew2p_kernel<1><<<sl_blocks(2 * n), 256, 0, (hipStream_t)st>>>(a, b, o, n, nb);
    return launch_status();
}

int primia_fpt_mul_local(const int64_t* x0, const int64_t* x1, const int64_t* y0, const int64_t* y1, const int64_t* a0,
                         const int64_t* b0, const int64_t* c0, const int64_t* a1, const int64_t* b1, const int64_t* c1,
                         const int64_t* add0, const int64_t* add1, int64_t* z0, int64_t* z1, int64_t n, int64_t nb,
                         int64_t div, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(x0 && x1 && y0 && y1 && a0 && b0 && c0 && a1 && b1 && c1 && z0 && z1 && n > 0 && nb > 0 && nb <= n &&
                   n % nb == 0 && div >= 0 && ((add0 == nullptr) == (add1 == nullptr)));
    fpt_mul_local_kernel<<<sl_blocks(n), 256, 0, (hipStream_t)st>>>(
        Pair{U(x0), U(x1)}, Pair{U(y0), U(y1)}, Triple{U(a0), U(b0), U(c0), U(a1), U(b1), U(c1)}, Pair{U(add0), U(add1)},
        OutPair{(u64*)z0, (u64*)z1}, n, nb, (u64)div);
    return launch_status();
}

int primia_max_combine_local(const int64_t* bit0, const int64_t* bit1, const int64_t* left0, const int64_t* left1, int wl,
                             int start_left, const int64_t* right0, const int64_t* right1, int wr, int start_right,
                             const int64_t* a0, const int64_t* b0, const int64_t* c0, const int64_t* a1, const int64_t* b1,
                             const int64_t* c1, int64_t* out0, int64_t* out1, int64_t rows, int len, primia_stream_t st) {
    PRIMIA_REQUIRE(bit0 && bit1 && left0 && left1 && right0 && right1 && a0 && b0 && c0 && a1 && b1 && c1 && out0 && out1 &&
                   rows > 0 && len > 0 && start_left >= 0 && start_right >= 0 && start_left + len <= wl &&
                   start_right + len <= wr);
    max_combine_local_kernel<<<sl_blocks(rows * len), 256, 0, (hipStream_t)st>>>(
        Pair{U(bit0), U(bit1)}, ColOperand{U(left0), U(left1), wl, start_left}, ColOperand{U(right0), U(right1), wr, start_right},
        Triple{U(a0), U(b0), U(c0), U(a1), U(b1), U(c1)}, OutPair{(u64*)out0, (u64*)out1}, rows, len);
    return launch_status();
}

int primia_bn_eval_local(const int64_t* x0, const int64_t* x1, const int64_t* mean0, const int64_t* mean1,
                         const int64_t* inv0, const int64_t* inv1, const int64_t* w0, const int64_t* w1,
                         const int64_t* bias0, const int64_t* bias1, const int64_t* const* t1, const int64_t* const* t2,
                         int64_t* out0, int64_t* out1, int C, int HW, int64_t div, primia_stream_t st) {
    PRIMIA_REQUIRE(x0 && x1 && mean0 && mean1 && inv0 && inv1 && w0 && w1 && bias0 && bias1 && t1 && t2 && out0 && out1 &&
                   C > 0 && HW > 0 && div > 0);
    for (int k = 0; k < 6; ++k) PRIMIA_REQUIRE(t1[k] && t2[k]);
    const dim3 grid((HW + 31) / 32, (C + 31) / 32);
    bn_eval_local_kernel<<<grid, 256, 0, (hipStream_t)st>>>(
        Pair{U(x0), U(x1)}, BnVec{U(mean0), U(mean1), U(inv0), U(inv1), U(w0), U(w1), U(bias0), U(bias1)},
        Triple{U(t1[0]), U(t1[1]), U(t1[2]), U(t1[3]), U(t1[4]), U(t1[5])},
        Triple{U(t2[0]), U(t2[1]), U(t2[2]), U(t2[3]), U(t2[4]), U(t2[5])}, OutPair{(u64*)out0, (u64*)out1}, C, HW, (u64)div);
    return launch_status();
}

int primia_im2col_syft_2p(const int64_t* x0, const int64_t* x1, int64_t* im0, int64_t* im1, int B, int C, int H, int W, int R,
                          int S, int stride, int pad, primia_stream_t st) {
    PRIMIA_REQUIRE(x0 && x1 && im0 && im1 && B > 0 && C > 0 && H > 0 && W > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0);
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    PRIMIA_REQUIRE(Ho > 0 && Wo > 0);
    const long total = (long)B * Ho * Wo * C * R * S;
    im2col_2p_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)st>>>(Pair{U(x0), U(x1)}, OutPair{(u64*)im0, (u64*)im1}, B, C,
                                                                         H, W, R, S, stride, pad, Ho, Wo);
    return launch_status();
}

int64_t primia_beaver_matmul_local_scratch_elems(int M, int K, int N) {
    return M > 0 && K > 0 && N > 0 ? (int64_t)M * K + 2 * (int64_t)K * N : PRIMIA_ERR_ARG;
}

// z_j [M N] = the parties' shares of x @ y (spdz_mul "matmul", mpc/spdz.py:125-197); scratch: int64 [M K + 2 K N]
int primia_beaver_matmul_local(const int64_t* x0, const int64_t* x1, const int64_t* y0, const int64_t* y1, const int64_t* a0,
                               const int64_t* b0, const int64_t* c0, const int64_t* a1, const int64_t* b1, const int64_t* c1,
                               int64_t* z0, int64_t* z1, int64_t* scratch, int M, int K, int N, primia_stream_t st) {
    PRIMIA_REQUIRE(x0 && x1 && y0 && y1 && a0 && b0 && c0 && a1 && b1 && c1 && z0 && z1 && scratch && M > 0 && K > 0 && N > 0);
    hipStream_t s = (hipStream_t)st;
    const long mk = (long)M * K, kn = (long)K * N, mn = (long)M * N;
    u64* delta = (u64*)scratch;
    u64* eps = delta + mk;
    u64* b0e = eps + kn;
    const OutPair z{(u64*)z0, (u64*)z1};
    matmul_open_local_kernel<<<sl_blocks(mk + kn + mn), 256, 0, s>>>(Pair{U(x0), U(x1)}, Pair{U(y0), U(y1)},
                                                                     Triple{U(a0), U(b0), U(c0), U(a1), U(b1), U(c1)}, delta, eps,
                                                                     b0e, z, mk, kn, mn);
    const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
    int ksplit = 1;
    if (2 * tiles < 256) {       // few output tiles (one image: M = 49 .. 784 rows in layer3 / 4): split K to fill the chip
        ksplit = (512 + 2 * tiles - 1) / (2 * tiles);
        const int kmax = (K + 63) / 64;
        if (ksplit > kmax) ksplit = kmax;
        if (ksplit < 1) ksplit = 1;
    }
    const dim3 grid((N + 63) / 64, (M + 63) / 64, 2 * ksplit);
    ring_gemm2_kernel<<<grid, 256, 0, s>>>(delta, eps, b0e, U(b1), U(a0), U(a1), z, M, K, N, ksplit);
    return launch_status();
}

int primia_trunc_col2out_2p(const int64_t* res0, const int64_t* res1, const int64_t* bias0, const int64_t* bias1,
                            int64_t* out0, int64_t* out1, int B, int HoWo, int O, int64_t div, primia_stream_t st) {
    PRIMIA_REQUIRE(res0 && res1 && out0 && out1 && B > 0 && HoWo > 0 && O > 0 && div > 0 &&
                   ((bias0 == nullptr) == (bias1 == nullptr)));
    const long total = (long)B * HoWo * O;
    trunc_col2out_2p_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)st>>>(
        Pair{U(res0), U(res1)}, Pair{U(bias0), U(bias1)}, OutPair{(u64*)out0, (u64*)out1}, B, HoWo, O, (u64)div);
    return launch_status();
}

int primia_pool_unroll_syft_2p(const int64_t* x0, const int64_t* x1, int64_t* out0, int64_t* out1, int B, int C, int H, int W,
                               int k, int stride, int pad, primia_stream_t st) {
    PRIMIA_REQUIRE(x0 && x1 && out0 && out1 && B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0);
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    PRIMIA_REQUIRE(Ho > 0 && Wo > 0);
    const long total = (long)B * C * Ho * Wo * k * k;
    pool_unroll_2p_kernel<<<ceil_div(total, 256), 256, 0, (hipStream_t)st>>>(Pair{U(x0), U(x1)}, OutPair{(u64*)out0, (u64*)out1},
                                                                              B, C, H, W, k, stride, pad, Ho, Wo);
    return launch_status();
}

}  // extern "C"
