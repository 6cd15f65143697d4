// Function Secret Sharing kernels (2-party DIF comparison / DPF equality on 32-bit inputs,
// lambda = 127) — the per-party evaluation and the dealer's key generation of
// syft/frameworks/torch/mpc/fss.py, one comparison per lane.
//
// The PRG is SHA-512 (DIF) / SHA-256 (DPF) of the 16-byte seed (the reference calls the `shaloop`
// wheel on (n,16)-byte rows, fss.py:532,581).  A 16-byte message is one padded block whose words
// 2..15 are constants, so the message schedule is half constant-folded by the compiler.  The
// work is 64-bit rotate/xor/add chains on the vector ALU (no matrix-core form exists); keys are
// read once, coalesced over the comparison index (struct-of-arrays layout, see primia_hip.h).
#include "common.h"

namespace primia {

typedef unsigned long long u64;
typedef unsigned int u32;

// 64-bit rotate / shift / 3-way xor on explicit 32-bit halves: v_alignbit_b32 and v_bitop3_b32 are full-rate
// 32-bit ops, where the generic (x >> n) | (x << (64 - n)) compiles to two 64-bit shifts plus two ORs.
// n is a compile-time constant after unrolling.
__device__ __forceinline__ u64 ror64(u64 x, int n) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    u32 rl, rh;
    if (n < 32) {
        rl = __builtin_amdgcn_alignbit(hi, lo, n);
        rh = __builtin_amdgcn_alignbit(lo, hi, n);
    } else if (n == 32) {
        rl = hi;
        rh = lo;
    } else {
        rl = __builtin_amdgcn_alignbit(lo, hi, n - 32);
        rh = __builtin_amdgcn_alignbit(hi, lo, n - 32);
    }
    return ((u64)rh << 32) | rl;
}
__device__ __forceinline__ u64 shr64(u64 x, int n) {  // 0 < n < 32
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    return ((u64)(hi >> n) << 32) | __builtin_amdgcn_alignbit(hi, lo, n);
}
__device__ __forceinline__ u64 xor3_64(u64 a, u64 b, u64 c) {
    const u32 lo = __builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0x96);
    const u32 hi = __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), 0x96);
    return ((u64)hi << 32) | lo;
}
// Ch(e, f, g) = (e & f) ^ (~e & g) (truth table 0xCA) and Maj(a, b, c) (0xE8), one v_bitop3_b32 per half
__device__ __forceinline__ u64 bitop64(u64 a, u64 b, u64 c, int tt) {
    u32 lo, hi;
    if (tt == 0xCA) {
        lo = __builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0xCA);
        hi = __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), 0xCA);
    } else {
        lo = __builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0xE8);
        hi = __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), 0xE8);
    }
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 ror32(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ __forceinline__ u64 bswap64(u64 x) { return __builtin_bswap64(x); }
__device__ __forceinline__ u32 bswap32(u32 x) { return __builtin_bswap32(x); }

__constant__ u64 K512[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

__constant__ u32 K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
    0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
    0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
    0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};

// SHA-512 of the 16-byte message (seed word 0 little-endian bytes, then word 1).  Returns the
// digest as the reference sees it: 8 uint64 read little-endian from the 64 digest bytes.
__device__ __forceinline__ void sha512_seed(u64 s0, u64 s1, u64 out[8]) {
    u64 w[16];
    w[0] = bswap64(s0);
    w[1] = bswap64(s1);
    w[2] = 0x8000000000000000ULL;
#pragma unroll
    for (int i = 3; i < 15; ++i) w[i] = 0;
    w[15] = 128;
    u64 a = 0x6a09e667f3bcc908ULL, b = 0xbb67ae8584caa73bULL, c = 0x3c6ef372fe94f82bULL, d = 0xa54ff53a5f1d36f1ULL,
        e = 0x510e527fade682d1ULL, f = 0x9b05688c2b3e6c1fULL, g = 0x1f83d9abfb41bd6bULL, h = 0x5be0cd19137e2179ULL;
#pragma clang loop unroll(full)
    for (int i = 0; i < 80; ++i) {
        u64 wi;
        if (i < 16) {
            wi = w[i];
        } else {
            const u64 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const u64 g0 = xor3_64(ror64(w15, 1), ror64(w15, 8), shr64(w15, 7));
            const u64 g1 = xor3_64(ror64(w2, 19), ror64(w2, 61), shr64(w2, 6));
            wi = w[i & 15] + g0 + w[(i - 7) & 15] + g1;
            w[i & 15] = wi;
        }
        const u64 t1 = h + xor3_64(ror64(e, 14), ror64(e, 18), ror64(e, 41)) + bitop64(e, f, g, 0xCA) + K512[i] + wi;
        const u64 t2 = xor3_64(ror64(a, 28), ror64(a, 34), ror64(a, 39)) + bitop64(a, b, c, 0xE8);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    out[0] = bswap64(a + 0x6a09e667f3bcc908ULL);
    out[1] = bswap64(b + 0xbb67ae8584caa73bULL);
    out[2] = bswap64(c + 0x3c6ef372fe94f82bULL);
    out[3] = bswap64(d + 0xa54ff53a5f1d36f1ULL);
    out[4] = bswap64(e + 0x510e527fade682d1ULL);
    out[5] = bswap64(f + 0x9b05688c2b3e6c1fULL);
    out[6] = bswap64(g + 0x1f83d9abfb41bd6bULL);
    out[7] = bswap64(h + 0x5be0cd19137e2179ULL);
}

// SHA-256 of the same 16-byte message; digest as 4 little-endian uint64.
__device__ __forceinline__ void sha256_seed(u64 s0, u64 s1, u64 out[4]) {
    u32 w[16];
    w[0] = bswap32((u32)s0);
    w[1] = bswap32((u32)(s0 >> 32));
    w[2] = bswap32((u32)s1);
    w[3] = bswap32((u32)(s1 >> 32));
    w[4] = 0x80000000u;
#pragma unroll
    for (int i = 5; i < 15; ++i) w[i] = 0;
    w[15] = 128;
    const u32 iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    u32 a = iv[0], b = iv[1], c = iv[2], d = iv[3], e = iv[4], f = iv[5], g = iv[6], h = iv[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        u32 wi;
        if (i < 16) {
            wi = w[i];
        } else {
            const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const u32 g0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3);
            const u32 g1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + g0 + w[(i - 7) & 15] + g1;
            w[i & 15] = wi;
        }
        const u32 t1 = h + (ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + wi;
        const u32 t2 = (ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    const u32 dg[8] = {a + iv[0], b + iv[1], c + iv[2], d + iv[3], e + iv[4], f + iv[5], g + iv[6], h + iv[7]};
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (u64)bswap32(dg[2 * i]) | ((u64)bswap32(dg[2 * i + 1]) << 32);
}

// One side of H's output (fss.py:553-601): sigma (2 words), tau, s (2 words), t.
struct HSide {
    u64 sg0, sg1, tau, s0, s1, t;
};
__device__ __forceinline__ HSide h_side(const u64 buf[8], int side) {
    HSide r;
    const u64 w0 = buf[4 * side], w2 = buf[4 * side + 2];
    r.sg0 = w0 & ~1ULL;
    r.sg1 = buf[4 * side + 1];
    r.tau = w0 & 1ULL;
    r.s0 = w2 & ~1ULL;
    r.s1 = buf[4 * side + 3];
    r.t = w2 & 1ULL;
    return r;
}
__device__ __forceinline__ long conv31(u64 last_word) { return (long)(last_word & 0x7fffffffULL); }

// ---- DIF.eval (fss.py:400-428) ------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void dif_eval_kernel(int b, const u32* __restrict__ x, const u64* __restrict__ s0,
                                                       const uint8_t* __restrict__ cw_bits,
                                                       const u64* __restrict__ cw_sigma, const u64* __restrict__ cw_s,
                                                       const int32_t* __restrict__ cw_leaf, int64_t* __restrict__ out,
                                                       long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 sa = s0[i], sb = s0[n + i];
    u64 t = (u64)b;
    const u32 xv = x[i];
    u64 acc = 0;
    const long sgn = b ? -1 : 1;
    for (int lvl = 0; lvl < 32; ++lvl) {
        u64 buf[8];
        sha512_seed(sa, sb, buf);
        const int bit = (xv >> (31 - lvl)) & 1;
        HSide hs = h_side(buf, bit);
        // t * CW_i, side `bit` of the uncompressed correction word (fss.py:456-477)
        const u64 m = (u64)0 - t;  // all-ones if t == 1
        const u32 cb = cw_bits[(long)lvl * n + i];
        const u64 csg0 = cw_sigma[((long)lvl * 2 + 0) * n + i], csg1 = cw_sigma[((long)lvl * 2 + 1) * n + i];
        const u64 cs0 = cw_s[((long)lvl * 2 + 0) * n + i], cs1 = cw_s[((long)lvl * 2 + 1) * n + i];
        const u64 ctau = (cb >> (2 * bit)) & 1, ct = (cb >> (2 * bit + 1)) & 1;
        const u64 sg1 = hs.sg1 ^ (csg1 & m);
        const u64 tau = hs.tau ^ (ctau & m);
        sa = hs.s0 ^ (cs0 & m);
        sb = hs.s1 ^ (cs1 & m);
        t = hs.t ^ (ct & m);
        (void)csg0;
        const long leaf = (long)cw_leaf[(long)lvl * n + i];
        acc += (u64)(sgn * ((long)tau * leaf + conv31(sg1)));
    }
    const long leaf = (long)cw_leaf[32L * n + i];
    acc += (u64)(sgn * ((long)t * leaf + conv31(sb)));
    out[i] = (int64_t)acc;
}

// ---- fss.le for both parties hosted on this GPU (mpc/fss.py:97-185): mask_builder, the open and evaluate in ONE launch --
// masked = ((x1_0 - x2_0 + alpha_0) + (x1_1 - x2_1 + alpha_1)) mod 2^32, then each party's DIF.eval of it with ITS seed
// (blockIdx.y = party; the correction words are common).  x1 / x2 may be column ranges of a [rows][w] matrix (the
// unrolled pool image): element i = (row i / len, column start + i % len); x1 == NULL stands for shares of zero (relu).
struct LeOperand {
    const u64 *p0, *p1;
    int w, start;
};
__global__ __launch_bounds__(256, 2) void dif_eval_local_kernel(LeOperand x1, LeOperand x2, int len, const u64* __restrict__ alpha0,
                                                               const u64* __restrict__ alpha1, const u64* __restrict__ s0_0,
                                                               const u64* __restrict__ s0_1,
                                                               const uint8_t* __restrict__ cw_bits,
                                                               const u64* __restrict__ cw_sigma, const u64* __restrict__ cw_s,
                                                               const int32_t* __restrict__ cw_leaf, int64_t* __restrict__ out0,
                                                               int64_t* __restrict__ out1, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const long r = i / len, c = i - r * len;
    const long i2 = r * x2.w + x2.start + c;
    u64 m0 = alpha0[i] - x2.p0[i2], m1 = alpha1[i] - x2.p1[i2];
    if (x1.p0) {
        const long i1 = r * x1.w + x1.start + c;
        m0 += x1.p0[i1];
        m1 += x1.p1[i1];
    }
    const u32 xv = (u32)(m0 + m1);
    const u64* __restrict__ s0 = b ? s0_1 : s0_0;
    u64 sa = s0[i], sb = s0[n + i];
    u64 t = (u64)b;
    u64 acc = 0;
    const long sgn = b ? -1 : 1;
    for (int lvl = 0; lvl < 32; ++lvl) {
        u64 buf[8];
        sha512_seed(sa, sb, buf);
        const int bit = (xv >> (31 - lvl)) & 1;
        HSide hs = h_side(buf, bit);
        const u64 m = (u64)0 - t;
        const u32 cb = cw_bits[(long)lvl * n + i];
        const u64 csg1 = cw_sigma[((long)lvl * 2 + 1) * n + i];
        const u64 cs0 = cw_s[((long)lvl * 2 + 0) * n + i], cs1 = cw_s[((long)lvl * 2 + 1) * n + i];
        const u64 ctau = (cb >> (2 * bit)) & 1, ct = (cb >> (2 * bit + 1)) & 1;
        const u64 sg1 = hs.sg1 ^ (csg1 & m);
        const u64 tau = hs.tau ^ (ctau & m);
        sa = hs.s0 ^ (cs0 & m);
        sb = hs.s1 ^ (cs1 & m);
        t = hs.t ^ (ct & m);
        const long leaf = (long)cw_leaf[(long)lvl * n + i];
        acc += (u64)(sgn * ((long)tau * leaf + conv31(sg1)));
    }
    const long leaf = (long)cw_leaf[32L * n + i];
    acc += (u64)(sgn * ((long)t * leaf + conv31(sb)));
    (b ? out1 : out0)[i] = (int64_t)acc;
}

// ---- DPF.eval (fss.py:320-338) ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dpf_eval_kernel(int b, const u32* __restrict__ x, const u64* __restrict__ s0,
                                                       const uint8_t* __restrict__ cw_bits,
                                                       const u64* __restrict__ cw_s, const int64_t* __restrict__ cw_n,
                                                       int64_t* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 sa = s0[i], sb = s0[n + i];
    u64 t = (u64)b;
    const u32 xv = x[i];
    for (int lvl = 0; lvl < 32; ++lvl) {
        u64 buf[4];
        sha256_seed(sa, sb, buf);
        const int bit = (xv >> (31 - lvl)) & 1;
        const u64 w0 = buf[2 * bit];
        const u64 m = (u64)0 - t;
        const u32 cb = cw_bits[(long)lvl * n + i];
        sa = (w0 & ~1ULL) ^ (cw_s[((long)lvl * 2 + 0) * n + i] & m);
        sb = buf[2 * bit + 1] ^ (cw_s[((long)lvl * 2 + 1) * n + i] & m);
        t = (w0 & 1ULL) ^ ((u64)((cb >> bit) & 1) & m);
    }
    const long sgn = b ? -1 : 1;
    out[i] = sgn * ((long)t * cw_n[i] + conv31(sb));
}

// ---- build_fss_keys' host arithmetic on raw keystream words (mpc/primitives.py:237-253, mpc/fss.py:344-358,495-501), in place:
// alpha and its mask r are reduced mod 2^32, word 0 of both parties' seeds keeps 63 bits (randbit), and party 0's share of alpha
// is (alpha - r) mod 2^32 (party 1's is r) — what primia_amd.secure.Dealer did with torch `&` / `-` before round 6.
__global__ __launch_bounds__(256) void fss_alpha_split_kernel(u64* __restrict__ alpha, u64* __restrict__ s0p, u64* __restrict__ r,
                                                              u64* __restrict__ alpha0, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 m32 = 0xFFFFFFFFull, m63 = 0x7FFFFFFFFFFFFFFFull;
    const u64 a = alpha[i] & m32, rr = r[i] & m32;
    alpha[i] = a;
    r[i] = rr;
    alpha0[i] = (a - rr) & m32;
    s0p[i] &= m63;              // [party 0][word 0]
    s0p[2 * n + i] &= m63;      // [party 1][word 0]
}

// ---- DIF.keygen (fss.py:344-398), one comparison per lane --------------------------------------------
__global__ __launch_bounds__(256, 2) void dif_keygen_kernel(const u64* __restrict__ alpha, const u64* __restrict__ s0p,
                                                         uint8_t* __restrict__ cw_bits, u64* __restrict__ cw_sigma,
                                                         u64* __restrict__ cw_s, int32_t* __restrict__ cw_leaf,
                                                         long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 av = (u32)alpha[i];
    // s0_pair [party][word][n]
    u64 s[2][2] = {{s0p[i], s0p[n + i]}, {s0p[2 * n + i], s0p[3 * n + i]}};
    u64 t[2] = {0, 1};
    for (int lvl = 0; lvl < 32; ++lvl) {
        const int ai = (av >> (31 - lvl)) & 1;
        u64 h0[8], h1[8];
        sha512_seed(s[0][0], s[0][1], h0);
        sha512_seed(s[1][0], s[1][1], h1);
        HSide a0 = h_side(h0, 0), a1 = h_side(h0, 1), b0 = h_side(h1, 0), b1 = h_side(h1, 1);  // party a/b, side L/R
        // randomness re-used along the special path: side alpha (L if alpha bit is 1... see fss.py:371-372)
        const HSide& ka = ai ? a0 : a1;
        const HSide& kb = ai ? b0 : b1;
        const u64 sr0 = ka.s0 ^ kb.s0, sr1 = ka.s1 ^ kb.s1;      // s_rand
        const u64 gr0 = ka.sg0 ^ kb.sg0, gr1 = ka.sg1 ^ kb.sg1;  // sigma_rand
        // SwitchTableDIF (fss.py:635-647): leaf table on side L scaled by alpha, on R by 1-alpha;
        // next table on L by 1-alpha, on R by alpha.  CW = table ^ h0 ^ h1.
        const u64 mL = (u64)0 - (u64)ai, mR = ~mL;  // masks: alpha, 1-alpha
        HSide cwL, cwR;
        cwL.sg0 = (gr0 & mL) ^ a0.sg0 ^ b0.sg0;
        cwL.sg1 = (gr1 & mL) ^ a0.sg1 ^ b0.sg1;
        cwL.tau = ((u64)ai) ^ a0.tau ^ b0.tau;
        cwL.s0 = (sr0 & mR) ^ a0.s0 ^ b0.s0;
        cwL.s1 = (sr1 & mR) ^ a0.s1 ^ b0.s1;
        cwL.t = ((u64)(1 - ai)) ^ a0.t ^ b0.t;
        cwR.sg0 = (gr0 & mR) ^ a1.sg0 ^ b1.sg0;
        cwR.sg1 = (gr1 & mR) ^ a1.sg1 ^ b1.sg1;
        cwR.tau = ((u64)(1 - ai)) ^ a1.tau ^ b1.tau;
        cwR.s0 = (sr0 & mL) ^ a1.s0 ^ b1.s0;
        cwR.s1 = (sr1 & mL) ^ a1.s1 ^ b1.s1;
        cwR.t = ((u64)ai) ^ a1.t ^ b1.t;
        // compress (fss.py:431-453): sigma from side R if alpha else L; s from side L if alpha else R
        const u64 csg0 = ai ? cwR.sg0 : cwL.sg0, csg1 = ai ? cwR.sg1 : cwL.sg1;
        const u64 cs0 = ai ? cwL.s0 : cwR.s0, cs1 = ai ? cwL.s1 : cwR.s1;
        const u64 tauL = cwL.tau & 1, tL = cwL.t & 1, tauR = cwR.tau & 1, tR = cwR.t & 1;
        cw_bits[(long)lvl * n + i] = (uint8_t)(tauL | (tL << 1) | (tauR << 2) | (tR << 3));
        cw_sigma[((long)lvl * 2 + 0) * n + i] = csg0;
        cw_sigma[((long)lvl * 2 + 1) * n + i] = csg1;
        cw_s[((long)lvl * 2 + 0) * n + i] = cs0;
        cw_s[((long)lvl * 2 + 1) * n + i] = cs1;
        // advance both parties with the UNCOMPRESSED word (same sigma / s on both sides)
        u64 sig_last[2], tau_n[2];
        u64 ns[2][2], nt[2];
        for (int p = 0; p < 2; ++p) {
            const u64 m = (u64)0 - t[p];
            const HSide& L = p == 0 ? a0 : b0;
            const HSide& R = p == 0 ? a1 : b1;
            // state: side alpha -> (s, t); anti-state: side 1-alpha -> (sigma, tau)
            const HSide& st = ai ? R : L;
            const HSide& an = ai ? L : R;
            const u64 ct = ai ? tR : tL;
            const u64 ctau = ai ? tauL : tauR;
            ns[p][0] = st.s0 ^ (cs0 & m);
            ns[p][1] = st.s1 ^ (cs1 & m);
            nt[p] = st.t ^ (ct & m);
            sig_last[p] = an.sg1 ^ (csg1 & m);
            tau_n[p] = an.tau ^ (ctau & m);
        }
        const long sign = tau_n[1] ? -1 : 1;
        const long leaf = sign * (1 - conv31(sig_last[0]) + conv31(sig_last[1]) - (long)(1 - ai));
        cw_leaf[(long)lvl * n + i] = (int32_t)leaf;
        for (int p = 0; p < 2; ++p) {
            s[p][0] = ns[p][0];
            s[p][1] = ns[p][1];
            t[p] = nt[p];
        }
    }
    const long sign = t[1] ? -1 : 1;
    cw_leaf[32L * n + i] = (int32_t)(sign * (1 - conv31(s[0][1]) + conv31(s[1][1])));
}

// ---- DPF.keygen (fss.py:286-318) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void dpf_keygen_kernel(const u64* __restrict__ alpha, const u64* __restrict__ s0p,
                                                         uint8_t* __restrict__ cw_bits, u64* __restrict__ cw_s,
                                                         int64_t* __restrict__ cw_n, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 av = (u32)alpha[i];
    u64 s[2][2] = {{s0p[i], s0p[n + i]}, {s0p[2 * n + i], s0p[3 * n + i]}};
    u64 t[2] = {0, 1};
    for (int lvl = 0; lvl < 32; ++lvl) {
        const int ai = (av >> (31 - lvl)) & 1;
        u64 g[2][4];
        sha256_seed(s[0][0], s[0][1], g[0]);
        sha256_seed(s[1][0], s[1][1], g[1]);
        // side L = words 0,1 ; side R = words 2,3 ; s = (w0 & ~1, w1), t = w0 & 1
        u64 S0[2][2], S1[2][2], T[2][2];  // [party][side]
        for (int p = 0; p < 2; ++p)
            for (int sd = 0; sd < 2; ++sd) {
                S0[p][sd] = g[p][2 * sd] & ~1ULL;
                S1[p][sd] = g[p][2 * sd + 1];
                T[p][sd] = g[p][2 * sd] & 1ULL;
            }
        const int ks = ai ? 0 : 1;  // s_rand from side L if alpha else R (fss.py:305)
        const u64 sr0 = S0[0][ks] ^ S0[1][ks], sr1 = S1[0][ks] ^ S1[1][ks];
        const u64 mL = (u64)0 - (u64)ai, mR = ~mL;
        // SwitchTableDPF: side L scaled by 1-alpha, side R by alpha
        const u64 cL0 = (sr0 & mR) ^ S0[0][0] ^ S0[1][0], cL1 = (sr1 & mR) ^ S1[0][0] ^ S1[1][0];
        const u64 cLt = ((u64)(1 - ai)) ^ T[0][0] ^ T[1][0];
        const u64 cR0 = (sr0 & mL) ^ S0[0][1] ^ S0[1][1], cR1 = (sr1 & mL) ^ S1[0][1] ^ S1[1][1];
        const u64 cRt = ((u64)ai) ^ T[0][1] ^ T[1][1];
        const u64 cs0 = ai ? cL0 : cR0, cs1 = ai ? cL1 : cR1;  // compress: s from L if alpha else R
        const u64 tL = cLt & 1, tR = cRt & 1;
        cw_bits[(long)lvl * n + i] = (uint8_t)(tL | (tR << 1));
        cw_s[((long)lvl * 2 + 0) * n + i] = cs0;
        cw_s[((long)lvl * 2 + 1) * n + i] = cs1;
        for (int p = 0; p < 2; ++p) {
            const u64 m = (u64)0 - t[p];
            const u64 n0 = S0[p][ai] ^ (cs0 & m), n1 = S1[p][ai] ^ (cs1 & m);
            const u64 ntp = T[p][ai] ^ ((ai ? tR : tL) & m);
            s[p][0] = n0;
            s[p][1] = n1;
            t[p] = ntp;
        }
    }
    const long sign = t[1] ? -1 : 1;
    cw_n[i] = sign * (1 - conv31(s[0][1]) + conv31(s[1][1]));
}

__global__ __launch_bounds__(256) void fss_mask_kernel(const u64* __restrict__ x1, const u64* __restrict__ x2,
                                                       const u64* __restrict__ alpha, u64* __restrict__ r, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) r[i] = x1[i] - x2[i] + alpha[i];
}
__global__ __launch_bounds__(256) void fss_open_kernel(const u64* __restrict__ r0, const u64* __restrict__ r1,
                                                       u32* __restrict__ x, long n) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = (u32)(r0[i] + r1[i]);
}

}  // namespace primia

using namespace primia;

extern "C" {

int primia_fss_mask(const int64_t* x1, const int64_t* x2, const uint64_t* alpha_share, int64_t* r, int64_t n,
                    primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(x1 && x2 && alpha_share && r && n >= 0);
    if (n == 0) return PRIMIA_OK;
    long b = (n + 255) / 256;
    fss_mask_kernel<<<(int)(b > 4096 ? 4096 : b), 256, 0, (hipStream_t)st>>>((const u64*)x1, (const u64*)x2,
                                                                             (const u64*)alpha_share, (u64*)r, n);
    return launch_status();
}

int primia_fss_open(const int64_t* r0, const int64_t* r1, uint32_t* x, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(r0 && r1 && x && n >= 0);
    if (n == 0) return PRIMIA_OK;
    long b = (n + 255) / 256;
    fss_open_kernel<<<(int)(b > 4096 ? 4096 : b), 256, 0, (hipStream_t)st>>>((const u64*)r0, (const u64*)r1, x, n);
    return launch_status();
}

int primia_dif_eval(int b, const uint32_t* x, const uint64_t* s0, const uint8_t* cw_bits, const uint64_t* cw_sigma,
                    const uint64_t* cw_s, const int32_t* cw_leaf, int64_t* out, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE((b == 0 || b == 1) && x && s0 && cw_bits && cw_sigma && cw_s && cw_leaf && out && n >= 0);
    if (n == 0) return PRIMIA_OK;
    dif_eval_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)st>>>(b, x, (const u64*)s0, cw_bits, (const u64*)cw_sigma,
                                                                    (const u64*)cw_s, cw_leaf, out, n);
    return launch_status();
}

int primia_dif_eval_local(const int64_t* x1_0, const int64_t* x1_1, int w1, int start1, const int64_t* x2_0,
                          const int64_t* x2_1, int w2, int start2, int len, const uint64_t* alpha0, const uint64_t* alpha1,
                          const uint64_t* s0_0, const uint64_t* s0_1, const uint8_t* cw_bits, const uint64_t* cw_sigma,
                          const uint64_t* cw_s, const int32_t* cw_leaf, int64_t* out0, int64_t* out1, int64_t n,
                          primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(x2_0 && x2_1 && ((x1_0 == nullptr) == (x1_1 == nullptr)) && alpha0 && alpha1 && s0_0 && s0_1 && cw_bits &&
                   cw_sigma && cw_s && cw_leaf && out0 && out1 && n > 0 && len > 0 && n % len == 0);
    PRIMIA_REQUIRE(w2 >= len && start2 >= 0 && start2 + len <= w2 && (!x1_0 || (w1 >= len && start1 >= 0 && start1 + len <= w1)));
    const dim3 grid((unsigned)ceil_div(n, 256), 2);
    dif_eval_local_kernel<<<grid, 256, 0, (hipStream_t)st>>>(
        LeOperand{(const u64*)x1_0, (const u64*)x1_1, w1, start1}, LeOperand{(const u64*)x2_0, (const u64*)x2_1, w2, start2}, len,
        (const u64*)alpha0, (const u64*)alpha1, (const u64*)s0_0, (const u64*)s0_1, cw_bits, (const u64*)cw_sigma,
        (const u64*)cw_s, cw_leaf, out0, out1, n);
    return launch_status();
}

int primia_dpf_eval(int b, const uint32_t* x, const uint64_t* s0, const uint8_t* cw_bits, const uint64_t* cw_s,
                    const int64_t* cw_n, int64_t* out, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE((b == 0 || b == 1) && x && s0 && cw_bits && cw_s && cw_n && out && n >= 0);
    if (n == 0) return PRIMIA_OK;
    dpf_eval_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)st>>>(b, x, (const u64*)s0, cw_bits, (const u64*)cw_s,
                                                                    cw_n, out, n);
    return launch_status();
}

int primia_dif_keygen(const uint64_t* alpha, const uint64_t* s0_pair, uint8_t* cw_bits, uint64_t* cw_sigma,
                      uint64_t* cw_s, int32_t* cw_leaf, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(alpha && s0_pair && cw_bits && cw_sigma && cw_s && cw_leaf && n >= 0);
    if (n == 0) return PRIMIA_OK;
    dif_keygen_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)st>>>((const u64*)alpha, (const u64*)s0_pair, cw_bits,
                                                                      (u64*)cw_sigma, (u64*)cw_s, cw_leaf, n);
    return launch_status();
}

int primia_fss_alpha_split(uint64_t* alpha, uint64_t* s0_pair, uint64_t* r, uint64_t* alpha0, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(alpha && s0_pair && r && alpha0 && n > 0);
    fss_alpha_split_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)st>>>((u64*)alpha, (u64*)s0_pair, (u64*)r, (u64*)alpha0, n);
    return launch_status();
}

int primia_dpf_keygen(const uint64_t* alpha, const uint64_t* s0_pair, uint8_t* cw_bits, uint64_t* cw_s,
                      int64_t* cw_n, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;  // empty input: no-op, pointers may be null
    PRIMIA_REQUIRE(alpha && s0_pair && cw_bits && cw_s && cw_n && n >= 0);
    if (n == 0) return PRIMIA_OK;
    dpf_keygen_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)st>>>((const u64*)alpha, (const u64*)s0_pair, cw_bits,
                                                                      (u64*)cw_s, cw_n, n);
    return launch_status();
}

}  // extern "C"
