// ChaCha20 keystream on the GPU: the crypto provider's randomness (Beaver triples, re-sharing masks, FSS seeds).
//
// The reference's provider draws from torch's / NumPy's Mersenne Twisters seeded by the process
// (mpc/beaver.py:31-34, additive_shared.py:336-365, mpc/fss.py:344-358,495-501) — predictable generators; the
// privacy of additive sharing rests entirely on these masks, so the provider here runs a stream cipher keyed from
// the operating system's entropy pool instead.  State layout of the original ChaCha20 (64-bit block counter in
// words 12-13, 64-bit nonce in words 14-15); with counter < 2^32 the block equals RFC 8439's for the nonce
// (word13, word14, word15) — tests/test_gpu_secure.py checks the RFC 8439 §2.3.2 vector.
//
// One thread = one 64-byte block = 8 output words, stored as four 16-byte stores; HBM-write bound.
#include "common.h"

namespace primia {

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define PRIMIA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

struct ChaChaKey {
    uint32_t k[8];
    uint32_t n[2];
};

// `counter`: null, or a device word added to block0 — the serving form keeps the provider's block counter ON THE DEVICE so
// that a captured refill (hipGraph) draws fresh keystream at every replay (primia_chacha20_fill_ctr).
__global__ __launch_bounds__(256) void chacha20_kernel(ChaChaKey key, uint64_t block0, const uint64_t* __restrict__ counter,
                                                       uint64_t* __restrict__ out, int64_t nwords) {
    const int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk * 8 >= nwords) return;
    const uint64_t ctr = block0 + (counter ? *counter : 0) + (uint64_t)blk;
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
                      (uint32_t)ctr, (uint32_t)(ctr >> 32), key.n[0], key.n[1]};   // 12,13 counter; 14,15 nonce
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        PRIMIA_QR(x[0], x[4], x[8], x[12])
        PRIMIA_QR(x[1], x[5], x[9], x[13])
        PRIMIA_QR(x[2], x[6], x[10], x[14])
        PRIMIA_QR(x[3], x[7], x[11], x[15])
        PRIMIA_QR(x[0], x[5], x[10], x[15])
        PRIMIA_QR(x[1], x[6], x[11], x[12])
        PRIMIA_QR(x[2], x[7], x[8], x[13])
        PRIMIA_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] += s[i];
    uint64_t* o = out + blk * 8;
    if (blk * 8 + 8 <= nwords) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(o + 2 * i) = u32x4{x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]};
    } else {
        for (int i = 0; blk * 8 + i < nwords; ++i) o[i] = (uint64_t)x[2 * i] | ((uint64_t)x[2 * i + 1] << 32);
    }
}

}  // namespace primia

using namespace primia;

extern "C" int primia_chacha20_fill(uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3, uint64_t nonce,
                                    uint64_t block0, int64_t* out, int64_t n, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(out && n > 0 && ((uintptr_t)out & 15) == 0);
    ChaChaKey key;
    const uint64_t kk[4] = {k0, k1, k2, k3};
    for (int i = 0; i < 4; ++i) {
        key.k[2 * i] = (uint32_t)kk[i];
        key.k[2 * i + 1] = (uint32_t)(kk[i] >> 32);
    }
    key.n[0] = (uint32_t)nonce;
    key.n[1] = (uint32_t)(nonce >> 32);
    const int64_t blocks = (n + 7) / 8;
    chacha20_kernel<<<ceil_div(blocks, 256), 256, 0, (hipStream_t)st>>>(key, block0, nullptr, (uint64_t*)out, n);
    return launch_status();
}

namespace primia {
__global__ void u64_add_kernel(uint64_t* word, uint64_t delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *word += delta;
}
}  // namespace primia

extern "C" int primia_chacha20_fill_ctr(uint64_t k0, uint64_t k1, uint64_t k2, uint64_t k3, uint64_t nonce,
                                        const uint64_t* counter, uint64_t block_offset, int64_t* out, int64_t n,
                                        primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(counter && out && n > 0 && ((uintptr_t)out & 15) == 0);
    ChaChaKey key;
    const uint64_t kk[4] = {k0, k1, k2, k3};
    for (int i = 0; i < 4; ++i) {
        key.k[2 * i] = (uint32_t)kk[i];
        key.k[2 * i + 1] = (uint32_t)(kk[i] >> 32);
    }
    key.n[0] = (uint32_t)nonce;
    key.n[1] = (uint32_t)(nonce >> 32);
    const int64_t blocks = (n + 7) / 8;
    chacha20_kernel<<<ceil_div(blocks, 256), 256, 0, (hipStream_t)st>>>(key, block_offset, counter, (uint64_t*)out, n);
    return launch_status();
}

extern "C" int primia_u64_add(uint64_t* word, uint64_t delta, primia_stream_t st) {
    PRIMIA_REQUIRE(word);
    u64_add_kernel<<<1, 64, 0, (hipStream_t)st>>>(word, delta);
    return launch_status();
}
