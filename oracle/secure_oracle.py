"""ORACLE (test infrastructure, not product code) — CPU restatement of the arithmetic of PriMIA's
encrypted-inference path: PySyft 0.2.9 fixed-precision additive secret sharing over Z_2^64 with
Beaver triples (SPDZ) and Function Secret Sharing comparisons, as vendored under
/root/reference/syft (commit 9dc09f3d, syft/version.py:1).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Everything is plain NumPy integer arithmetic; int64 tensors wrap mod 2^64 like torch.int64.
Each function cites the reference lines it follows (paths relative to
/root/reference/syft/frameworks/torch/).  Randomness is ALWAYS an explicit input (shares, triples,
FSS keys, re-sharing masks): outputs are bit-exact functions of it, which is the only meaningful
notion of parity for this path (SURVEY.md §7 "Defining bit-exact").

Third-party arithmetic that is not in the reference tree:
  * torch 1.4 int64 `/` = C truncation toward zero (precision.py:149-151 relies on it);
  * `shaloop.sha256_loop_func / sha512_loop_func` (unpinned wheel, mpc/fss.py:14,532,581):
    independent SHA-256 / SHA-512 of each 16-byte row — here hashlib, or the C loop in
    oracle/sha_loop.c when built;
  * numpy 1.18 promotion at fss.py:317,390-394,423-426 (uint64 x int64 -> float64 -> uint64 cast
    that wraps): restated with exact integers.

Parity pin: tests/golden/make_secure_golden.py loads the reference's own nn/functional.py and
mpc/fss.py (with the documented shims) in the build container and checks this file against them
on seeded inputs; the resulting vectors are committed under tests/golden/.
"""
import ctypes
import hashlib
import os

import numpy as np

U64 = np.uint64
I64 = np.int64
MASK31 = U64(0x7FFFFFFF)
NOT1 = U64(0xFFFFFFFFFFFFFFFE)
N_BITS = 32     # fss.py:27  n
LAMBDA = 127    # fss.py:26  λ


# ------------------------------------------------------------------------------------------------
# ring helpers
# ------------------------------------------------------------------------------------------------
def wrap(x):
    """int64 view of arithmetic mod 2^64."""
    return np.asarray(x).astype(U64).view(I64) if np.asarray(x).dtype != I64 else np.asarray(x)


def radd(a, b):
    return (np.asarray(a, I64).view(U64) + np.asarray(b, I64).view(U64)).view(I64)


def rsub(a, b):
    return (np.asarray(a, I64).view(U64) - np.asarray(b, I64).view(U64)).view(I64)


def rmul(a, b):
    return (np.asarray(a, I64).view(U64) * np.asarray(b, I64).view(U64)).view(I64)


def rmatmul(a, b):
    """int64 matmul with wrap-around (torch.matmul on LongTensors, mpc/spdz.py:54-59)."""
    a = np.asarray(a, I64).view(U64)
    b = np.asarray(b, I64).view(U64)
    with np.errstate(over="ignore"):
        return (a @ b).view(I64)


def trunc_div(x, d):
    """C-style int64 division truncating toward zero: torch-1.4 `share / divisor`
    (tensors/interpreters/additive_shared.py:672-678, precision.py:146-154)."""
    x = np.asarray(x, I64)
    q = np.abs(x) // I64(d)          # |x| fits except x == -2^63, handled below
    q = np.where(x < 0, -q, q)
    mn = np.iinfo(I64).min
    if np.any(x == mn):
        q = np.where(x == mn, I64(-((1 << 63) // int(d))), q)
    return q.astype(I64)


# ------------------------------------------------------------------------------------------------
# S1 / S2: fixed precision and additive sharing
# ------------------------------------------------------------------------------------------------
def fix_encode(x, base=10, precision_fractional=16):
    """precision.py:117-132 — float32 multiply by base**p, then .long() (truncate)."""
    up = np.asarray(x, np.float32) * np.float32(float(base ** precision_fractional))
    return np.trunc(up.astype(np.float64)).astype(I64)


def fix_decode(q, base=10, precision_fractional=16):
    """precision.py:134-144 — .float() / base**p."""
    return np.asarray(q, I64).astype(np.float32) / np.float32(float(base ** precision_fractional))


def generate_shares(secret, r):
    """additive_shared.py:336-365 for 2 workers: share_0 = r (uniform int64), share_1 = secret - r."""
    return np.asarray(r, I64), rsub(secret, r)


def reconstruct(s0, s1):
    """additive_shared.py:287-301 — wrapping sum of the shares."""
    return radd(s0, s1)


# ------------------------------------------------------------------------------------------------
# S3 / S10: im2col and pool unrolling (nn/functional.py:78-201, 311-417)
# ------------------------------------------------------------------------------------------------
def pre_conv(x, w, stride=1, padding=0):
    """_pre_conv: x [B,C,H,W] -> im [B, Ho*Wo, C*kh*kw] (row order (ho, wo), column order
    (c, r, s)), w [O,C,kh,kw] -> [C*kh*kw, O]."""
    x = np.asarray(x)
    B, C, H, W = x.shape
    O, Cw, kh, kw = w.shape
    assert C == Cw
    Ho = int(((H + 2 * padding - (kh - 1) - 1) / stride) + 1)
    Wo = int(((W + 2 * padding - (kw - 1) - 1) / stride) + 1)
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding)))
    cols = np.empty((B, Ho * Wo, C * kh * kw), dtype=x.dtype)
    k = 0
    for c in range(C):
        for r in range(kh):
            for s in range(kw):
                patch = xp[:, c, r:r + stride * (Ho - 1) + 1:stride, s:s + stride * (Wo - 1) + 1:stride]
                cols[:, :, k] = patch.reshape(B, Ho * Wo)
                k += 1
    return cols, np.ascontiguousarray(np.asarray(w).reshape(O, -1).T), (B, O, Ho, Wo)


def post_conv(bias, res, B, O, Ho, Wo):
    """_post_conv: (+ bias share), permute(0,2,1), reshape to [B,O,Ho,Wo]."""
    if bias is not None:
        res = radd(res, np.asarray(bias, I64))
    return np.ascontiguousarray(np.transpose(res, (0, 2, 1)).reshape(B, O, Ho, Wo))


def pre_pool(x, kernel, stride, padding):
    """_pre_pool: x [B,C,H,W] -> [B, C, Ho*Wo, k*k]; zero padding (of the SHARES)."""
    x = np.asarray(x)
    B, C, H, W = x.shape
    Ho = int(((H + 2 * padding - (kernel - 1) - 1) / stride) + 1)
    Wo = int(((W + 2 * padding - (kernel - 1) - 1) / stride) + 1)
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding)))
    out = np.empty((B, C, Ho * Wo, kernel * kernel), dtype=x.dtype)
    k = 0
    for r in range(kernel):
        for s in range(kernel):
            patch = xp[:, :, r:r + stride * (Ho - 1) + 1:stride, s:s + stride * (Wo - 1) + 1:stride]
            out[:, :, :, k] = patch.reshape(B, C, Ho * Wo)
            k += 1
    return out, (B, C, Ho, Wo)


# ------------------------------------------------------------------------------------------------
# S4 / S5: Beaver multiplication (mpc/spdz.py:21-122, mpc/beaver.py:7-63)
# ------------------------------------------------------------------------------------------------
def build_triple(op, a, b, ra, rb, rc):
    """beaver.build_triple with the randomness given: a, b uniform int64 of the operand shapes,
    c = a ∘ b (wrapping); each split as (r, v - r).  Returns per-party (a_j, b_j, c_j)."""
    c = rmul(a, b) if op == "mul" else rmatmul(a, b)
    sa, sb, sc = generate_shares(a, ra), generate_shares(b, rb), generate_shares(c, rc)
    return [(sa[j], sb[j], sc[j]) for j in range(2)]


def spdz_mask(x_j, y_j, a_j, b_j):
    """spdz.py:21-45 — party-local: delta_j = x_j - a_j, epsilon_j = y_j - b_j."""
    return rsub(x_j, a_j), rsub(y_j, b_j)


def spdz_compute(j, delta, epsilon, a_j, b_j, c_j, op):
    """spdz.py:63-122 — z_j = delta∘b_j + a_j∘epsilon + c_j (+ delta∘epsilon for j == 0)."""
    f = rmul if op == "mul" else rmatmul
    z = radd(radd(f(delta, b_j), f(a_j, epsilon)), c_j)
    if j == 0:
        z = radd(z, f(delta, epsilon))
    return z


def beaver(op, x, y, triple):
    """spdz_mul (spdz.py:125-197) for two parties: x, y, triple are per-party lists."""
    d, e = zip(*[spdz_mask(x[j], y[j], triple[j][0], triple[j][1]) for j in range(2)])
    delta, epsilon = radd(d[0], d[1]), radd(e[0], e[1])
    return [spdz_compute(j, delta, epsilon, *triple[j], op) for j in range(2)]


# ------------------------------------------------------------------------------------------------
# S8 / S9: Function Secret Sharing (mpc/fss.py)
# ------------------------------------------------------------------------------------------------
_sha_lib = None


def _load_sha_lib():
    global _sha_lib
    if _sha_lib is None:
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libsha_loop.so")
        _sha_lib = ctypes.CDLL(p) if os.path.exists(p) else False
    return _sha_lib


def sha_loop(seeds16, kind):
    """shaloop.sha256_loop_func / sha512_loop_func: hash every 16-byte row independently.
    seeds16: uint8 [n, 16] -> uint8 [n, 32 | 64]."""
    n = seeds16.shape[0]
    width = 32 if kind == 256 else 64
    out = np.empty((n, width), dtype=np.uint8)
    lib = _load_sha_lib()
    seeds16 = np.ascontiguousarray(seeds16)
    if lib:
        fn = lib.sha256_loop if kind == 256 else lib.sha512_loop
        fn(seeds16.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(n))
        return out
    h = hashlib.sha256 if kind == 256 else hashlib.sha512
    for i in range(n):
        out[i] = np.frombuffer(h(seeds16[i].tobytes()).digest(), dtype=np.uint8)
    return out


def _seed_bytes(seed):
    """fss.py:521-526 — seed [2, n] uint64 -> [n, 16] bytes (little-endian words, word 0 first)."""
    return np.ascontiguousarray(np.asarray(seed, U64).T).view(np.uint8).reshape(-1, 16)


def prg_G(seed):
    """fss.py:516-547 — SHA-256 PRG: [2, n] -> [2 sides, 3 rows (s_hi&~1, s_lo, t), n]."""
    buf = sha_loop(_seed_bytes(seed), 256).view(U64).T   # [4, n]
    v = np.empty((2, 3, seed.shape[1]), dtype=U64)
    for side in range(2):
        v[side, 0] = buf[2 * side] & NOT1
        v[side, 1] = buf[2 * side + 1]
        v[side, 2] = buf[2 * side] & U64(1)
    return v


def prg_H(seed):
    """fss.py:553-601 — SHA-512 PRG: [2, n] -> [2 sides, 6 rows (σ(2), τ, s(2), t), n]."""
    buf = sha_loop(_seed_bytes(seed), 512).view(U64).T   # [8, n]
    v = np.empty((2, 6, seed.shape[1]), dtype=U64)
    for side in range(2):
        for half in range(2):
            w = buf[4 * side + 2 * half]
            v[side, 3 * half + 0] = w & NOT1
            v[side, 3 * half + 1] = buf[4 * side + 2 * half + 1]
            v[side, 3 * half + 2] = w & U64(1)
    return v


def bit_decomposition(x):
    """fss.py:484-492 — low 32 bits, most significant first: returns [32, n] of 0/1 (uint64)."""
    x = np.asarray(x).astype(U64) & U64(0xFFFFFFFF)
    return np.stack([(x >> U64(31 - i)) & U64(1) for i in range(N_BITS)])


def convert(x):
    """fss.py:655-661 — low 31 bits of the LAST word, as int64."""
    return (x[-1] & MASK31).astype(I64)


def _sel(pair, bit):
    """multi_dim_filter (fss.py:650-652): pair[0] where bit == 0, pair[1] where bit == 1."""
    return np.where(bit.astype(bool), pair[1], pair[0])


# The reference fans FSS work above MULTI_LIMIT elements out over N_CORES processes, slicing the element axis
# (fss.py:43-44, 47-95, 214-266).  Every element is independent, so the sliced result is the serial one; the
# oracle does the same when a process pool is installed (full-size parity tests and the CPU baseline).
MULTI_LIMIT = 50_000
_POOL = None


def use_pool(pool, n_slices=None):
    """Install (or, with None, remove) a multiprocessing pool for dif_keygen / dif_eval above MULTI_LIMIT."""
    global _POOL
    _POOL = None if pool is None else (pool, int(n_slices or getattr(pool, "_processes", 8)))


def _cuts(n, k):
    step = -(-n // k)
    return [(i, min(n, i + step)) for i in range(0, n, step)]


def _keygen_job(args):
    return _dif_keygen_serial(*args)[1]


def _eval_job(args):
    return _dif_eval_serial(*args)


def dif_keygen(alpha, s0_pair):
    """DIF.keygen; element slices in worker processes above MULTI_LIMIT (fss.py:47-95), else in place."""
    n = alpha.shape[0]
    if _POOL is None or n <= MULTI_LIMIT:
        return _dif_keygen_serial(alpha, s0_pair)
    pool, k = _POOL
    s0_pair = np.asarray(s0_pair, U64)
    parts = pool.map(_keygen_job, [(alpha[a:b], np.ascontiguousarray(s0_pair[:, :, a:b])) for a, b in _cuts(n, k)])
    keys = [{name: np.concatenate([p[b][name] for p in parts], axis=-1) for name in parts[0][b]} for b in range(2)]
    return alpha, keys


def dif_eval(b, x, key):
    """DIF.eval; element slices in worker processes above MULTI_LIMIT (fss.py:214-266), else in place."""
    x = np.asarray(x)
    n = x.shape[0]
    if _POOL is None or n <= MULTI_LIMIT:
        return _dif_eval_serial(b, x, key)
    pool, k = _POOL
    jobs = [(b, x[a:c], {name: np.ascontiguousarray(v[..., a:c]) for name, v in key.items()}) for a, c in _cuts(n, k)]
    return np.concatenate(pool.map(_eval_job, jobs))


def _dif_keygen_serial(alpha, s0_pair):
    """DIF.keygen (fss.py:344-398) with explicit randomness.
    alpha : uint64 [n] in [0, 2^32);  s0_pair : uint64 [2 parties, 2 words, n] (word 0 < 2^63).
    Returns (alpha, keys) with keys[b] = dict(s0 [2,n], bits [32,4,n] (τL,tL,τR,tR), cw_sigma
    [32,2,n], cw_s [32,2,n], cw_leaf int32 [33,n]) — the content of the reference key tuple
    (alpha, s[0][b], *_CW, CW_leaf); correction words are identical for both parties."""
    n = alpha.shape[0]
    a_bits = bit_decomposition(alpha)
    s = np.asarray(s0_pair, U64).copy()           # [2, 2, n]
    t = np.stack([np.zeros(n, U64), np.ones(n, U64)])
    bits = np.empty((N_BITS, 4, n), U64)
    cw_sigma = np.empty((N_BITS, 2, n), U64)
    cw_s = np.empty((N_BITS, 2, n), U64)
    cw_leaf = np.empty((N_BITS + 1, n), I64)
    for i in range(N_BITS):
        ai = a_bits[i]
        h = [prg_H(s[0]), prg_H(s[1])]            # each [2, 6, n]
        # rows: 0-1 σ, 2 τ, 3-4 s, 5 t ; side 0 = L, 1 = R
        s_rand = np.where(ai.astype(bool), h[0][0, 3:5] ^ h[1][0, 3:5], h[0][1, 3:5] ^ h[1][1, 3:5])
        sg_rand = np.where(ai.astype(bool), h[0][0, 0:2] ^ h[1][0, 0:2], h[0][1, 0:2] ^ h[1][1, 0:2])
        one = np.ones((1, n), U64)
        tab = np.empty((2, 6, n), U64)
        # SwitchTableDIF (fss.py:635-647): leaf table switched by 1-α, next table by α
        tab[0, 0:3] = ai * np.concatenate([sg_rand, one])
        tab[1, 0:3] = (U64(1) - ai) * np.concatenate([sg_rand, one])
        tab[0, 3:6] = (U64(1) - ai) * np.concatenate([s_rand, one])
        tab[1, 3:6] = ai * np.concatenate([s_rand, one])
        cw = tab ^ h[0] ^ h[1]
        # compress (fss.py:431-453) then uncompress (:456-477)
        bits[i] = np.stack([cw[0, 2], cw[0, 5], cw[1, 2], cw[1, 5]]) & U64(1)
        cw_sigma[i] = np.where(ai.astype(bool), cw[1, 0:2], cw[0, 0:2])
        cw_s[i] = np.where(ai.astype(bool), cw[0, 3:5], cw[1, 3:5])
        cwi = np.empty((2, 6, n), U64)
        for side in range(2):
            cwi[side, 0:2] = cw_sigma[i]
            cwi[side, 2] = bits[i][2 * side]
            cwi[side, 3:5] = cw_s[i]
            cwi[side, 5] = bits[i][2 * side + 1]
        sig, tau = [None, None], [None, None]
        s_next = np.empty_like(s)
        t_next = np.empty_like(t)
        for b in range(2):
            dual = h[b] ^ (t[b] * cwi)
            state = _sel(dual, ai)                 # stay on the special path
            s_next[b], t_next[b] = state[3:5], state[5]
            anti = _sel(dual, U64(1) - ai)         # leave it
            sig[b], tau[b] = anti[0:2], anti[2]
        sign = np.where(tau[1].astype(bool), I64(-1), I64(1))
        cw_leaf[i] = sign * (I64(1) - convert(sig[0]) + convert(sig[1]) - (I64(1) - ai.astype(I64)))
        s, t = s_next, t_next
    sign = np.where(t[1].astype(bool), I64(-1), I64(1))
    cw_leaf[N_BITS] = sign * (I64(1) - convert(s[0]) + convert(s[1]))
    cw_leaf32 = cw_leaf.astype(np.int32)          # CW_leaf.astype(np.int32) (fss.py:396)
    keys = [dict(s0=np.asarray(s0_pair, U64)[b].copy(), bits=bits.astype(np.uint8), cw_sigma=cw_sigma,
                 cw_s=cw_s, cw_leaf=cw_leaf32) for b in range(2)]
    return alpha, keys


def _dif_eval_serial(b, x, key):
    """DIF.eval (fss.py:400-428): int64 share of [x <= alpha]... evaluated on the masked input x."""
    x_bits = bit_decomposition(x)
    n = x_bits.shape[1]
    s = key["s0"].copy()
    t = np.full(n, b, U64)
    leaf = key["cw_leaf"].astype(I64)
    sgn = I64(-1) if b else I64(1)
    acc = np.zeros(n, U64)
    for i in range(N_BITS):
        h = prg_H(s)
        cwi = np.empty((2, 6, n), U64)
        for side in range(2):
            cwi[side, 0:2] = key["cw_sigma"][i]
            cwi[side, 2] = key["bits"][i][2 * side]
            cwi[side, 3:5] = key["cw_s"][i]
            cwi[side, 5] = key["bits"][i][2 * side + 1]
        state = _sel(h ^ (t * cwi), x_bits[i])
        sigma, tau, s, t = state[0:2], state[2], state[3:5], state[5]
        out_i = sgn * (tau.astype(I64) * leaf[i] + convert(sigma))
        acc = acc + out_i.view(U64)
    out_n = sgn * (t.astype(I64) * leaf[N_BITS] + convert(s))
    return (acc + out_n.view(U64)).view(I64)


def dpf_keygen(alpha, s0_pair):
    """DPF.keygen (fss.py:286-318) with explicit randomness; beta = 1."""
    n = alpha.shape[0]
    a_bits = bit_decomposition(alpha)
    s = np.asarray(s0_pair, U64).copy()
    t = np.stack([np.zeros(n, U64), np.ones(n, U64)])
    bits = np.empty((N_BITS, 2, n), U64)
    cw_s = np.empty((N_BITS, 2, n), U64)
    for i in range(N_BITS):
        ai = a_bits[i]
        g = [prg_G(s[0]), prg_G(s[1])]             # [2 sides, 3, n]
        s_rand = np.where(ai.astype(bool), g[0][0, 0:2] ^ g[1][0, 0:2], g[0][1, 0:2] ^ g[1][1, 0:2])
        one = np.ones((1, n), U64)
        tab = np.empty((2, 3, n), U64)
        tab[0] = (U64(1) - ai) * np.concatenate([s_rand, one])
        tab[1] = ai * np.concatenate([s_rand, one])
        cw = tab ^ g[0] ^ g[1]
        bits[i] = np.stack([cw[0, 2], cw[1, 2]]) & U64(1)
        cw_s[i] = np.where(ai.astype(bool), cw[0, 0:2], cw[1, 0:2])
        cwi = np.empty((2, 3, n), U64)
        for side in range(2):
            cwi[side, 0:2] = cw_s[i]
            cwi[side, 2] = bits[i][side]
        s_next, t_next = np.empty_like(s), np.empty_like(t)
        for b in range(2):
            state = _sel(g[b] ^ (t[b] * cwi), ai)
            s_next[b], t_next[b] = state[0:2], state[2]
        s, t = s_next, t_next
    sign = np.where(t[1].astype(bool), I64(-1), I64(1))
    cw_n = sign * (I64(1) - convert(s[0]) + convert(s[1]))
    keys = [dict(s0=np.asarray(s0_pair, U64)[b].copy(), bits=bits.astype(np.uint8), cw_s=cw_s, cw_n=cw_n.astype(I64))
            for b in range(2)]
    return alpha, keys


def dpf_eval(b, x, key):
    """DPF.eval (fss.py:320-338): int64 share of [x == alpha]."""
    x_bits = bit_decomposition(x)
    n = x_bits.shape[1]
    s = key["s0"].copy()
    t = np.full(n, b, U64)
    for i in range(N_BITS):
        g = prg_G(s)
        cwi = np.empty((2, 3, n), U64)
        for side in range(2):
            cwi[side, 0:2] = key["cw_s"][i]
            cwi[side, 2] = key["bits"][i][side]
        state = _sel(g ^ (t * cwi), x_bits[i])
        s, t = state[0:2], state[2]
    sgn = I64(-1) if b else I64(1)
    return (sgn * (t.astype(I64) * key["cw_n"] + convert(s))).astype(I64)


def split_alpha(alpha, r):
    """primitives.py:249-251 — alpha additively split mod 2^32 with the mask r: party 0 receives
    (alpha - r) mod 2^32, party 1 receives r."""
    r = np.asarray(r, U64) & U64(0xFFFFFFFF)
    return (np.asarray(alpha, U64) - r) & U64(0xFFFFFFFF), r


def fss_mask(x1_j, x2_j, alpha_j):
    """mask_builder (fss.py:189-204): r_j = (x1_j - x2_j) + alpha_j  (int64)."""
    return radd(rsub(x1_j, x2_j), np.asarray(alpha_j, U64).view(I64))


def fss_open(r0, r1):
    """fss.py:158 — sum(shares) % 2**32 (Python modulo: non-negative)."""
    return (radd(r0, r1).view(U64) & U64(0xFFFFFFFF))


def fss_le(x1, x2, alpha_shares, keys):
    """fss.le / fss_op(op='comp') (fss.py:97-185, 279): shares of [x1 <= x2] as int64."""
    r = [fss_mask(x1[j], x2[j], alpha_shares[j]) for j in range(2)]
    masked = fss_open(r[0], r[1])
    return [dif_eval(j, masked, keys[j]) for j in range(2)]


# ------------------------------------------------------------------------------------------------
# composites (tensors/interpreters/additive_shared.py, precision.py, nn/functional.py)
# ------------------------------------------------------------------------------------------------
def relu_fss(x, alpha_shares, keys, triple):
    """AST.relu under protocol fss (additive_shared.py:922-925): x * (x >= 0) with
    (x >= 0) = fss.le(0, x); the product is a Beaver mul WITHOUT truncation (the bit is unscaled)."""
    zero = [rsub(x[j], x[j]) for j in range(2)]
    bit = fss_le(zero, x, alpha_shares, keys)
    return beaver("mul", x, bit, triple)


def max_pair(left, right, alpha_shares, keys, triple):
    """max_half_split step (nn/functional.py:489-495): left + (right >= left) * (right - left)."""
    bit = fss_le(left, right, alpha_shares, keys)
    diff = [rsub(right[j], left[j]) for j in range(2)]
    prod = beaver("mul", bit, diff, triple)
    return [radd(left[j], prod[j]) for j in range(2)]


def fpt_matmul(x, w, triple, base=10, precision_fractional=16):
    """FixedPrecisionTensor.matmul on two shared operands (precision.py:419-463): Beaver matmul,
    then every party truncates ITS share by base**p (truncate -> AST._public_div)."""
    z = beaver("matmul", x, w, triple)
    d = base ** precision_fractional
    return [trunc_div(z[j], d) for j in range(2)]


def fpt_mul(x, y, triple, base=10, precision_fractional=16):
    """FPT * FPT on shared operands (precision.py:309-316, 356-358): Beaver mul + per-share trunc."""
    z = beaver("mul", x, y, triple)
    d = base ** precision_fractional
    return [trunc_div(z[j], d) for j in range(2)]


# ------------------------------------------------------------------------------------------------
# S7 / S10 / S11 / S12: layers and the ResNet-18 forward on shares, driven by a recorded dealer
# stream (so that the GPU path and this restatement consume IDENTICAL randomness, in the order the
# reference consumes it).
# ------------------------------------------------------------------------------------------------
class ReplayDealer:
    """Feeds primitives from a recorded stream: entries ("triple", op, per-party (a,b,c)),
    ("dif", n, alpha, s0_pair, r) and ("mask", r).  DIF keys are re-derived HERE from
    (alpha, s0_pair) with the oracle's keygen."""

    def __init__(self, log):
        self.log = list(log)
        self.pos = 0

    def _next(self, kind):
        e = self.log[self.pos]
        self.pos += 1
        assert e[0] == kind, f"dealer stream out of order: wanted {kind}, got {e[0]} at {self.pos - 1}"
        return e

    def triple(self, op, xshape, yshape):
        _, eop, t = self._next("triple")
        assert eop == op and t[0][0].size == int(np.prod(xshape)) and t[0][1].size == int(np.prod(yshape))
        # the producer may have used a flattened view of the same elements (e.g. [rows, w] for
        # [B, C, P, w]); element order is identical, so only the shape is restored here
        cshape = np.broadcast_shapes(tuple(xshape), tuple(yshape)) if op == "mul" else tuple(xshape[:-1]) + (yshape[-1],)
        return [(tj[0].reshape(xshape), tj[1].reshape(yshape), tj[2].reshape(cshape)) for tj in t]

    def dif_keys(self, n):
        _, en, alpha, s0, r = self._next("dif")
        assert en == n
        _, keys = dif_keygen(alpha.astype(U64), s0.view(U64))
        a0, a1 = split_alpha(alpha.astype(U64), r.astype(U64))
        return [a0, a1], keys

    def const_mask(self, *shape):
        _, r = self._next("mask")
        assert tuple(r.shape) == tuple(shape)
        return r


class OracleContext:
    """NumPy mirror of the reference's FPT > AST operations used by ResNet-18 inference."""

    def __init__(self, dealer, base=10, precision_fractional=16):
        self.dealer, self.base, self.pf = dealer, base, precision_fractional
        self.scale = base ** precision_fractional

    def share(self, q):
        return list(generate_shares(q, self.dealer.const_mask(*q.shape)))

    def add(self, a, b):
        return [radd(a[j], b[j]) for j in range(2)]

    def sub(self, a, b):
        return [rsub(a[j], b[j]) for j in range(2)]

    def neg(self, a):
        return [rmul(a[j], I64(-1)) for j in range(2)]

    def trunc(self, a, d):
        return [trunc_div(a[j], d) for j in range(2)]

    def sub_public_scalar(self, a, value):
        # additive_shared.py:453-484: the int becomes torch.tensor([value]).share(...)
        return self.sub(a, self.share(np.array([value], dtype=I64)))

    def beaver_mul(self, x, y):
        return beaver("mul", x, y, self.dealer.triple("mul", x[0].shape, y[0].shape))

    def fpt_mul(self, x, y):
        return self.trunc(self.beaver_mul(x, y), self.scale)

    def fpt_matmul(self, x, y):
        t = self.dealer.triple("matmul", x[0].shape, y[0].shape)
        return self.trunc(beaver("matmul", x, y, t), self.scale)

    def le(self, x1, x2):
        shape = x1[0].shape
        alpha_sh, keys = self.dealer.dif_keys(x1[0].size)
        out = fss_le([v.reshape(-1) for v in x1], [v.reshape(-1) for v in x2], alpha_sh, keys)
        return [o.reshape(shape) for o in out]

    def relu(self, x):
        bit = self.le(self.sub(x, x), x)
        return self.beaver_mul(x, bit)

    def max_pair(self, left, right):
        bit = self.le(left, right)
        return self.add(left, self.beaver_mul(bit, self.sub(right, left)))

    def conv2d(self, x, w, stride, padding):
        """nn/functional.py:204-308."""
        pre = [pre_conv(x[j], w[j], stride, padding) for j in range(2)]
        res = self.fpt_matmul([p[0] for p in pre], [p[1] for p in pre])
        return [post_conv(None, res[j], *pre[j][2]) for j in range(2)]

    def reciprocal_newton(self, v):
        """precision.py:507-518."""
        C = 20
        x = self.trunc(self.neg(self.sub_public_scalar(v, (C + 1) * self.scale)), C)
        for _ in range(79):
            vxx = self.fpt_mul(v, self.fpt_mul(x, x))
            y = self.neg(self.sub_public_scalar(vxx, (C + 1) * self.scale))
            x = self.trunc(self.fpt_mul(y, x), C)
        return x

    def batch_norm_eval(self, x, mean, var, weight, bias, inv=None):
        """nn/functional.py:44-75, eval branch.  `inv` = shares of newton(var) if computed earlier."""
        shp = x[0].shape
        rows = [np.ascontiguousarray(np.transpose(x[j], (1, 0, 2, 3)).reshape(shp[1], -1).T) for j in range(2)]
        if inv is None:
            inv = self.reciprocal_newton(var)
        normalized = self.fpt_mul(inv, self.sub(rows, mean))
        result = self.add(self.fpt_mul(normalized, weight), bias)
        return [np.ascontiguousarray(np.transpose(result[j].T.reshape(shp[1], shp[0], shp[2], shp[3]), (1, 0, 2, 3)))
                for j in range(2)]

    def max_pool2d_3x3s2(self, x):
        """nn/functional.py:460-508 for a 9-element window."""
        pre = [pre_pool(x[j], 3, 2, 1) for j in range(2)]
        im = [p[0] for p in pre]
        B, C, Ho, Wo = pre[0][1]

        def cols(t, a, b):
            return [np.ascontiguousarray(v[..., a:b]) for v in t]

        res = self.max_pair(cols(im, 0, 4), cols(im, 4, 8))
        res = self.max_pair(cols(res, 0, 2), cols(res, 2, 4))
        left = self.max_pair(cols(res, 0, 1), cols(res, 1, 2))
        res = self.max_pair(left, cols(im, 8, 9))
        return [r.reshape(B, C, Ho, Wo) for r in res]

    def avg_pool2d(self, x, k):
        out = []
        for j in range(2):
            im, (B, C, Ho, Wo) = pre_pool(x[j], k, k, 0)
            s = im.view(U64).sum(axis=-1, dtype=U64).view(I64)
            out.append(trunc_div(s, k * k).reshape(B, C, Ho, Wo))
        return out

    def linear(self, x, w, b):
        wt = [np.ascontiguousarray(w[j].T) for j in range(2)]
        return self.add(self.fpt_matmul(x, wt), b)


def share_order(keys):
    """Order in which model.fix_precision().share() walks a module: every parameter, THEN every buffer
    (syft/frameworks/torch/hook/hook.py:624-632,738-765 — `parameters()` before `buffers()`).  The 0-dim
    num_batches_tracked buffers draw nothing: generate_shares sizes its random share with
    LongTensor(torch.Size([])), an empty tensor (additive_shared.py:352), and eval mode never reads them."""
    keys = [k for k in keys if not k.endswith("num_batches_tracked")]
    buf = [k for k in keys if k.endswith("running_mean") or k.endswith("running_var")]
    return [k for k in keys if k not in buf] + buf


def secure_resnet_forward(ctx, state_dict, image, blocks=None, batched_newton=True):
    """inference.py:279-321 on numpy: share every parameter/buffer, share the image, run the
    forward with the stem swap (:289), return the output shares.  state_dict values / image are
    float32 numpy arrays.  batched_newton=False follows the reference's primitive order exactly
    (newton(running_var) inside every batch_norm call, nn/functional.py:62-69)."""
    p = {}
    for k in share_order(list(state_dict.keys())):
        p[k] = ctx.share(fix_encode(state_dict[k], ctx.base, ctx.pf))
    if blocks is None:
        blocks = [(f"layer{li}.{bi}", (2 if (li > 1 and bi == 0) else 1)) for li in range(1, 5) for bi in range(2)]
    x = ctx.share(fix_encode(image, ctx.base, ctx.pf))
    # newton(running_var) is image independent and element-wise: all BatchNorm layers are run as ONE
    # vector (same per-channel arithmetic as the reference's layer-by-layer calls,
    # nn/functional.py:62-69; only the grouping of the dealer's primitives differs).
    names = ["bn1"]
    for prefix, _ in blocks:
        names += [prefix + ".bn1", prefix + ".bn2"]
        if (prefix + ".downsample.0.weight") in p:
            names.append(prefix + ".downsample.1")
    inv = {n: None for n in names}
    if batched_newton:
        inv_all = ctx.reciprocal_newton([np.concatenate([p[n + ".running_var"][j] for n in names]) for j in range(2)])
        off = 0
        for n in names:
            k = p[n + ".running_var"][0].size
            inv[n] = [inv_all[j][off:off + k] for j in range(2)]
            off += k

    def bn(t, prefix):
        return ctx.batch_norm_eval(t, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"],
                                   p[prefix + ".bias"], inv=inv[prefix])

    x = ctx.conv2d(x, p["conv1.weight"], 2, 3)
    x = bn(x, "bn1")
    x = ctx.max_pool2d_3x3s2(x)
    x = ctx.relu(x)
    for prefix, stride in blocks:
        identity = x
        out = ctx.conv2d(x, p[prefix + ".conv1.weight"], stride, 1)
        out = ctx.relu(bn(out, prefix + ".bn1"))
        out = ctx.conv2d(out, p[prefix + ".conv2.weight"], 1, 1)
        out = bn(out, prefix + ".bn2")
        if (prefix + ".downsample.0.weight") in p:
            identity = ctx.conv2d(x, p[prefix + ".downsample.0.weight"], stride, 0)
            identity = bn(identity, prefix + ".downsample.1")
        x = ctx.relu(ctx.add(out, identity))
    x = ctx.avg_pool2d(x, x[0].shape[-1])
    x = [t.reshape(1, -1) for t in x]
    return ctx.linear(x, p["fc.weight"], p["fc.bias"])
