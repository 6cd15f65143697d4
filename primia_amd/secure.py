"""Encrypted inference on GPU: fixed-precision additive secret sharing over Z_2^64 between two
parties (model_owner = party 0, data_owner = party 1) with a dealer (crypto provider) handing out
Beaver triples and FSS keys — the arithmetic of PySyft's FPT > AST tensor chain as PriMIA's
inference.py drives it (inference.py:279-321, protocol "fss").

Each method below is the host-side orchestration of one reference operation; the arithmetic runs
in the party-local HIP kernels of libprimia_hip.so (the functions PySyft registers with
@allow_command).  A shared tensor is a pair [share_0, share_1] of int64 device tensors.  An
"open" sums the two parties' buffers: with both parties on one GPU that is one ring-add kernel,
with one party per GPU it is a 2-rank RCCL all-reduce (`DistOpener`).

Randomness is never drawn implicitly: every triple, FSS key and re-sharing mask comes from a
`Dealer`, so a run is a bit-exact function of the dealer's stream (and can be replayed by the CPU
oracle in the tests).

Reference map (paths under syft/frameworks/torch/):
  encode / share / open      tensors/interpreters/precision.py:117-144, additive_shared.py:287-365
  beaver mul / matmul        mpc/spdz.py:21-197, mpc/beaver.py:7-63
  truncation                 precision.py:146-160, additive_shared.py:672-678
  fss comparison             mpc/fss.py:97-281, mpc/primitives.py:237-253
  relu                       additive_shared.py:922-925
  conv2d / pools / bn / lin  nn/functional.py:10-14, 44-75, 204-308, 460-525
  public add / sub / rsub    additive_shared.py:440-527 (constants are RE-SHARED with fresh randomness)
"""
import os

import torch

from . import _lib
from ._lib import call

I64 = torch.int64


def _empty_like(t):
    return torch.empty_like(t)


class Dealer:
    """The crypto provider: generates correlated randomness on the GPU.

    build_triple (mpc/beaver.py:7-63): a, b uniform int64, c = a∘b, each split into two shares.
    build_fss_keys (mpc/primitives.py:237-253): DIF keys + alpha additively split mod 2^32.

    Randomness is a ChaCha20 keystream (`primia_chacha20_fill`) under a 256-bit key taken from the operating
    system's entropy pool (`os.urandom`) when the dealer is constructed — never shared, never derived from anything
    public: the parties' privacy rests on not being able to predict these masks (the request schedule itself IS
    public).  `seed` is for tests, benchmarks and oracle replay only: it derives the key from the seed
    (SHA-256), making the whole stream reproducible — and therefore worthless as a secret."""

    def __init__(self, device, seed=None):
        import hashlib
        import os

        self.device = torch.device(device)
        self.seeded = seed is not None
        raw = os.urandom(40) if seed is None else hashlib.sha256(b"primia-dealer-debug-seed:%d" % int(seed)).digest() + bytes(8)
        self._key = [int.from_bytes(raw[8 * i:8 * i + 8], "little") for i in range(4)]
        self._nonce = int.from_bytes(raw[32:40], "little")
        self._block = 0
        self.log = None  # set to a list to record every primitive handed out (tests replay it)
        self.tape = None  # set to a list to keep every primitive ON THE DEVICE (offline phase, see PreloadedDealer)
        self.requests = None  # set to a list to record (method, args) of every request (GraphedSecureInference)

    def rand64(self, *shape):
        out = torch.empty(shape, dtype=I64, device=self.device)
        n = out.numel()
        call("primia_chacha20_fill", *self._key, self._nonce, self._block, out, n)
        self._block += (n + 7) // 8
        return out

    def _split(self, v):
        r = self.rand64(*v.shape)
        s1 = _empty_like(v)
        call("primia_ring_sub", v, r, s1, v.numel(), v.numel())
        return [r, s1]

    def triple(self, op, xshape, yshape):
        if self.requests is not None:
            self.requests.append(("triple", (op, tuple(xshape), tuple(yshape)), {}))
        # the input shares are drawn directly (a = a0 + a1 with both shares uniform IS the sharing of a uniform a that
        # build_triple produces by drawing a and one share, mpc/beaver.py:7-63): 9 launches per triple instead of 21
        sa = [self.rand64(*xshape), self.rand64(*xshape)]
        sb = [self.rand64(*yshape), self.rand64(*yshape)]
        c0 = self.rand64(*_triple_c_shape(op, xshape, yshape))
        c1 = _empty_like(c0)
        self.triple_c1(op, xshape, yshape, sa[0], sa[1], sb[0], sb[1], c0, c1)
        sc = [c0, c1]
        t = [(sa[j], sb[j], sc[j]) for j in range(2)]
        if self.log is not None:
            self.log.append(("triple", op, [tuple(x.cpu().numpy() for x in t[j]) for j in range(2)]))
        if self.tape is not None:
            self.tape.append(t)
        return t

    def triple_c1(self, op, xshape, yshape, a0, a1, b0, b1, c0, c1, scratch=None):
        """c1 = (a0 + a1) o (b0 + b1) - c0: the product share that completes a triple whose other five shares are uniform."""
        if op == "mul":
            # element-wise with the smaller operand broadcast over the leading dims
            if a0.numel() >= b0.numel():
                call("primia_triple_mul_c1", a0, a1, b0, b1, c0, c1, a0.numel(), b0.numel())
            else:
                call("primia_triple_mul_c1", b0, b1, a0, a1, c0, c1, b0.numel(), a0.numel())
        else:
            M, K, N = xshape[-2], xshape[-1], yshape[-1]
            if scratch is None:
                scratch = torch.empty(M * K + K * N, dtype=I64, device=self.device)
            call("primia_triple_matmul_c1", a0, a1, b0, b1, c0, c1, scratch, M, K, N)

    def dif_keys(self, n):
        if self.requests is not None:
            self.requests.append(("dif_keys", (n,), {}))
        dev = self.device
        # raw keystream words, then build_fss_keys' arithmetic in place (primia_fss_alpha_split): alpha and its mask r below
        # 2^32 (mpc/fss.py:346, mpc/primitives.py:249), word 0 of each seed 63 bits (randbit, fss.py:495-501), and
        # primitives.py:249-251: party 0 receives (alpha - mask) mod 2^32, party 1 the mask
        alpha = self.rand64(n)
        s0 = self.rand64(2, 2, n)
        r = self.rand64(n)
        a0 = torch.empty(n, dtype=I64, device=dev)
        call("primia_fss_alpha_split", alpha, s0, r, a0, n)
        bits = torch.empty(32, n, dtype=torch.uint8, device=dev)
        cw_sigma = torch.empty(32, 2, n, dtype=I64, device=dev)
        cw_s = torch.empty(32, 2, n, dtype=I64, device=dev)
        leaf = torch.empty(33, n, dtype=torch.int32, device=dev)
        call("primia_dif_keygen", alpha, s0, bits, cw_sigma, cw_s, leaf, n)
        keys = [dict(alpha=[a0, r][b], s0=s0[b], bits=bits, cw_sigma=cw_sigma, cw_s=cw_s, cw_leaf=leaf) for b in range(2)]
        if self.log is not None:
            self.log.append(("dif", n, alpha.cpu().numpy(), s0.cpu().numpy(), r.cpu().numpy()))
        if self.tape is not None:
            self.tape.append(keys)
        return keys

    def const_mask(self, *shape, owner=None):
        """The mask of a fresh sharing; `owner` (None = public value, else the party that holds the secret)
        only matters to a distributed dealer, which sends the mask to the owner alone."""
        if self.requests is not None:
            self.requests.append(("const_mask", tuple(shape), {"owner": owner}))
        r = self.rand64(*shape)
        if self.log is not None:
            self.log.append(("mask", r.cpu().numpy()))
        if self.tape is not None:
            self.tape.append(r)
        return r


class PreloadedDealer:
    """Online phase only: hands out primitives that a Dealer generated earlier (its `tape`), in
    the same order — the reference's pre-provisioned crypto store (mpc/primitives.py:161-235)."""

    def __init__(self, tape, device):
        self.tape, self.pos, self.device = tape, 0, torch.device(device)
        self.log = None

    def _next(self):
        v = self.tape[self.pos]
        self.pos += 1
        return v

    def triple(self, op, xshape, yshape):
        return self._next()

    def dif_keys(self, n):
        return self._next()

    def const_mask(self, *shape, owner=None):
        return self._next()


class LocalOpener:
    """Both parties' shares live on this GPU: open = one ring add."""

    def open(self, shares):
        out = _empty_like(shares[0])
        call("primia_ring_add", shares[0], shares[1], out, shares[0].numel(), shares[0].numel())
        return out


class DistOpener:
    """One party per rank (party j = rank j of the 2-rank party group, e.g. two GPUs over xGMI): open = an
    int64 all_reduce(SUM) of the local share, which wraps mod 2^64 — the 2-party exchange the reference routes
    through the orchestrator (mpc/spdz.py:162-176, mpc/fss.py:158-170)."""

    def __init__(self, group=None, party=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.party = dist.get_rank(group) if party is None else party

    def open(self, shares):
        mine = shares[self.party].clone()
        self.dist.all_reduce(mine, op=self.dist.ReduceOp.SUM, group=self.group)
        return mine


class SecureContext:
    def __init__(self, dealer, base=10, precision_fractional=16, opener=None, party=None, link=None):
        """party=None: both parties' shares live in this process (a share is a 2-list).  party=j: this process
        is party j of a distributed run — every 2-list carries only entry j (the other is None), `opener` must
        be a DistOpener and `link` a PartyLink for the owner-to-peer transfer of freshly shared secrets."""
        self.dealer = dealer
        self.base = base
        self.pf = precision_fractional
        self.scale = base ** precision_fractional
        self.opener = opener or LocalOpener()
        self.party = party
        self.parties = (0, 1) if party is None else (party,)
        self.link = link
        self.stats = {"beaver_mul": 0, "beaver_matmul": 0, "dif_evals": 0}
        self.fuse_newton = True
        self._newton_tables, self._newton_calls = [], 0     # see _reciprocal_newton_fused
        self._wt_cache = {}                                 # transposed weight shares of conv2d / linear (static per model)

    # Both parties hosted here: a layer's mask -> open -> combine -> truncate -> re-layout chain runs as one pass over the
    # shares of party 0 AND party 1 (csrc/secure_local.hip; an open is an addition of two values that sit side by side).
    # The dealer's primitives are requested in exactly the order of the step-by-step chain, results are bit-identical;
    # `local_fused = False` selects the chain (what a three-role run executes, where the opens are messages).
    local_fused = True

    @property
    def _local(self):
        return self.party is None and isinstance(self.opener, LocalOpener) and self.local_fused

    # ---- encode / share / reconstruct -----------------------------------------------------------
    def encode(self, x):
        q = torch.empty(x.shape, dtype=I64, device=x.device)
        call("primia_fx_encode", x.contiguous().to(torch.float32), q, x.numel(), float(self.scale))
        return q

    def decode(self, q):
        x = torch.empty(q.shape, dtype=torch.float32, device=q.device)
        call("primia_fx_decode", q, x, q.numel(), float(self.scale))
        return x

    def share(self, q, owner=None, shape=None):
        """share_secret (additive_shared.py:317-365): (r, q - r) with r from the crypto provider.
        `owner`: the party that knows q (0 = model owner, 1 = data owner) or None for a value both parties
        know.  In a distributed run the mask goes to the owner only (to both for a public value); the owner
        forms q - r and hands the peer ITS share; a non-owner passes q = None and `shape`."""
        shape = tuple(q.shape) if q is not None else tuple(shape)
        if self.party is None:
            r = self.dealer.const_mask(*shape, owner=owner)
            s1 = _empty_like(q)
            call("primia_ring_sub", q, r, s1, q.numel(), q.numel())
            return [r, s1]
        me, out = self.party, [None, None]
        if owner is None or owner == me:
            r = self.dealer.const_mask(*shape, owner=owner)
            s1 = _empty_like(r)
            call("primia_ring_sub", q, r, s1, r.numel(), r.numel())
            mine, theirs = (r, s1) if me == 0 else (s1, r)
            out[me] = mine
            if owner is not None:
                self.link.send_to_peer(theirs)
        else:
            out[me] = self.link.recv_from_peer(shape)
        return out

    def reconstruct(self, x):
        return self.opener.open(x)

    def _ref(self, x):
        """The share this process holds (shapes / devices are read from it)."""
        return x[self.parties[0]]

    def _each(self, fn):
        """[fn(j) for the parties hosted here], as a 2-list."""
        out = [None, None]
        for j in self.parties:
            out[j] = fn(j)
        return out

    # ---- local (per-share) ops -------------------------------------------------------------------
    def _ew(self, fn, a, b):
        if self._local and fn in ("primia_ring_add", "primia_ring_sub"):
            big, small = a[0], b[0]
            if small.numel() > big.numel():
                raise ValueError("second operand must not be larger")
            o = [_empty_like(big), _empty_like(big)]
            call("primia_ring_ew_2p", 0 if fn == "primia_ring_add" else 1, a[0], a[1], b[0], b[1], o[0], o[1], big.numel(),
                 small.numel())
            return o

        def one(j):
            big, small = (a[j], b[j])
            if small.numel() > big.numel():
                raise ValueError("second operand must not be larger")
            o = _empty_like(big)
            call(fn, big, small, o, big.numel(), small.numel())
            return o

        return self._each(one)

    def add(self, a, b):
        return self._ew("primia_ring_add", a, b)

    def sub(self, a, b):
        return self._ew("primia_ring_sub", a, b)

    def neg(self, a):
        def one(j):
            o = _empty_like(a[j])
            call("primia_ring_scale", a[j], -1, o, a[j].numel())
            return o

        return self._each(one)

    def trunc(self, a, d):
        def one(j):
            o = _empty_like(a[j])
            call("primia_trunc_div", a[j], int(d), o, a[j].numel())
            return o

        return self._each(one)

    def sub_public_scalar(self, a, value):
        """AST - int (additive_shared.py:453-484, 506-524): the constant becomes a FRESH random
        sharing of shape [1] that is subtracted share-wise (broadcast)."""
        return self.sub(a, self.share(self._const(int(value), self._ref(a).device)))

    def _const(self, value, device, n=1):
        """A public constant as a device tensor, uploaded once per (value, length) and kept (no fill kernel per use; a captured
        forward finds it in the cache its eager pass filled — an upload cannot be captured)."""
        cache = self.__dict__.setdefault("_consts", {})
        key = (value, n, str(device))
        if key not in cache:
            cache[key] = torch.full((n,), value, dtype=I64).to(device)
        return cache[key]

    # ---- Beaver -----------------------------------------------------------------------------------
    fuse_beaver = True

    def beaver_mul(self, x, y, trunc=None):
        """Element-wise private product; trunc = d: followed by each party's truncation of its share by d (fpt_mul).
        One operand may be a vector broadcast over the other's leading dims; the ring product is symmetric so the
        big one is taken first."""
        xr, yr = self._ref(x), self._ref(y)
        swap = xr.numel() < yr.numel()
        t = self.dealer.triple("mul", tuple(xr.shape), tuple(yr.shape))
        if swap:
            x, y, xr, yr = y, x, yr, xr
            t = [None if tj is None else (tj[1], tj[0], tj[2]) for tj in t]
        n, nb = xr.numel(), yr.numel()
        if self._local:
            out = [_empty_like(xr), _empty_like(xr)]
            call("primia_fpt_mul_local", x[0], x[1], y[0], y[1], t[0][0], t[0][1], t[0][2], t[1][0], t[1][1], t[1][2],
                 None, None, out[0], out[1], n, nb, 0 if trunc is None else int(trunc))
            self.stats["beaver_mul"] += 1
            return out
        d, e = [None, None], [None, None]
        for j in self.parties:  # spdz_mask
            d[j], e[j] = _empty_like(xr), _empty_like(yr)
            if self.fuse_beaver:
                call("primia_beaver_mask", x[j], t[j][0], d[j], n, y[j], t[j][1], e[j], nb)
            else:
                call("primia_ring_sub", x[j], t[j][0], d[j], n, n)
                call("primia_ring_sub", y[j], t[j][1], e[j], nb, nb)
        delta, eps = self.opener.open(d), self.opener.open(e)
        fused_trunc = trunc is not None and self.fuse_beaver

        def one(j):  # spdz_compute
            o = _empty_like(xr)
            if fused_trunc:
                call("primia_beaver_combine_mul_trunc", j, delta, eps, t[j][0], t[j][1], t[j][2], o, n, nb, int(trunc))
            else:
                call("primia_beaver_combine_mul", j, delta, eps, t[j][0], t[j][1], t[j][2], o, n, nb)
            return o

        self.stats["beaver_mul"] += 1
        out = self._each(one)
        return self.trunc(out, trunc) if (trunc is not None and not fused_trunc) else out

    def beaver_matmul(self, x, y):
        xr, yr = self._ref(x), self._ref(y)
        M, K = xr.shape[-2], xr.shape[-1]
        N = yr.shape[-1]
        t = self.dealer.triple("matmul", tuple(xr.shape), tuple(yr.shape))
        if self._local:
            out = [torch.empty(*xr.shape[:-1], N, dtype=I64, device=xr.device) for _ in range(2)]
            scratch = torch.empty(M * K + 2 * K * N, dtype=I64, device=xr.device)
            call("primia_beaver_matmul_local", x[0], x[1], y[0], y[1], t[0][0], t[0][1], t[0][2], t[1][0], t[1][1], t[1][2],
                 out[0], out[1], scratch, M, K, N)
            self.stats["beaver_matmul"] += 1
            return out
        d, e = [None, None], [None, None]
        for j in self.parties:
            d[j], e[j] = _empty_like(xr), _empty_like(yr)
            if self.fuse_beaver:
                call("primia_beaver_mask", x[j], t[j][0], d[j], xr.numel(), y[j], t[j][1], e[j], yr.numel())
            else:
                call("primia_ring_sub", x[j], t[j][0], d[j], xr.numel(), xr.numel())
                call("primia_ring_sub", y[j], t[j][1], e[j], yr.numel(), yr.numel())
        delta, eps = self.opener.open(d), self.opener.open(e)
        scratch = torch.empty(K * N, dtype=I64, device=xr.device)

        def one(j):
            o = torch.empty(*xr.shape[:-1], N, dtype=I64, device=xr.device)
            call("primia_beaver_combine_matmul", j, delta, eps, t[j][0], t[j][1], t[j][2], o, scratch, M, K, N)
            return o

        self.stats["beaver_matmul"] += 1
        return self._each(one)

    def fpt_mul(self, x, y):
        """FPT * FPT (precision.py:309-316, 356-358): Beaver mul, then per-share truncation."""
        return self.beaver_mul(x, y, trunc=self.scale)

    def fpt_matmul(self, x, y):
        """FPT @ FPT (precision.py:419-463)."""
        return self.trunc(self.beaver_matmul(x, y), self.scale)

    # ---- FSS comparison ------------------------------------------------------------------------------
    def _le_local(self, x1, x2, n, shape, cols1=(1, 0), cols2=(1, 0), length=1):
        """fss.le with both parties here: mask_builder, the open and both DIF evaluations in one launch.  x1 = None stands
        for shares of zero; colsK = (row width, first column) when operand K is a column range of a matrix."""
        keys = self.dealer.dif_keys(n)
        dev = x2[0].device
        out = [torch.empty(shape, dtype=I64, device=dev), torch.empty(shape, dtype=I64, device=dev)]
        k0, k1 = keys
        call("primia_dif_eval_local", None if x1 is None else x1[0], None if x1 is None else x1[1], cols1[0], cols1[1],
             x2[0], x2[1], cols2[0], cols2[1], length, k0["alpha"], k1["alpha"], k0["s0"], k1["s0"], k0["bits"],
             k0["cw_sigma"], k0["cw_s"], k0["cw_leaf"], out[0], out[1], n)
        self.stats["dif_evals"] += n
        return out

    def le(self, x1, x2):
        """fss.le(x1, x2) (mpc/fss.py:97-185, 279): int64 shares of the bit [x1 <= x2]."""
        xr = self._ref(x1)
        n = xr.numel()
        if self._local:
            return self._le_local(x1, x2, n, tuple(xr.shape))
        keys = self.dealer.dif_keys(n)
        r = [None, None]
        for j in self.parties:  # mask_builder
            r[j] = _empty_like(xr)
            call("primia_fss_mask", x1[j], x2[j], keys[j]["alpha"], r[j], n)
        masked = torch.empty(n, dtype=torch.int32, device=xr.device)
        if self.party is None:
            call("primia_fss_open", r[0], r[1], masked, n)
        else:  # the parties exchange their masked shares; the sum is taken mod 2^32 (fss.py:158-170)
            opened = self.opener.open(r)
            call("primia_fss_open", opened, self._const(0, opened.device, opened.numel()), masked, n)

        def one(j):  # evaluate
            o = _empty_like(xr)
            k = keys[j]
            call("primia_dif_eval", j, masked, k["s0"], k["bits"], k["cw_sigma"], k["cw_s"], k["cw_leaf"], o, n)
            return o

        self.stats["dif_evals"] += n
        return self._each(one)

    def relu(self, x):
        """AST.relu under fss (additive_shared.py:922-925): x * (x >= 0), (x >= 0) = le(x - x, x)."""
        if self._local:      # le(x - x, x): the first operand is a sharing of zero (no launch for it)
            xr = x[0]
            return self.beaver_mul(x, self._le_local(None, x, xr.numel(), tuple(xr.shape)))
        zero = self.sub(x, x)
        return self.beaver_mul(x, self.le(zero, x))

    def _max_pair(self, left, right):
        """left + (right >= left) * (right - left)  (nn/functional.py:494)."""
        bit = self.le(left, right)
        return self.add(left, self.beaver_mul(bit, self.sub(right, left)))

    def _max_pair_cols(self, left, wl, sl, right, wr, sr, rows, length):
        """_max_pair on column ranges (both parties here): comparison, Beaver product and the final add in two launches."""
        n = rows * length
        bit = self._le_local(left, right, n, (rows, length), (wl, sl), (wr, sr), length)
        t = self.dealer.triple("mul", (rows, length), (rows, length))
        dev = left[0].device
        out = [torch.empty(rows, length, dtype=I64, device=dev), torch.empty(rows, length, dtype=I64, device=dev)]
        call("primia_max_combine_local", bit[0], bit[1], left[0], left[1], wl, sl, right[0], right[1], wr, sr, t[0][0],
             t[0][1], t[0][2], t[1][0], t[1][1], t[1][2], out[0], out[1], rows, length)
        self.stats["beaver_mul"] += 1
        return out

    def _cols(self, x, rows, w, start, length):
        def one(j):
            o = torch.empty(rows, length, dtype=I64, device=x[j].device)
            call("primia_ring_slice_cols", x[j], o, rows, w, start, length)
            return o

        return self._each(one)

    # ---- layers (nn/functional.py) ------------------------------------------------------------------
    def conv2d(self, x, w, stride, padding):
        """conv2d (nn/functional.py:204-308): per-share im2col, Beaver matmul + truncation,
        per-share reshape.  x shares [1,C,H,W]; w shares [O,C,R,S]; no bias in ResNet convs."""
        B, C, H, W = self._ref(x).shape
        O, _, R, S = self._ref(w).shape
        Ho, Wo = (H + 2 * padding - R) // stride + 1, (W + 2 * padding - S) // stride + 1
        K = C * R * S
        if self._local:
            dev = x[0].device
            im = [torch.empty(B, Ho * Wo, K, dtype=I64, device=dev), torch.empty(B, Ho * Wo, K, dtype=I64, device=dev)]
            call("primia_im2col_syft_2p", x[0], x[1], im[0], im[1], B, C, H, W, R, S, stride, padding)
            wt = self._weight_t(w, O, K)
            res = self.beaver_matmul(im, wt)
            out = [torch.empty(B, O, Ho, Wo, dtype=I64, device=dev), torch.empty(B, O, Ho, Wo, dtype=I64, device=dev)]
            call("primia_trunc_col2out_2p", res[0], res[1], None, None, out[0], out[1], B, Ho * Wo, O, int(self.scale))
            return out
        im, wt = [None, None], [None, None]
        for j in self.parties:
            a = torch.empty(B, Ho * Wo, K, dtype=I64, device=x[j].device)
            call("primia_im2col_syft", x[j], a, B, C, H, W, R, S, stride, padding)
            im[j] = a
            # weight.reshape(O, -1).t(): [K, O]
            t = torch.empty(K, O, dtype=I64, device=x[j].device)
            call("primia_col2out_syft", w[j], None, t, 1, O, K)
            wt[j] = t
        res = self.fpt_matmul(im, wt)

        def one(j):
            o = torch.empty(B, O, Ho, Wo, dtype=I64, device=res[j].device)
            call("primia_col2out_syft", res[j], None, o, B, Ho * Wo, O)
            return o

        return self._each(one)

    _wt_cache_max = 64      # a ResNet-18 has 21 shared weight matrices: room for three models on one context

    def _weight_t(self, w, O, K):
        """weight.reshape(O, -1).t() of both parties' (static) weight shares, formed once per model.

        The cache is keyed on the share TENSORS, which it keeps alive (so an address the caching allocator hands out again
        can never alias an entry) and compares by identity together with their in-place version counters.  torch bumps a
        version counter for ITS in-place writes only: a caller that rewrites shares in place through the C ABI (a kernel
        launched on `data_ptr()` — nothing in this module does that to model shares) must call `invalidate_weight_cache()`,
        as `SecureResNet18.__init__` does when a model is (re-)shared.  At most `_wt_cache_max` entries, least recently used
        evicted: a long-lived context fed changing weight objects does not grow."""
        key = (id(w[0]), id(w[1]), O, K)
        hit = self._wt_cache.get(key)
        if hit is not None:
            w0, w1, v0, v1, wt = hit
            if w0 is w[0] and w1 is w[1] and v0 == w[0]._version and v1 == w[1]._version:
                self._wt_cache[key] = self._wt_cache.pop(key)      # most recently used: to the back
                return wt
            del self._wt_cache[key]
        wt = []
        for j in (0, 1):
            t = torch.empty(K, O, dtype=I64, device=w[j].device)
            call("primia_col2out_syft", w[j], None, t, 1, O, K)
            wt.append(t)
        while len(self._wt_cache) >= self._wt_cache_max:
            del self._wt_cache[next(iter(self._wt_cache))]
        self._wt_cache[key] = (w[0], w[1], w[0]._version, w[1]._version, wt)
        return wt

    def invalidate_weight_cache(self):
        """Forget every cached transposed weight share (call after re-sharing or editing a model's weights)."""
        self._wt_cache.clear()

    def reciprocal_newton(self, v):
        """FPT.reciprocal(method="newton") (precision.py:507-518), C = 20, 80 iterations.

        With both parties in this process (the default deployment) the whole iteration is ONE launch
        (`primia_newton_reciprocal_local`): the primitives are requested from the dealer in exactly the order the
        step-by-step chain below consumes them and handed to the kernel as a pointer table, so the two forms are
        bit-identical — the chain is what a three-role run (opens = messages) executes, and `fuse_newton = False`
        selects it here too."""
        if self.party is None and isinstance(self.opener, LocalOpener) and self.fuse_newton:
            return self._reciprocal_newton_fused(v)
        C = 20
        # x0 = (C + 1 - v) / C
        y = self.neg(self.sub_public_scalar(v, (C + 1) * self.scale))
        x = self.trunc(y, C)
        for _ in range(79):
            xx = self.fpt_mul(x, x)
            vxx = self.fpt_mul(v, xx)
            y = self.neg(self.sub_public_scalar(vxx, (C + 1) * self.scale))
            x = self.trunc(self.fpt_mul(y, x), C)
        return x

    def _reciprocal_newton_fused(self, v):
        shape = tuple(v[0].shape)
        n = v[0].numel()
        dev = v[0].device
        prims = [self.dealer.const_mask(1, owner=None)]
        for _ in range(79):
            for k in range(3):
                if k == 2:
                    m = self.dealer.const_mask(1, owner=None)
                    prims.append(m)
                t = self.dealer.triple("mul", shape, shape)
                prims += [t[0][0], t[0][1], t[0][2], t[1][0], t[1][1], t[1][2]]
        ptrs = [p.data_ptr() for p in prims]
        if isinstance(self.dealer, PreloadedDealer):
            # static primitive buffers (pre-provisioned store, GraphedSecureInference): the device pointer table is
            # cached per call position, so a replayed / captured forward uploads nothing (a host-to-device copy
            # cannot be captured into a hipGraph).  Nothing is retained beyond the table: the tape owns the buffers.
            idx = self._newton_calls
            self._newton_calls += 1
            if idx < len(self._newton_tables) and self._newton_tables[idx][0] == ptrs:
                table = self._newton_tables[idx][1]
            else:
                table = torch.tensor(ptrs, dtype=I64).to(dev)
                if idx < len(self._newton_tables):
                    self._newton_tables[idx] = (ptrs, table)
                else:
                    self._newton_tables.append((ptrs, table))
        else:
            # live dealer: fresh primitives every call — build the table, launch, drop every reference.  The launch
            # is on the current stream and torch's allocator is stream-ordered, so the 237 triples may be recycled
            # as soon as this function returns (round 2 pinned all of them per image: ~55 MB leaked per forward).
            table = torch.tensor(ptrs, dtype=I64).to(dev)
        out = [torch.empty(shape, dtype=I64, device=dev), torch.empty(shape, dtype=I64, device=dev)]
        call("primia_newton_reciprocal_local", v[0].contiguous(), v[1].contiguous(), table, int(self.scale), out[0], out[1], n)
        self.stats["beaver_mul"] += 3 * 79
        return out

    def batch_norm_eval(self, x, mean, var, weight, bias, inv=None):
        """batch_norm in eval mode (nn/functional.py:44-75): ((x - mean) * newton(var)) * w + b on
        [H*W, C] rows — no explicit sqrt and no eps, exactly as the reference computes it (its
        "newton" iteration converges to var^-1/2).  `inv` may carry shares of newton(var) computed
        earlier (see SecureResNet18.precompute_inv)."""
        B, C, H, W = self._ref(x).shape
        if B != 1:
            raise ValueError("encrypted inference runs one image at a time (inference.py:292)")

        def to_rows(j):  # permute(1,0,2,3).reshape(C,-1).t()  -> [B*H*W, C]  (B == 1)
            o = torch.empty(B * H * W, C, dtype=I64, device=x[j].device)
            call("primia_col2out_syft", x[j], None, o, 1, C, B * H * W)
            return o

        if self._local:
            if inv is None:
                inv = self.reciprocal_newton(var)
            import ctypes

            HW = B * H * W
            t1 = self.dealer.triple("mul", (C,), (HW, C))           # fpt_mul(inv, rows - mean)
            t2 = self.dealer.triple("mul", (HW, C), (C,))           # fpt_mul(normalized, weight)
            arr = lambda t: (ctypes.c_void_p * 6)(*[q.data_ptr() for q in (t[0][0], t[0][1], t[0][2], t[1][0], t[1][1], t[1][2])])
            dev = x[0].device
            out = [torch.empty(B, C, H, W, dtype=I64, device=dev), torch.empty(B, C, H, W, dtype=I64, device=dev)]
            call("primia_bn_eval_local", x[0], x[1], mean[0], mean[1], inv[0], inv[1], weight[0], weight[1], bias[0], bias[1],
                 arr(t1), arr(t2), out[0], out[1], C, HW, int(self.scale))
            self.stats["beaver_mul"] += 2
            return out
        rows = self._each(to_rows)
        if inv is None:
            inv = self.reciprocal_newton(var)
        normalized = self.fpt_mul(inv, self.sub(rows, mean))
        result = self.add(self.fpt_mul(normalized, weight), bias)

        def back(j):
            o = torch.empty(B, C, H, W, dtype=I64, device=result[j].device)
            call("primia_col2out_syft", result[j], None, o, 1, B * H * W, C)
            return o

        return self._each(back)

    def max_pool2d_3x3s2(self, x):
        """_pool2d(mode="max") for a 3x3 window (nn/functional.py:460-508): unroll to 9 columns,
        binary tree on the first 8, then against the 9th."""
        B, C, H, W = self._ref(x).shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        rows = B * C * Ho * Wo

        def unroll(j):
            o = torch.empty(rows, 9, dtype=I64, device=x[j].device)
            call("primia_pool_unroll_syft", x[j], o, B, C, H, W, 3, 2, 1)
            return o

        if self._local:
            dev = x[0].device
            im = [torch.empty(rows, 9, dtype=I64, device=dev), torch.empty(rows, 9, dtype=I64, device=dev)]
            call("primia_pool_unroll_syft_2p", x[0], x[1], im[0], im[1], B, C, H, W, 3, 2, 1)
            res = self._max_pair_cols(im, 9, 0, im, 9, 4, rows, 4)
            res = self._max_pair_cols(res, 4, 0, res, 4, 2, rows, 2)
            left = self._max_pair_cols(res, 2, 0, res, 2, 1, rows, 1)
            res = self._max_pair_cols(left, 1, 0, im, 9, 8, rows, 1)
            return [r.view(B, C, Ho, Wo) for r in res]
        im = self._each(unroll)
        res = self._max_pair(self._cols(im, rows, 9, 0, 4), self._cols(im, rows, 9, 4, 4))
        res = self._max_pair(self._cols(res, rows, 4, 0, 2), self._cols(res, rows, 4, 2, 2))
        left = self._max_pair(self._cols(res, rows, 2, 0, 1), self._cols(res, rows, 2, 1, 1))
        res = self._max_pair(left, self._cols(im, rows, 9, 8, 1))
        return [None if r is None else r.view(B, C, Ho, Wo) for r in res]

    def avg_pool2d(self, x, k):
        """_pool2d(mode="avg"), stride = kernel (nn.AvgPool2d(k)): per-share window sum, then the
        per-share truncating division of AST.mean (additive_shared.py:719-729)."""
        B, C, H, W = self._ref(x).shape
        Ho, Wo = (H - k) // k + 1, (W - k) // k + 1
        rows = B * C * Ho * Wo

        def one(j):
            im = torch.empty(rows, k * k, dtype=I64, device=x[j].device)
            call("primia_pool_unroll_syft", x[j], im, B, C, H, W, k, k, 0)
            s = torch.empty(rows, dtype=I64, device=x[j].device)
            call("primia_ring_rowsum", im, s, rows, k * k)
            o = _empty_like(s)
            call("primia_trunc_div", s, k * k, o, rows)
            return o.view(B, C, Ho, Wo)

        return self._each(one)

    def linear(self, x, w, b):
        """F.linear -> torch.addmm(bias, input, weight.t()) -> FPT.addmm (nn/functional.py:10-14,
        precision.py:822-827): matmul + truncation, then + bias."""
        O, I = self._ref(w).shape

        def tr(j):
            t = torch.empty(I, O, dtype=I64, device=w[j].device)
            call("primia_col2out_syft", w[j], None, t, 1, O, I)
            return t

        if self._local:
            return self.add(self.fpt_matmul(x, self._weight_t(w, O, I)), b)
        return self.add(self.fpt_matmul(x, self._each(tr)), b)


def share_order(keys):
    """model.fix_precision().share() walks `parameters()` and then `buffers()` (hook.py:624-632,738-765)."""
    keys = [k for k in keys if not k.endswith("num_batches_tracked")]
    buf = [k for k in keys if k.endswith("running_mean") or k.endswith("running_var")]
    return [k for k in keys if k not in buf] + buf


class SecureResNet18:
    """ResNet-18 forward on secret shares — `model.fix_precision().share()` followed by
    `model(data)` in inference.py:279-321, including the stem swap `model.pool, model.relu =
    model.relu, model.pool` (:289): conv1 -> bn1 -> MAXPOOL -> RELU."""

    def __init__(self, ctx: SecureContext, state_dict, input_size=224, blocks=None, batched_newton=True):
        """batched_newton=False consumes the provider's primitives in exactly the reference's order (newton(running_var)
        inside every batch_norm call, nn/functional.py:62-69) — the mode the reference-minted fixtures pin; the
        default hoists the image-independent iterations of all BatchNorm layers into one batched vector."""
        self.ctx = ctx
        self.input_size = input_size
        self.batched_newton = batched_newton
        dev = ctx.dealer.device
        self.p = {}
        ctx.invalidate_weight_cache()      # (a model shared on this context before: its transposed shares go with it)
        # hook.py:624-632,738-765: every parameter is encoded and shared, THEN every buffer (`parameters()` before
        # `buffers()`); the 0-dim num_batches_tracked buffers draw no randomness (generate_shares sizes its random
        # share with LongTensor(torch.Size([])) = an empty tensor, additive_shared.py:352) and are never read.
        for k in share_order(state_dict.keys()):
            v = state_dict[k]
            if ctx.party in (None, 0):  # the model owner is party 0 (inference.py:279-283)
                self.p[k] = ctx.share(ctx.encode(v.to(dev)), owner=0)
            else:
                self.p[k] = ctx.share(None, owner=0, shape=v.shape)
        self.blocks = blocks if blocks is not None else [
            (f"layer{li}.{bi}", (2 if (li > 1 and bi == 0) else 1)) for li in range(1, 5) for bi in range(2)]

    def bn_prefixes(self):
        out = ["bn1"]
        for prefix, _ in self.blocks:
            out += [prefix + ".bn1", prefix + ".bn2"]
            if (prefix + ".downsample.0.weight") in self.p:
                out.append(prefix + ".downsample.1")
        return out

    def precompute_inv(self):
        """newton(running_var) of every BatchNorm layer in ONE batched 80-step iteration.

        The reference recomputes it layer by layer inside each forward (nn/functional.py:62-69);
        it does not depend on the image, and the iteration is element-wise, so running all layers'
        channels as one vector performs exactly the same arithmetic per channel — it only turns
        20 x 80 x 3 tiny Beaver rounds into 80 x 3 (the per-element randomness is whatever slice of
        the batched triple that channel receives)."""
        c = self.ctx
        names = self.bn_prefixes()
        var = self._var_cat()
        inv = c.reciprocal_newton(var)
        out, off = {}, 0
        for n in names:
            k = c._ref(self.p[n + ".running_var"]).numel()
            out[n] = c._each(lambda j: inv[j][off:off + k].contiguous())
            off += k
        return out

    def _var_cat(self):
        """Every BatchNorm layer's running_var shares side by side (static per model: formed once, by the library's own copy —
        a ring scale by 1 — not by torch.cat)."""
        if getattr(self, "_var_cat_cache", None) is None:
            c, names = self.ctx, self.bn_prefixes()
            total = sum(c._ref(self.p[n + ".running_var"]).numel() for n in names)

            def one(j):
                out = torch.empty(total, dtype=I64, device=self.p[names[0] + ".running_var"][j].device)
                off = 0
                for n in names:
                    v = self.p[n + ".running_var"][j]
                    call("primia_ring_scale", v, 1, out[off:off + v.numel()], v.numel())
                    off += v.numel()
                return out

            self._var_cat_cache = c._each(one)
        return self._var_cat_cache

    def _bn(self, x, prefix):
        p = self.p
        return self.ctx.batch_norm_eval(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                                        p[prefix + ".weight"], p[prefix + ".bias"], inv=self._inv[prefix])

    def forward_shares(self, x):
        c, p = self.ctx, self.p
        self._inv = self.precompute_inv() if self.batched_newton else {n: None for n in self.bn_prefixes()}
        x = c.conv2d(x, p["conv1.weight"], 2, 3)
        x = self._bn(x, "bn1")
        x = c.max_pool2d_3x3s2(x)      # swapped stem (inference.py:289)
        x = c.relu(x)
        for prefix, stride in self.blocks:
            identity = x
            out = c.conv2d(x, p[prefix + ".conv1.weight"], stride, 1)
            out = c.relu(self._bn(out, prefix + ".bn1"))
            out = c.conv2d(out, p[prefix + ".conv2.weight"], 1, 1)
            out = self._bn(out, prefix + ".bn2")
            if (prefix + ".downsample.0.weight") in p:
                identity = c.conv2d(x, p[prefix + ".downsample.0.weight"], stride, 0)
                identity = self._bn(identity, prefix + ".downsample.1")
            x = c.relu(c.add(out, identity))
        k = c._ref(x).shape[-1]
        x = c.avg_pool2d(x, k)
        x = [None if t is None else t.reshape(1, -1) for t in x]
        return c.linear(x, p["fc.weight"], p["fc.bias"])

    def __call__(self, image):
        """image: fp32 [1, C, S, S] on the GPU -> decoded fp32 logits [1, classes].  The data owner is
        party 1 (inference.py:292-300); in a distributed run party 0 passes image = None."""
        c = self.ctx
        if c.party in (None, 1):
            xs = c.share(c.encode(image), owner=1)
        else:
            cin = c._ref(self.p["conv1.weight"]).shape[1]        # 3 (pretrained) or 1 (train.py:262)
            xs = c.share(None, owner=1, shape=(1, cin, self.input_size, self.input_size))
        out = self.forward_shares(xs)
        return c.decode(c.reconstruct(out))


class GraphedSecureInference:
    """Serving form of the encrypted forward: the online phase (about 6,500 small launches per image) is
    captured ONCE as a hipGraph over static buffers — the input image and every correlated-randomness
    primitive — and replayed per image; `refill()` has the dealer regenerate all per-image primitives into
    the same buffers (the reference's pre-provisioned crypto store, mpc/primitives.py:161-235, refilled
    between requests).  Results are bit-identical to the eager SecureResNet18 fed the same primitives.

    The refill is ONE graph launch too (round 6).  Every uniformly random word of an image's primitives — triple input
    shares and c0, re-sharing masks, FSS alpha / seeds / alpha masks — lives in ONE int64 arena that the tape entries are
    views of: a single ChaCha20 launch fills it (`primia_chacha20_fill_ctr`: the block counter is a DEVICE word the graph's
    last node advances, so every replay draws fresh keystream — a host counter would be frozen into the graph and every
    image would get the same masks), then one launch per triple forms c1 = a o b - c0 and two per comparison batch form
    the key (alpha split, DIF keygen).  ~3,000 eager launches per image became ~450 graph nodes; `refill_graph = False`
    launches the same calls eagerly (same bits).  The constructor ends with one refill: no image is ever served on the
    primitives the all-zero warm-up image consumed (opened values on a known input reveal the masks)."""

    refill_graph = True

    def __init__(self, state_dict, device, input_size=224, precision_fractional=16, base=10, seed=None, blocks=None):
        self.device = torch.device(device)
        # (an all-zero warm-up image, uploaded: no fill kernel of torch's on the path)
        self.image = torch.zeros(1, state_dict["conv1.weight"].shape[1], input_size, input_size, dtype=torch.float32).to(self.device)
        self.dealer = Dealer(self.device, seed)
        self.dealer.tape, self.dealer.requests = [], []
        ctx = SecureContext(self.dealer, base, precision_fractional)
        model = SecureResNet18(ctx, state_dict, input_size, blocks)
        self._n_model = len(self.dealer.tape)          # primitives consumed by sharing the model (kept)
        model(self.image)                              # offline pass: fills the tape, warms every kernel
        self.tape, self.requests = self.dealer.tape, self.dealer.requests
        self.dealer.tape = self.dealer.requests = None
        self.stats = dict(ctx.stats)
        self._rehome_tape()                            # per-image random words -> one arena (before any pointer is captured)
        pre = PreloadedDealer(self.tape, self.device)
        self._ctx = SecureContext(pre, base, precision_fractional)
        self._model = SecureResNet18(self._ctx, state_dict, input_size, blocks)   # re-shares with the same masks
        self._model(self.image)                        # eager pass over the static buffers: builds the pointer tables
        pre.pos, self._ctx._newton_calls = self._n_model, 0
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.out = self._model(self.image)
        torch.cuda.current_stream().wait_stream(side)
        # the provider's block counter moves to the device, behind everything the offline pass drew
        self._ctr = torch.tensor([self.dealer._block], dtype=I64).to(self.device)      # (uploaded: no fill kernel)
        self.dealer._block = None                      # (the host counter is dead from here on: rand64 would raise)
        self.refills = 0
        self._refill_launches()                        # eager: warms the dealer kernels AND replaces the warm-up primitives
        self.refills = 1
        self._refill_g = None
        if self.refill_graph:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._refill_g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._refill_g, stream=side):
                    self._refill_launches()
            torch.cuda.current_stream().wait_stream(side)

    def _rehome_tape(self):
        """Move every uniformly random tensor of the per-image primitives into ONE int64 arena (64-byte aligned slices, current
        values kept) and record, per primitive, the launches that complete it after a fill."""
        dev = self.device
        al = lambda n: (n + 7) // 8 * 8
        total, plan = 0, []
        for i in range(self._n_model, len(self.tape)):
            kind, args, _ = self.requests[i]
            e = self.tape[i]
            if kind == "const_mask":
                sizes = [e.numel()]
            elif kind == "triple":
                (a0, b0, c0), (a1, b1, _c1) = e
                sizes = [a0.numel(), a1.numel(), b0.numel(), b1.numel(), c0.numel()]
            elif kind == "dif_keys":
                sizes = [args[0], 4 * args[0], args[0]]        # raw alpha, both parties' seeds, alpha's mask
            else:
                raise _lib.PrimiaError(f"unknown primitive kind {kind!r}")
            offs = []
            for n in sizes:
                offs.append(total)
                total += al(n)
            plan.append(offs)
        self._arena = torch.empty(max(total, 8), dtype=I64, device=dev)

        def view(off, src=None, shape=None):
            shape = tuple(src.shape) if src is not None else shape
            n = 1
            for d in shape:
                n *= d
            v = self._arena[off:off + n].view(shape)
            if src is not None:
                v.copy_(src)
            return v

        self._ops, max_kn = [], 1
        for i, offs in zip(range(self._n_model, len(self.tape)), plan):
            kind, args, _ = self.requests[i]
            e = self.tape[i]
            if kind == "const_mask":
                self.tape[i] = view(offs[0], e)
            elif kind == "triple":
                op, xshape, yshape = args
                (a0, b0, c0), (a1, b1, c1) = e
                a0, a1, b0, b1, c0 = (view(o, t) for o, t in zip(offs, (a0, a1, b0, b1, c0)))
                self.tape[i] = [(a0, b0, c0), (a1, b1, c1)]
                self._ops.append(("triple", op, xshape, yshape, a0, a1, b0, b1, c0, c1))
                if op != "mul":
                    max_kn = max(max_kn, xshape[-2] * xshape[-1] + xshape[-1] * yshape[-1])
            else:
                (n,) = args
                k0, k1 = e
                alpha = view(offs[0], shape=(n,))
                s0 = view(offs[1], shape=(2, 2, n))
                s0[0].copy_(k0["s0"])
                s0[1].copy_(k1["s0"])
                r = view(offs[2], k1["alpha"])
                k0["s0"], k1["s0"], k1["alpha"] = s0[0], s0[1], r
                self._ops.append(("dif", n, alpha, s0, r, k0["alpha"], k0["bits"], k0["cw_sigma"], k0["cw_s"], k0["cw_leaf"]))
        self._mm_scratch = torch.empty(max_kn, dtype=I64, device=dev)

    def _refill_launches(self):
        d = self.dealer
        n = self._arena.numel()
        call("primia_chacha20_fill_ctr", *d._key, d._nonce, self._ctr, 0, self._arena, n)
        for op in self._ops:
            if op[0] == "triple":
                _, kind, xshape, yshape, a0, a1, b0, b1, c0, c1 = op
                d.triple_c1(kind, xshape, yshape, a0, a1, b0, b1, c0, c1, scratch=self._mm_scratch)
            else:
                _, m, alpha, s0, r, a0, bits, cw_sigma, cw_s, leaf = op
                call("primia_fss_alpha_split", alpha, s0, r, a0, m)
                call("primia_dif_keygen", alpha, s0, bits, cw_sigma, cw_s, leaf, m)
        call("primia_u64_add", self._ctr, (n + 7) // 8)

    def refill(self):
        """Fresh per-image primitives from the dealer, written into the captured buffers (on the current stream)."""
        if self._refill_g is not None:
            self._refill_g.replay()
        else:
            self._refill_launches()
        self.refills += 1

    def __call__(self, image, refill=True):
        """refill=False replays on the primitives the buffers hold (bit-identity checks against an eager forward on the
        same tape; a deployment never serves two images on one set of primitives)."""
        if refill:
            self.refill()
        self.image.copy_(image)
        self.graph.replay()
        return self.out


class PipelinedSecureInference:
    """A stream of encrypted inferences with the crypto provider HIDDEN behind the online phase.

    The reference provisions primitives on demand, serially with the protocol (mpc/fss.py:142-146, primitives.py:161-235).
    Here two `GraphedSecureInference` slots (each its own dealer, static primitive buffers and captured online graph)
    alternate: while image i replays slot i % 2 on the caller's stream, the dealer refills the OTHER slot for image i + 1
    on its own stream — one graph launch of its own (round 6; ~3,000 eager launches before) whose 21 key generations share the
    vector ALUs with the online evaluation.  Events order a slot's refill after the replay that consumed it and the next replay after
    the refill.  Every image's shares are those of an eager SecureResNet18 on the slot's primitives (the graph is
    bit-identical to it, tests/test_gpu_secure.py); nothing about the protocol changes, only who waits for whom.
    In the three-role deployment the same overlap is physical: the dealer rank runs ahead of the parties on its own GPU."""

    def __init__(self, state_dict, device, input_size=224, precision_fractional=16, base=10, seed=None, blocks=None, slots=2):
        self.device = torch.device(device)
        self.slots = [GraphedSecureInference(state_dict, device, input_size, precision_fractional, base,
                                             None if seed is None else seed + 7919 * k, blocks) for k in range(slots)]
        self.stats = self.slots[0].stats
        self.dealer_stream = torch.cuda.Stream(device=self.device)
        # event: slot k's primitives are fresh.  None = fresh since construction: GraphedSecureInference ends its constructor
        # with a refill, so the first image of a slot never runs on the primitives the all-zero warm-up image consumed
        assert all(sl.refills >= 1 for sl in self.slots)
        self._ready = [None] * slots
        self._n = 0

    def __call__(self, image):
        """One encrypted inference; returns what SecureResNet18 returns (a copy: the slot's buffers are reused)."""
        k = self._n % len(self.slots)
        slot = self.slots[k]
        main = torch.cuda.current_stream()
        if self._ready[k] is not None:
            main.wait_event(self._ready[k])
        slot.image.copy_(image)
        slot.graph.replay()
        out = slot.out.clone()
        consumed = torch.cuda.Event()
        consumed.record(main)
        # the dealer regenerates this slot's primitives for image n + len(slots) while the next image runs on another slot
        with torch.cuda.stream(self.dealer_stream):
            self.dealer_stream.wait_event(consumed)
            slot.refill()
            ev = torch.cuda.Event()
            ev.record(self.dealer_stream)
        self._ready[k] = ev
        self._n += 1
        return out


# ---- three-role deployment: party 0 (model owner), party 1 (data owner), crypto provider --------------------
# inference.py:262-321 runs the same three roles as websocket workers (configs/websetting/config_inference.csv);
# here each role is one rank of a torch.distributed job — one GPU each over xGMI with the RCCL backend — and the
# only traffic is what the protocol itself exchanges: dealer -> party primitives, owner -> peer fresh shares, and
# the 2-party opens of DistOpener.

DEALER_RANK = 2


class PartyLink:
    """Point-to-point transport between the three roles, expressed as broadcasts inside 2-rank groups so the
    same code runs on RCCL (device buffers over xGMI) and on gloo (tests).  Every rank of the job constructs
    it (group creation is collective)."""

    def __init__(self, device, ranks=(0, 1, 2)):
        import torch.distributed as dist

        self.dist, self.device = dist, torch.device(device)
        self.rank = dist.get_rank()
        p0, p1, d = ranks
        self.ranks, self.dealer_rank = (p0, p1), d
        self.role = {p0: 0, p1: 1, d: "dealer"}[self.rank]
        self.parties_group = dist.new_group([p0, p1])
        self.dealer_groups = [dist.new_group(sorted([p0, d])), dist.new_group(sorted([p1, d]))]
        self.all_group = dist.new_group(sorted(ranks))

    # party <-> party
    def send_to_peer(self, t):
        self.dist.broadcast(t.contiguous(), src=self.rank, group=self.parties_group)

    def recv_from_peer(self, shape, dtype=I64):
        t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        self.dist.broadcast(t, src=self.ranks[1 - self.role], group=self.parties_group)
        return t

    # dealer -> party (to = 0 / 1) or dealer -> both (to = None)
    def _dealer_group(self, to):
        return self.all_group if to is None else self.dealer_groups[to]

    def dealer_send(self, t, to=None):
        self.dist.broadcast(t.contiguous(), src=self.dealer_rank, group=self._dealer_group(to))

    def from_dealer(self, shape, dtype=I64, private=True):
        t = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        self.dist.broadcast(t, src=self.dealer_rank, group=self._dealer_group(self.role if private else None))
        return t


def _triple_c_shape(op, xshape, yshape):
    if op == "mul":
        nx = int(torch.Size(xshape).numel())
        return tuple(xshape) if nx >= int(torch.Size(yshape).numel()) else tuple(yshape)
    return tuple(xshape[:-1]) + (yshape[-1],)


class PartyDealer:
    """What party j sees of the crypto provider: it receives ITS half of each primitive, in protocol order
    (mpc/primitives.py:161-235 keeps the same per-worker store; here nothing is requested — the provider
    follows the public request schedule of the network, see request_schedule)."""

    def __init__(self, link: PartyLink):
        self.link, self.device, self.j = link, link.device, link.role
        self.log = None

    def _mine(self, v):
        out = [None, None]
        out[self.j] = v
        return out

    def triple(self, op, xshape, yshape):
        f = self.link.from_dealer
        return self._mine((f(xshape), f(yshape), f(_triple_c_shape(op, xshape, yshape))))

    def dif_keys(self, n):
        f = self.link.from_dealer
        k = dict(alpha=f((n,)), s0=f((2, n)))
        k["bits"] = f((32, n), torch.uint8, private=False)
        k["cw_sigma"] = f((32, 2, n), I64, private=False)
        k["cw_s"] = f((32, 2, n), I64, private=False)
        k["cw_leaf"] = f((33, n), torch.int32, private=False)
        return self._mine(k)

    def const_mask(self, *shape, owner=None):
        return self.link.from_dealer(shape, private=owner is not None)


def architecture_of(state_dict):
    """name -> shape of every shared tensor: all a non-owner needs to know about the model."""
    return {k: tuple(v.shape) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}


def request_schedule(arch, input_size, device, blocks=None, precision_fractional=16, base=10):
    """The (public) sequence of primitives one model sharing + one encrypted forward consumes, as
    (model_requests, image_requests): it depends on the architecture and the input size only, so the crypto
    provider derives it from a dry run on a dummy model of that architecture."""
    d = Dealer(device, seed=0)
    d.requests = []
    ctx = SecureContext(d, base, precision_fractional)
    dummy = {k: torch.ones(shape, dtype=torch.float32) for k, shape in arch.items()}
    model = SecureResNet18(ctx, dummy, input_size, blocks)
    n_model = len(d.requests)
    model(torch.zeros(1, arch["conv1.weight"][1], input_size, input_size, dtype=torch.float32, device=device))
    return d.requests[:n_model], d.requests[n_model:]


class DealerService:
    """The crypto provider's rank: generate each scheduled primitive on its GPU and ship each half to its
    party (build_triple / build_fss_keys, mpc/beaver.py:7-63, mpc/primitives.py:237-253)."""

    def __init__(self, dealer: Dealer, link: PartyLink):
        self.dealer, self.link = dealer, link

    def serve(self, requests):
        send = self.link.dealer_send
        for kind, args, kw in requests:
            v = getattr(self.dealer, kind)(*args, **kw)
            if kind == "triple":
                for j in range(2):
                    for t in v[j]:
                        send(t, j)
            elif kind == "dif_keys":
                for j in range(2):
                    send(v[j]["alpha"], j)
                    send(v[j]["s0"], j)
                for name in ("bits", "cw_sigma", "cw_s", "cw_leaf"):
                    send(v[0][name], None)
            else:
                send(v, kw.get("owner"))


def party_context(link: PartyLink, precision_fractional=16, base=10):
    """SecureContext of the party this rank plays."""
    return SecureContext(PartyDealer(link), base, precision_fractional,
                         opener=DistOpener(link.parties_group, link.role), party=link.role, link=link)


def run_three_role(link: PartyLink, arch, input_size, n_images, state_dict=None, images=None, seed=None, blocks=None,
                   precision_fractional=16, base=10):
    """One rank's part of the three-role encrypted inference of inference.py:279-321.
    Party 0 passes `state_dict`, party 1 passes `images` (fp32 [n,3,S,S] on its GPU), the dealer neither;
    all know the architecture, the input size and how many images will be classified.  Parties return the
    list of decoded logits (the reference `.get()`s the prediction to the orchestrator), the dealer None."""
    if link.role == "dealer":
        model_req, image_req = request_schedule(arch, input_size, link.device, blocks, precision_fractional, base)
        svc = DealerService(Dealer(link.device, seed), link)
        svc.serve(model_req)
        for _ in range(n_images):
            svc.serve(image_req)
        return None
    ctx = party_context(link, precision_fractional, base)
    if link.role == 0:
        if state_dict is None:
            raise ValueError("party 0 is the model owner: it needs the state dict")
        shapes = {k: v for k, v in state_dict.items()}
    else:
        if images is None:
            raise ValueError("party 1 is the data owner: it needs the images")
        shapes = {k: torch.empty(shape, device="meta") for k, shape in arch.items()}
    model = SecureResNet18(ctx, shapes, input_size, blocks)
    out = []
    for i in range(n_images):
        out.append(model(images[i:i + 1] if link.role == 1 else None))
    return out
