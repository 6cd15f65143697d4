"""Test infrastructure: the crypto provider's randomness exactly as the REFERENCE draws it.

PySyft's provider takes its randomness from the process-wide generators: torch's CPU generator for Beaver
triples and fresh sharings, NumPy's legacy `np.random` for FSS keys.  Seeding both therefore fixes the whole
primitive stream of an encrypted forward; a full-width ResNet-18 stream is ~0.5 GB, so the committed fixture
(`tests/golden/secure_ref_forward.npz`, minted by `tests/golden/make_secure_ref_golden.py` by RUNNING the
reference's own code) holds the seeds, one checksum per primitive and the output shares, and this module
re-draws the stream on demand — the same calls, shapes and order as

    mpc/beaver.py:31-34,57-63        a, b = th.randint(-2^63, 2^63 - 1, (1, *shape)); c = a∘b; generate_shares x3
    additive_shared.py:336-365       share_0 = LongTensor(shape).random_(-2^63, 2^63 - 1); share_1 = secret - share_0
    mpc/fss.py:344-358,495-501       alpha = np.random.randint(0, 2^32, n); s0 = randbit((2, 127, n))
    mpc/primitives.py:249-251        mask = np.random.randint(0, 2^32, n): party 0 gets alpha - mask, party 1 the mask

(paths under /root/reference/syft/frameworks/torch/).  The minting script checks every re-drawn primitive
against what the reference actually produced before it writes the fixture; the tests check the checksums.
Entries have the layout oracle.secure_oracle.ReplayDealer consumes.
"""
import numpy as np
import torch

LOW, HIGH = -(2 ** 63), 2 ** 63 - 1


def _u64sum(*arrays):
    s = np.uint64(0)
    with np.errstate(over="ignore"):
        for a in arrays:
            s = s + np.ascontiguousarray(a).view(np.uint64).sum(dtype=np.uint64)
    return s


def entry_checksum(e):
    if e[0] == "triple":
        return _u64sum(*[t for party in e[2] for t in party])
    if e[0] == "dif":
        return _u64sum(e[2], e[3], e[4])
    return _u64sum(e[1])


def entry_descriptor(e):
    """(kind code, element counts) — 0 mask, 1 mul triple, 2 matmul triple, 3 dif."""
    if e[0] == "triple":
        return (1 if e[1] == "mul" else 2, e[2][0][0].size, e[2][0][1].size)
    if e[0] == "dif":
        return (3, int(e[1]), 0)
    return (0, e[1].size, 0)


def _rmul(a, b):
    with np.errstate(over="ignore"):
        return (a.view(np.uint64) * b.view(np.uint64)).view(np.int64)


def _rmatmul(a, b):
    with np.errstate(over="ignore"):
        return (a.view(np.uint64) @ b.view(np.uint64)).view(np.int64)


def _rsub(a, b):
    with np.errstate(over="ignore"):
        return (a.view(np.uint64) - b.view(np.uint64)).view(np.int64)


class RefStream:
    """Re-draws the reference provider's primitives from (torch seed, numpy seed)."""

    def __init__(self, torch_seed, numpy_seed):
        self.tgen = torch.Generator().manual_seed(int(torch_seed))
        self.nrng = np.random.RandomState(int(numpy_seed))

    def _random(self, shape):
        return torch.empty(tuple(shape), dtype=torch.int64).random_(LOW, HIGH, generator=self.tgen).numpy()

    def mask(self, shape):
        return ("mask", self._random(shape))

    def triple(self, op, xshape, yshape):
        a = torch.randint(LOW, HIGH, (1, *xshape), dtype=torch.int64, generator=self.tgen).numpy()
        b = torch.randint(LOW, HIGH, (1, *yshape), dtype=torch.int64, generator=self.tgen).numpy()
        c = _rmul(a, b) if op == "mul" else _rmatmul(a, b)
        parts = []
        for v in (a, b, c):
            r = self._random(v.shape)
            parts.append((r[0], _rsub(v, r)[0]))
        return ("triple", op, [tuple(p[j] for p in parts) for j in range(2)])

    def dif(self, n):
        alpha = self.nrng.randint(0, 2 ** 32, size=(n,), dtype=np.uint64)
        s0 = self.nrng.randint(0, 2 ** 64, size=(2, 2, n), dtype=np.uint64)
        s0[:, 0] = s0[:, 0] % np.uint64(2 ** 63)
        r = self.nrng.randint(0, 2 ** 32, size=(n,), dtype=np.uint64)
        return ("dif", n, alpha.astype(np.int64), s0.view(np.int64), r.astype(np.int64))


class CheckedStream:
    """A RefStream whose every primitive is compared with the fixture's (descriptor, checksum) table."""

    def __init__(self, torch_seed, numpy_seed, desc, sums):
        self.s = RefStream(torch_seed, numpy_seed)
        self.desc, self.sums, self.pos = np.asarray(desc), np.asarray(sums), 0

    def _check(self, e):
        i = self.pos
        assert i < len(self.desc), "more primitives requested than the reference consumed"
        assert tuple(int(v) for v in self.desc[i]) == entry_descriptor(e), \
            f"primitive {i}: reference requested {tuple(self.desc[i])}, this run {entry_descriptor(e)}"
        assert np.uint64(self.sums[i]) == entry_checksum(e), f"primitive {i}: re-drawn randomness differs"
        self.pos += 1
        return e

    def mask(self, shape):
        return self._check(self.s.mask(shape))

    def triple(self, op, xshape, yshape):
        return self._check(self.s.triple(op, xshape, yshape))

    def dif(self, n):
        return self._check(self.s.dif(n))

    def done(self):
        return self.pos == len(self.desc)


class StreamReplayDealer:
    """oracle.secure_oracle.ReplayDealer interface over a (checked) stream instead of a stored log."""

    def __init__(self, stream):
        self.stream = stream

    def triple(self, op, xshape, yshape):
        from oracle import secure_oracle as S

        _, _, t = self.stream.triple(op, tuple(xshape), tuple(yshape))
        cshape = np.broadcast_shapes(tuple(xshape), tuple(yshape)) if op == "mul" else tuple(xshape[:-1]) + (yshape[-1],)
        del S
        return [(tj[0].reshape(xshape), tj[1].reshape(yshape), tj[2].reshape(cshape)) for tj in t]

    def dif_keys(self, n):
        from oracle import secure_oracle as S

        _, _, alpha, s0, r = self.stream.dif(n)
        _, keys = S.dif_keygen(alpha.astype(np.uint64), s0.view(np.uint64))
        return list(S.split_alpha(alpha.astype(np.uint64), r.astype(np.uint64))), keys

    def const_mask(self, *shape):
        return self.stream.mask(shape)[1]


# ---- log (de)serialisation for the small per-op fixtures ---------------------------------------------------
def pack_log(prefix, log, out):
    for i, e in enumerate(log):
        k = f"{prefix}/{i:04d}"
        if e[0] == "triple":
            out[f"{k}/triple_{e[1]}"] = np.array([0])
            for j in range(2):
                for name, t in zip("abc", e[2][j]):
                    out[f"{k}/{name}{j}"] = t
        elif e[0] == "dif":
            out[f"{k}/dif"] = np.array([e[1]])
            out[f"{k}/alpha"], out[f"{k}/s0"], out[f"{k}/r"] = e[2], e[3], e[4]
        else:
            out[f"{k}/mask"] = e[1]
    out[f"{prefix}/n"] = np.array([len(log)])


def unpack_log(prefix, z):
    log = []
    for i in range(int(z[f"{prefix}/n"][0])):
        k = f"{prefix}/{i:04d}"
        if f"{k}/mask" in z:
            log.append(("mask", z[f"{k}/mask"]))
        elif f"{k}/dif" in z:
            log.append(("dif", int(z[f"{k}/dif"][0]), z[f"{k}/alpha"], z[f"{k}/s0"], z[f"{k}/r"]))
        else:
            op = "mul" if f"{k}/triple_mul" in z else "matmul"
            log.append(("triple", op, [tuple(z[f"{k}/{n}{j}"] for n in "abc") for j in range(2)]))
    return log
