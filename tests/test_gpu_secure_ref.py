"""The HIP encrypted-inference path against vectors minted by RUNNING the reference's own MPC code
(tests/golden/make_secure_ref_golden.py).  The product's SecureContext / SecureResNet18 is fed the reference
provider's primitives — stored logs for the per-operation cases, the seed-regenerated and checksum-verified stream
for the full ResNet-18 — and must return the reference's output shares bit for bit.  Everything goes through the C ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import secure_oracle as S  # noqa: E402
from primia_amd._lib import call  # noqa: E402
from primia_amd.secure import Dealer, SecureContext, SecureResNet18  # noqa: E402
from tests import ref_stream as RS  # noqa: E402
from tests import secure_cases as C  # noqa: E402

I64 = torch.int64
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class LogStream:
    """A stored primitive log behind the RefStream interface."""

    def __init__(self, log):
        self.log, self.pos = log, 0

    def _next(self, kind):
        e = self.log[self.pos]
        self.pos += 1
        assert e[0] == kind, (self.pos - 1, kind, e[0])
        return e

    def mask(self, shape):
        e = self._next("mask")
        assert e[1].size == int(np.prod(shape))
        return e

    def triple(self, op, xshape, yshape):
        e = self._next("triple")
        assert e[1] == op and e[2][0][0].size == int(np.prod(xshape)) and e[2][0][1].size == int(np.prod(yshape))
        return e

    def dif(self, n):
        e = self._next("dif")
        assert e[1] == n
        return e

    def done(self):
        return self.pos == len(self.log)


class DeviceStreamDealer:
    """The product's dealer interface (triple / dif_keys / const_mask) fed by the reference's primitives: tensors are
    uploaded as they are; DIF correction words are re-derived on the GPU from (alpha, s0) with primia_dif_keygen —
    itself pinned to the reference's keygen in test_gpu_secure.py — and alpha is split as primitives.py:249-251 does."""

    def __init__(self, stream, device):
        self.stream, self.device = stream, torch.device(device)
        self.log = self.tape = self.requests = None

    def _up(self, a, shape=None):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        return t if shape is None else t.reshape(shape)

    def triple(self, op, xshape, yshape):
        _, _, t = self.stream.triple(op, tuple(xshape), tuple(yshape))
        cshape = tuple(np.broadcast_shapes(tuple(xshape), tuple(yshape))) if op == "mul" else tuple(xshape[:-1]) + (yshape[-1],)
        return [(self._up(tj[0], tuple(xshape)), self._up(tj[1], tuple(yshape)), self._up(tj[2], cshape)) for tj in t]

    def dif_keys(self, n):
        _, _, alpha, s0, r = self.stream.dif(n)
        dev = self.device
        alpha_d, s0_d, r_d = self._up(alpha), self._up(s0).contiguous(), self._up(r)
        bits = torch.empty(32, n, dtype=torch.uint8, device=dev)
        cw_sigma = torch.empty(32, 2, n, dtype=I64, device=dev)
        cw_s = torch.empty(32, 2, n, dtype=I64, device=dev)
        leaf = torch.empty(33, n, dtype=torch.int32, device=dev)
        call("primia_dif_keygen", alpha_d, s0_d, bits, cw_sigma, cw_s, leaf, n)
        a0 = (alpha_d - r_d) & 0xFFFFFFFF
        return [dict(alpha=[a0, r_d][b], s0=s0_d[b].contiguous(), bits=bits, cw_sigma=cw_sigma, cw_s=cw_s, cw_leaf=leaf)
                for b in range(2)]

    def const_mask(self, *shape, owner=None):
        return self._up(self.stream.mask(tuple(shape))[1], tuple(shape))


def gpu_encode(cuda):
    def enc(x, base, pf):
        t = torch.from_numpy(np.ascontiguousarray(x)).to(cuda)
        q = torch.empty(t.shape, dtype=I64, device=cuda)
        call("primia_fx_encode", t, q, t.numel(), float(base ** pf))
        return q

    return enc


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "secure_ref_ops.npz"))


@pytest.mark.parametrize("pf", [3, 16])
@pytest.mark.parametrize("name", C.CASES)
def test_hip_op_matches_reference(cuda, ops, name, pf):
    tag = f"{name}.p{pf}"
    stream = LogStream(RS.unpack_log(tag, ops))
    ctx = SecureContext(DeviceStreamDealer(stream, cuda), 10, pf)
    out = C.run_case(ctx, name, C.make_inputs(name), gpu_encode(cuda))
    assert stream.done()
    for j in range(2):
        want = ops[f"{tag}/out{j}"]
        assert np.array_equal(out[j].cpu().numpy().reshape(want.shape), want), (tag, j)
    dec = ctx.decode(ctx.reconstruct(out)).cpu().numpy().reshape(ops[f"{tag}/decoded"].shape)
    assert np.array_equal(dec, ops[f"{tag}/decoded"])


@pytest.mark.parametrize("pf", [3, 16])
def test_hip_full_resnet18_forward_matches_reference(cuda, pf):
    """S12 at depth through the product path: all 8 blocks at real widths, 32x32 input, the reference's own
    ResNet-18 (torchlib/models.py) run on PySyft's MPC code as the expected value."""
    z = np.load(os.path.join(GOLD, "secure_ref_forward.npz"))
    tag = f"fwd.p{pf}"
    tseed, nseed = [int(v) for v in z[f"{tag}/seeds"]]
    stream = RS.CheckedStream(tseed, nseed, z[f"{tag}/desc"], z[f"{tag}/sums"])
    sd, image = C.forward_model_and_image()
    ctx = SecureContext(DeviceStreamDealer(stream, cuda), 10, pf)
    model = SecureResNet18(ctx, {k: torch.from_numpy(v) for k, v in sd.items()}, input_size=C.FWD_SIZE,
                           batched_newton=False)
    xs = ctx.share(ctx.encode(torch.from_numpy(image).to(cuda)), owner=1)
    out = model.forward_shares(xs)
    assert stream.done()
    for j in range(2):
        assert np.array_equal(out[j].cpu().numpy(), z[f"{tag}/out{j}"]), j
    assert np.array_equal(ctx.decode(ctx.reconstruct(out)).cpu().numpy(), z[f"{tag}/decoded"])
    assert ctx.stats["dif_evals"] > 60000


@pytest.mark.parametrize("pf", [16])
def test_hip_batched_newton_full_forward_matches_oracle(cuda, pf):
    """The default (hoisted-Newton) forward of the full network at 32x32 against the oracle in the same mode, on the
    GPU dealer's own ChaCha20 stream."""
    sd, image = C.forward_model_and_image()
    dealer = Dealer(cuda, seed=11)
    dealer.log = []
    ctx = SecureContext(dealer, 10, pf)
    model = SecureResNet18(ctx, {k: torch.from_numpy(v) for k, v in sd.items()}, input_size=C.FWD_SIZE)
    out = model.forward_shares(ctx.share(ctx.encode(torch.from_numpy(image).to(cuda)), owner=1))
    octx = S.OracleContext(S.ReplayDealer(dealer.log), 10, pf)
    oout = S.secure_resnet_forward(octx, sd, image)
    assert octx.dealer.pos == len(dealer.log)
    for j in range(2):
        assert np.array_equal(out[j].cpu().numpy(), oout[j])


def test_chacha20_rfc8439_vector_and_stream(cuda):
    """primia_chacha20_fill against the RFC 8439 section 2.3.2 block (key 00..1f, counter 1, nonce 00:00:00:09
    00:00:00:4a 00:00:00:00) and against a pure-Python ChaCha20 on a ragged multi-block request."""
    import struct

    def block(key_words, ctr, nonce):
        def rotl(v, c):
            return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))

        s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key_words) + [ctr & 0xFFFFFFFF, ctr >> 32,
                                                                                   nonce & 0xFFFFFFFF, nonce >> 32]
        x = list(s)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = rotl(x[b] ^ x[c], 7)

        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        return struct.pack("<16I", *[(a + b) & 0xFFFFFFFF for a, b in zip(x, s)])

    key = bytes(range(32))
    kw = struct.unpack("<8I", key)
    k64 = struct.unpack("<4Q", key)
    # RFC layout: word 12 = counter 1, words 13..15 = nonce 09000000 4a000000 00000000 -> here counter = 1 | 0x09000000 << 32
    ctr = 1 | (0x09000000 << 32)
    nonce = 0x4a000000
    out = torch.empty(8, dtype=I64, device=cuda)
    call("primia_chacha20_fill", *k64, nonce, ctr, out, 8)
    got = out.cpu().numpy().tobytes()
    assert got == block(kw, ctr, nonce)
    assert got[:16].hex() == "10f1e7e4d13b5915500fdd1fa32071c4"     # RFC 8439 2.3.2, first keystream bytes
    n = 8 * 37 + 5
    out = torch.empty(n, dtype=I64, device=cuda)
    call("primia_chacha20_fill", *k64, 77, 1000, out, n)
    want = b"".join(block(kw, 1000 + i, 77) for i in range(38))[:8 * n]
    assert out.cpu().numpy().tobytes() == want


def test_dealer_is_unpredictable_by_default(cuda):
    """Two dealers built without a seed draw different streams (keys from os.urandom); a debug seed reproduces."""
    a, b = Dealer(cuda).rand64(64), Dealer(cuda).rand64(64)
    assert not torch.equal(a, b)
    assert torch.equal(Dealer(cuda, seed=3).rand64(64), Dealer(cuda, seed=3).rand64(64))
    assert not torch.equal(Dealer(cuda, seed=3).rand64(64), Dealer(cuda, seed=4).rand64(64))


@pytest.mark.parametrize("pf", [3, 16])
def test_fused_newton_is_bit_identical_to_the_step_by_step_chain(cuda, pf):
    """primia_newton_reciprocal_local (one launch, both parties on this GPU) against the chain of ring / Beaver /
    truncation launches a three-role run executes, on the same dealer stream; the reference fixtures above
    (newton, bn_eval, the full forward) already run through the fused form."""
    g = torch.Generator().manual_seed(4)
    var = (torch.rand(1000, generator=g) * 1.5 + 0.5).to(cuda)
    outs, logs = [], []
    for fuse in (True, False):
        dealer = Dealer(cuda, seed=21)
        dealer.log = []
        ctx = SecureContext(dealer, 10, pf)
        ctx.fuse_newton = fuse
        v = ctx.share(ctx.encode(var))
        outs.append(ctx.reciprocal_newton(v))
        logs.append(dealer.log)
    assert len(logs[0]) == len(logs[1]) == 318 and all(a[0] == b[0] for a, b in zip(*logs))
    for j in range(2):
        assert torch.equal(outs[0][j], outs[1][j])
    if pf == 3:
        dec = ctx.decode(ctx.reconstruct(outs[0])).cpu()
        assert torch.allclose(dec, var.cpu().rsqrt(), atol=5e-3)
