"""The oracle against vectors minted by RUNNING the reference's own MPC code (tests/golden/make_secure_ref_golden.py:
PySyft's spdz / beaver / primitives / fss / additive_shared / precision / nn.functional modules and PriMIA's
torchlib/models.py, executed from /root/reference).  Bit-exact, no tolerance.  CPU only."""
import os

import numpy as np
import pytest

from oracle import secure_oracle as S
from tests import ref_stream as RS
from tests import secure_cases as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "secure_ref_ops.npz"))


@pytest.mark.parametrize("pf", [3, 16])
@pytest.mark.parametrize("name", C.CASES)
def test_oracle_op_matches_reference(ops, name, pf):
    tag = f"{name}.p{pf}"
    log = RS.unpack_log(tag, ops)
    ctx = S.OracleContext(S.ReplayDealer(log), 10, pf)
    out = C.run_case(ctx, name, C.make_inputs(name), S.fix_encode)
    assert ctx.dealer.pos == len(log)          # exactly the primitives the reference requested, in its order
    for j in range(2):
        want = ops[f"{tag}/out{j}"]
        assert np.array_equal(np.asarray(out[j]).reshape(want.shape), want)
    dec = S.fix_decode(S.reconstruct(*out), 10, pf).reshape(ops[f"{tag}/decoded"].shape)
    assert np.array_equal(dec, ops[f"{tag}/decoded"])


def test_reference_decodes_sensibly_at_p3(ops):
    """Sanity of the fixtures themselves: at precision 3 (no ring wrap) the reference's encrypted results are the
    plaintext results up to fixed-point noise."""
    x, y = C.make_inputs("mul")
    assert np.allclose(ops["mul.p3/decoded"], x * y, atol=5e-3)
    v, = C.make_inputs("newton")
    assert np.allclose(ops["newton.p3/decoded"], 1 / np.sqrt(v), atol=5e-3)   # the iteration converges to var^-1/2
    r, = C.make_inputs("relu")
    assert np.allclose(ops["relu.p3/decoded"], np.maximum(np.trunc(r * 1000) / 1000, 0), atol=2e-3)


@pytest.mark.parametrize("pf", [3, 16])
def test_oracle_full_resnet18_forward_matches_reference(pf):
    """S12 at depth: all 8 blocks at real widths (split-K shapes [1,4608]x[4608,512] included), 32x32 input,
    6,546 primitives re-drawn from the seeds and checked one by one against the reference's."""
    z = np.load(os.path.join(GOLD, "secure_ref_forward.npz"))
    tag = f"fwd.p{pf}"
    tseed, nseed = [int(v) for v in z[f"{tag}/seeds"]]
    stream = RS.CheckedStream(tseed, nseed, z[f"{tag}/desc"], z[f"{tag}/sums"])
    sd, image = C.forward_model_and_image()
    ctx = S.OracleContext(RS.StreamReplayDealer(stream), 10, pf)
    out = S.secure_resnet_forward(ctx, sd, image, batched_newton=False)
    assert stream.done()
    for j in range(2):
        assert np.array_equal(out[j], z[f"{tag}/out{j}"])
    assert np.array_equal(S.fix_decode(S.reconstruct(*out), 10, pf), z[f"{tag}/decoded"])
    if pf == 3:
        assert np.allclose(z[f"{tag}/decoded"], z[f"{tag}/plain"], atol=3e-2)


def test_batched_newton_equals_layerwise_on_the_oracle():
    """The product hoists newton(running_var) of all BatchNorm layers into one batched iteration.  Per channel it is
    the same arithmetic: with each channel fed the randomness it had in the reference-order run — except the [1]-shaped
    re-shared constant 21*scale, which the batched form draws once per step instead of once per layer per step — the
    per-channel x_k sequences are identical when that constant's mask is the same.  Checked on a 2-layer toy."""
    rng = np.random.default_rng(5)

    def r64(*s):
        return rng.integers(-2 ** 63, 2 ** 63 - 1, size=s, dtype=np.int64)

    pf = 3
    sizes = [3, 5]
    var = [(rng.random(n) * 1.5 + 0.5).astype(np.float32) for n in sizes]

    class Recorder:
        def __init__(self):
            self.log = []

        def triple(self, op, xs, ys):
            a, b = r64(*xs), r64(*ys)
            t = S.build_triple(op, a, b, r64(*xs), r64(*ys), r64(*np.broadcast_shapes(xs, ys)))
            self.log.append(("triple", op, [tuple(t[j]) for j in range(2)]))
            return t

        def const_mask(self, *shape):
            self.log.append(("mask", const[len([e for e in self.log if e[0] == "mask"]) % len(const)]))
            return self.log[-1][1]

    # the SAME constant masks in every layer (one per step), so that a batched run can reuse them
    const = [r64(1) for _ in range(80)]
    outs, logs = [], []
    for v in var:
        rec = Recorder()
        ctx = S.OracleContext(rec, 10, pf)
        vs = list(S.generate_shares(S.fix_encode(v, 10, pf), r64(*v.shape)))
        outs.append(ctx.reciprocal_newton(vs))
        logs.append((vs, rec.log))
    merged = []
    for k in range(len(logs[0][1])):
        es = [lg[1][k] for lg in logs]
        if es[0][0] == "mask":
            assert all(np.array_equal(e[1], es[0][1]) for e in es)
            merged.append(es[0])
        else:
            merged.append(("triple", "mul", [tuple(np.concatenate([e[2][j][t] for e in es]) for t in range(3))
                                             for j in range(2)]))
    ctx = S.OracleContext(S.ReplayDealer(merged), 10, pf)
    both = ctx.reciprocal_newton([np.concatenate([lg[0][j] for lg in logs]) for j in range(2)])
    for j in range(2):
        assert np.array_equal(both[j], np.concatenate([o[j] for o in outs]))
