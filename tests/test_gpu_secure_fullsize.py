"""GPU parity of the encrypted-inference path AT BASELINE configs[4]'s real size (one 3x224x224 image): the shapes
that only exist at full size — ring GEMMs [12544,147]x[147,64], [3136,576]x[576,64], [49,4608]x[4608,512]
(mpc/spdz.py:63-122), DIF batches of 802,816 / 200,704 comparisons (mpc/fss.py:400-428), the 9-window max tree
on [1,64,112,112] (nn/functional.py:460-525), im2col of the 224 stem (nn/functional.py:78-166) — and one
end-to-end segment stem -> pool -> relu -> layer1.0 -> head, all BIT-EXACT against the CPU oracle replaying the
GPU dealer's stream.  The oracle fans FSS work out over processes the way the reference does above MULTI_LIMIT
(mpc/fss.py:43-44,214-266); the workers are spawned (fresh interpreters, numpy only), never forked from this
process, which holds a HIP context."""
import multiprocessing as mp
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import secure_oracle as S  # noqa: E402
from primia_amd._lib import call  # noqa: E402
from primia_amd.secure import Dealer, SecureContext, SecureResNet18  # noqa: E402

I64 = torch.int64


@pytest.fixture(scope="module")
def oracle_pool():
    n = max(4, min(64, (os.cpu_count() or 8)))
    with mp.get_context("spawn").Pool(n) as pool:
        S.use_pool(pool, n_slices=2 * n)
        yield pool
        S.use_pool(None)


def dev(a, cuda):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(a).to(cuda)


def host(t):
    return t.cpu().numpy()


FULL_GEMMS = [(12544, 147, 64), (3136, 576, 64), (784, 1152, 128), (196, 2304, 256), (49, 4608, 512)]


@pytest.mark.parametrize("M,K,N", FULL_GEMMS)
def test_ring_matmul_full_size(cuda, M, K, N):
    rng = np.random.default_rng(M * 7 + K)
    a = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(M, K), dtype=np.int64)
    b = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(K, N), dtype=np.int64)
    c0 = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(M, N), dtype=np.int64)
    want = S.rmatmul(a, b)
    c = dev(c0, cuda)
    call("primia_ring_matmul", dev(a, cuda), dev(b, cuda), c, M, K, N, 1)          # accumulate form
    assert np.array_equal(host(c), S.radd(c0, want))
    c = torch.empty(M, N, dtype=I64, device=cuda)
    call("primia_ring_matmul", dev(a, cuda), dev(b, cuda), c, M, K, N, 0)
    assert np.array_equal(host(c), want)


@pytest.mark.parametrize("M,K,N", [FULL_GEMMS[0], FULL_GEMMS[1], FULL_GEMMS[4]])
def test_beaver_combine_matmul_full_size(cuda, M, K, N):
    """spdz_compute with op = matmul (mpc/spdz.py:63-122) on full-size operands: z_j = delta.b_j + a_j.eps + c_j
    (+ delta.eps for j = 0), the oracle's three (four) separate products against the kernel's fused two."""
    rng = np.random.default_rng(M + N)

    def r(*s):
        return rng.integers(-2 ** 63, 2 ** 63 - 1, size=s, dtype=np.int64)

    delta, eps = r(M, K), r(K, N)
    for j in range(2):
        a, b, c = r(M, K), r(K, N), r(M, N)
        want = S.spdz_compute(j, delta, eps, a, b, c, "matmul")
        z = torch.empty(M, N, dtype=I64, device=cuda)
        scratch = torch.empty(K * N, dtype=I64, device=cuda)
        call("primia_beaver_combine_matmul", j, dev(delta, cuda), dev(eps, cuda), dev(a, cuda), dev(b, cuda),
             dev(c, cuda), z, scratch, M, K, N)
        assert np.array_equal(host(z), want), j


@pytest.mark.parametrize("n", [802_816, 200_704])
def test_dif_full_size_batches(cuda, oracle_pool, n):
    """primia_dif_keygen + primia_dif_eval on the stem pool's / a layer1 ReLU's comparison count: every key part and
    both output shares equal the oracle's, and the shares reconstruct [x <= alpha] (incl. x = alpha, alpha +- 1)."""
    rng = np.random.default_rng(n)
    alpha = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64)
    s0 = rng.integers(0, 2 ** 64, size=(2, 2, n), dtype=np.uint64)
    s0[:, 0] &= np.uint64(0x7FFFFFFFFFFFFFFF)
    x = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64)
    x[::5] = alpha[::5]
    x[1::5] = (alpha[1::5] + np.uint64(1)) & np.uint64(0xFFFFFFFF)
    x[2::5] = (alpha[2::5] - np.uint64(1)) & np.uint64(0xFFFFFFFF)
    _, okeys = S.dif_keygen(alpha, s0)
    bits = torch.empty(32, n, dtype=torch.uint8, device=cuda)
    cw_sigma = torch.empty(32, 2, n, dtype=I64, device=cuda)
    cw_s = torch.empty(32, 2, n, dtype=I64, device=cuda)
    leaf = torch.empty(33, n, dtype=torch.int32, device=cuda)
    call("primia_dif_keygen", dev(alpha, cuda), dev(s0, cuda), bits, cw_sigma, cw_s, leaf, n)
    ob = okeys[0]["bits"]
    assert np.array_equal(host(bits), (ob[:, 0] | (ob[:, 1] << 1) | (ob[:, 2] << 2) | (ob[:, 3] << 3)).astype(np.uint8))
    assert np.array_equal(host(cw_sigma).view(np.uint64), okeys[0]["cw_sigma"])
    assert np.array_equal(host(cw_s).view(np.uint64), okeys[0]["cw_s"])
    assert np.array_equal(host(leaf), okeys[0]["cw_leaf"])
    xm = dev(x.astype(np.uint32).view(np.int32), cuda)
    outs = []
    for b in range(2):
        out = torch.empty(n, dtype=I64, device=cuda)
        call("primia_dif_eval", b, xm, dev(s0[b], cuda), bits, cw_sigma, cw_s, leaf, out, n)
        assert np.array_equal(host(out), S.dif_eval(b, x, okeys[b])), b
        outs.append(host(out))
    assert np.array_equal(S.radd(outs[0], outs[1]), (x <= alpha).astype(np.int64))


def test_im2col_of_the_224_stem_and_layer_shapes(cuda):
    """_pre_conv (nn/functional.py:78-166) at the full-size shapes: stem [1,3,224,224] 7x7/2 -> [12544,147], layer1
    [1,64,56,56] 3x3 -> [3136,576], layer4 [1,512,7,7] 3x3 -> [49,4608], layer2's 1x1/2 -> [784,64]; and the pool
    unroll of [1,64,112,112] (_pre_pool, nn/functional.py:311-393) -> [1,64,3136,9]."""
    rng = np.random.default_rng(5)
    for (C, H, O, R, stride, pad) in [(3, 224, 64, 7, 2, 3), (64, 56, 64, 3, 1, 1), (512, 7, 512, 3, 1, 1),
                                      (64, 56, 128, 1, 2, 0), (64, 56, 128, 3, 2, 1)]:
        x = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(1, C, H, H), dtype=np.int64)
        w = np.zeros((O, C, R, R), dtype=np.int64)
        want, _, (_, _, Ho, Wo) = S.pre_conv(x, w, stride, pad)
        im = torch.empty(1, Ho * Wo, C * R * R, dtype=I64, device=cuda)
        call("primia_im2col_syft", dev(x, cuda), im, 1, C, H, H, R, R, stride, pad)
        assert np.array_equal(host(im), want), (C, H, R, stride)
    x = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(1, 64, 112, 112), dtype=np.int64)
    want, _ = S.pre_pool(x, 3, 2, 1)
    out = torch.empty(want.shape, dtype=I64, device=cuda)
    call("primia_pool_unroll_syft", dev(x, cuda), out, 1, 64, 112, 112, 3, 2, 1)
    assert np.array_equal(host(out), want)


def shares_equal(gpu, ora):
    return all(np.array_equal(host(gpu[j]), ora[j]) for j in range(2))


@pytest.mark.parametrize("pf", [16])
def test_max_pool_tree_and_relu_on_112x112(cuda, oracle_pool, pf):
    """The stem's swapped tail at full size (inference.py:289): the 9-window max tree on [1,64,112,112] — 802,816 +
    401,408 + 200,704 + 200,704 comparisons — then ReLU on [1,64,56,56]."""
    dealer = Dealer(cuda, seed=31)
    dealer.log = []
    ctx = SecureContext(dealer, 10, pf)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 64, 112, 112, generator=g) * 2

    def run(c, enc):
        xs = c.share(enc(x))
        p = c.max_pool2d_3x3s2(xs)
        return {"pool": p, "relu": c.relu(p)}

    gout = run(ctx, lambda v: ctx.encode(v.to(cuda)))
    octx = S.OracleContext(S.ReplayDealer(dealer.log), 10, pf)
    oout = run(octx, lambda v: S.fix_encode(v.numpy(), 10, pf))
    assert octx.dealer.pos == len(dealer.log)
    assert ctx.stats["dif_evals"] == 802_816 + 401_408 + 200_704 + 200_704 + 200_704
    for k in gout:
        assert shares_equal(gout[k], oout[k]), k


def segment_state_dict(gen):
    sd = {}

    def conv(name, o, i, k):
        sd[name + ".weight"] = torch.randn(o, i, k, k, generator=gen) * (1.0 / (i * k * k) ** 0.5)

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=gen) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=gen) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=gen) * 0.1
        sd[name + ".running_var"] = torch.rand(c, generator=gen) + 0.5
        sd[name + ".num_batches_tracked"] = torch.tensor(1)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    conv("layer1.0.conv1", 64, 64, 3)
    bn("layer1.0.bn1", 64)
    conv("layer1.0.conv2", 64, 64, 3)
    bn("layer1.0.bn2", 64)
    sd["fc.weight"] = torch.randn(3, 64, generator=gen) * 0.1
    sd["fc.bias"] = torch.randn(3, generator=gen) * 0.1
    return sd


@pytest.mark.parametrize("pf", [16, 3])
def test_224_segment_stem_pool_layer1_bit_exact(cuda, oracle_pool, pf):
    """One 3x224x224 image through conv1 (7x7/2, [12544,147]x[147,64]) -> bn1 -> max pool -> relu -> layer1.0 (two
    [3136,576]x[576,64] convs, Newton BatchNorm, residual, two ReLUs of 200,704) -> AvgPool2d(56) -> fc: the GPU's
    output shares and every intermediate the head depends on equal the oracle's on the replayed dealer stream."""
    gen = torch.Generator().manual_seed(77)
    sd = segment_state_dict(gen)
    image = torch.randn(1, 3, 224, 224, generator=gen)
    blocks = [("layer1.0", 1)]
    dealer = Dealer(cuda, seed=9 + pf)
    dealer.log = []
    ctx = SecureContext(dealer, 10, pf)
    model = SecureResNet18(ctx, sd, input_size=224, blocks=blocks)
    out = model.forward_shares(ctx.share(ctx.encode(image.to(cuda))))
    octx = S.OracleContext(S.ReplayDealer(dealer.log), 10, pf)
    oout = S.secure_resnet_forward(octx, {k: v.numpy() for k, v in sd.items()}, image.numpy(), blocks)
    assert octx.dealer.pos == len(dealer.log)
    assert shares_equal(out, oout)
    assert ctx.stats["dif_evals"] == 1_605_632 + 3 * 200_704
    dec = ctx.decode(ctx.reconstruct(out)).cpu().numpy()
    assert np.array_equal(dec, S.fix_decode(S.reconstruct(*oout), 10, pf))
