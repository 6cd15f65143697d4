"""GPU parity of the encrypted-inference kernels and protocol: BIT-EXACT against
 (a) the golden vectors minted from the reference's nn/functional.py and mpc/fss.py, and
 (b) the CPU oracle replaying the very same dealer stream (triples, FSS keys, re-sharing masks).
Integer ring work: no tolerance anywhere in this file."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import secure_oracle as S  # noqa: E402
from primia_amd._lib import call  # noqa: E402
from primia_amd.secure import Dealer, SecureContext, SecureResNet18  # noqa: E402

I64 = torch.int64


def dev(a, cuda):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(cuda)


def host(t):
    return t.cpu().numpy()


def test_layout_kernels_golden(cuda, golden_dir):
    g = np.load(os.path.join(golden_dir, "secure_layouts.npz"))
    for name in ("stem", "c3", "s2", "ds"):
        stride, pad = [int(v) for v in g[f"conv.{name}.meta"]]
        x, w = g[f"conv.{name}.x"], g[f"conv.{name}.w"]
        B, C, H, W = x.shape
        O, _, R, Sk = w.shape
        want, _, (_, _, Ho, Wo) = S.pre_conv(x, w, stride, pad)
        im = torch.empty(B, Ho * Wo, C * R * Sk, dtype=I64, device=cuda)
        call("primia_im2col_syft", dev(x, cuda), im, B, C, H, W, R, Sk, stride, pad)
        assert np.array_equal(host(im), want)
        assert np.array_equal(host(im)[:, :3, :], g[f"conv.{name}.im_head"])
        out = torch.empty(B, O, Ho, Wo, dtype=I64, device=cuda)
        call("primia_col2out_syft", dev(g[f"conv.{name}.res"], cuda), dev(g[f"conv.{name}.bias"], cuda), out, B,
             Ho * Wo, O)
        assert np.array_equal(host(out), g[f"conv.{name}.post"])
    for name in ("p3", "p7", "p2"):
        k, stride, pad = [int(v) for v in g[f"pool.{name}.meta"]]
        x = g[f"pool.{name}.x"]
        B, C, H, W = x.shape
        want = g[f"pool.{name}.im"]
        out = torch.empty(want.shape, dtype=I64, device=cuda)
        call("primia_pool_unroll_syft", dev(x, cuda), out, B, C, H, W, k, stride, pad)
        assert np.array_equal(host(out), want)


def test_trunc_and_beaver_golden(cuda, golden_dir):
    g = np.load(os.path.join(golden_dir, "secure_restated.npz"))
    x = dev(g["trunc.x"], cuda)
    for d in (10 ** 16, 10 ** 3, 20, 49):
        out = torch.empty_like(x)
        call("primia_trunc_div", x, d, out, x.numel())
        assert np.array_equal(host(out), g[f"trunc.d{d}"])
    for tag, op in (("beaver.mul.5x7", "mul"), ("beaver.mul.6x4", "mul"), ("beaver.matmul.1x9x12", "matmul")):
        xs = [dev(g[f"{tag}.x{j}"], cuda) for j in range(2)]
        ys = [dev(g[f"{tag}.y{j}"], cuda) for j in range(2)]
        t = [tuple(dev(g[f"{tag}.{n}{j}"], cuda) for n in "abc") for j in range(2)]
        d = [torch.empty_like(xs[0]) for _ in range(2)]
        e = [torch.empty_like(ys[0]) for _ in range(2)]
        for j in range(2):
            call("primia_ring_sub", xs[j], t[j][0], d[j], xs[j].numel(), xs[j].numel())
            call("primia_ring_sub", ys[j], t[j][1], e[j], ys[j].numel(), ys[j].numel())
        delta, eps = torch.empty_like(d[0]), torch.empty_like(e[0])
        call("primia_ring_add", d[0], d[1], delta, delta.numel(), delta.numel())
        call("primia_ring_add", e[0], e[1], eps, eps.numel(), eps.numel())
        for j in range(2):
            z = torch.empty(g[f"{tag}.z{j}"].shape, dtype=I64, device=cuda)
            if op == "mul":
                call("primia_beaver_combine_mul", j, delta, eps, t[j][0], t[j][1], t[j][2], z, z.numel(), eps.numel())
            else:
                M, K = xs[0].shape[-2:]
                N = ys[0].shape[-1]
                scratch = torch.empty(K * N, dtype=I64, device=cuda)
                call("primia_beaver_combine_matmul", j, delta, eps, t[j][0], t[j][1], t[j][2], z, scratch, M, K, N)
            assert np.array_equal(host(z), g[f"{tag}.z{j}"]), (tag, j)


@pytest.mark.parametrize("M,K,N", [(1, 512, 3), (49, 256, 512), (196, 130, 70), (784, 64, 128), (100, 147, 64)])
def test_ring_matmul(cuda, M, K, N):
    rng = np.random.default_rng(M + K)
    a = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(M, K), dtype=np.int64)
    b = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(K, N), dtype=np.int64)
    c0 = rng.integers(-2 ** 63, 2 ** 63 - 1, size=(M, N), dtype=np.int64)
    c = dev(c0, cuda)
    call("primia_ring_matmul", dev(a, cuda), dev(b, cuda), c, M, K, N, 1)
    assert np.array_equal(host(c), S.radd(c0, S.rmatmul(a, b)))


@pytest.mark.parametrize("op,xs,ys", [("mul", (37, 64), (37, 64)), ("mul", (64,), (49, 64)), ("mul", (1, 3, 5, 7), (7,)),
                                      ("matmul", (1, 49, 256), (256, 512)), ("matmul", (196, 130), (130, 70)),
                                      ("matmul", (1, 512), (512, 3))])
def test_dealer_triple_completion_kernels(cuda, op, xs, ys):
    """primia_triple_mul_c1 / primia_triple_matmul_c1 (mpc/beaver.py:7-63): with five uniform shares given, c1 completes the
    triple: (a0 + a1) o (b0 + b1) == c0 + c1 in the ring, against the oracle's numpy ring arithmetic."""
    rng = np.random.default_rng(len(xs) * 100 + len(ys))
    rnd = lambda sh: rng.integers(-2 ** 63, 2 ** 63 - 1, size=sh, dtype=np.int64)
    a0, a1, b0, b1 = rnd(xs), rnd(xs), rnd(ys), rnd(ys)
    a, b = S.radd(a0, a1), S.radd(b0, b1)
    want = S.rmul(a, b) if op == "mul" else S.rmatmul(a.reshape(xs[-2], xs[-1]), b).reshape(xs[:-1] + (ys[-1],))
    c0 = rnd(want.shape)
    d = Dealer(cuda, seed=1)
    c1 = torch.empty(want.shape, dtype=I64, device=cuda)
    d.triple_c1(op, xs, ys, dev(a0, cuda), dev(a1, cuda), dev(b0, cuda), dev(b1, cuda), dev(c0, cuda), c1)
    assert np.array_equal(S.radd(c0, host(c1)), want)
    # the dealer's own triples are complete ones
    t = d.triple(op, xs, ys)
    aa, bb, cc = (S.radd(host(t[0][k]), host(t[1][k])) for k in range(3))
    assert np.array_equal(cc, S.rmul(aa, bb) if op == "mul" else S.rmatmul(aa.reshape(xs[-2], xs[-1]), bb).reshape(cc.shape))


def test_dealer_keystream_counter_on_the_device_and_alpha_split(cuda):
    """primia_chacha20_fill_ctr draws the block (device counter + offset) of primia_chacha20_fill's stream — the RFC 8439
    checked keystream — and primia_u64_add advances the counter; primia_fss_alpha_split is build_fss_keys' host arithmetic
    (mpc/primitives.py:249-251, mpc/fss.py:346,495-501) in place, against numpy."""
    key, nonce = [3, 5, 7, 11], 77
    n = 8 * 300 + 5
    ref = torch.empty(n, dtype=I64, device=cuda)
    call("primia_chacha20_fill", *key, nonce, 1000 + 17, ref, n)
    ctr = torch.tensor([1000], dtype=I64, device=cuda)
    out = torch.empty(n, dtype=I64, device=cuda)
    call("primia_chacha20_fill_ctr", *key, nonce, ctr, 17, out, n)
    assert torch.equal(out, ref)
    call("primia_u64_add", ctr, 17)
    call("primia_chacha20_fill_ctr", *key, nonce, ctr, 0, out, n)
    assert torch.equal(out, ref) and int(ctr.item()) == 1017
    m = 1000
    rng = np.random.default_rng(9)
    raw = lambda *sh: rng.integers(0, 2 ** 64, size=sh, dtype=np.uint64)
    alpha, s0, r = raw(m), raw(2, 2, m), raw(m)
    ta, ts, tr = dev(alpha, cuda), dev(s0, cuda), dev(r, cuda)
    a0 = torch.empty(m, dtype=I64, device=cuda)
    call("primia_fss_alpha_split", ta, ts, tr, a0, m)
    m32, m63 = np.uint64(0xFFFFFFFF), np.uint64(0x7FFFFFFFFFFFFFFF)
    want_s0 = s0.copy()
    want_s0[:, 0] &= m63
    assert np.array_equal(host(ta).view(np.uint64), alpha & m32) and np.array_equal(host(tr).view(np.uint64), r & m32)
    assert np.array_equal(host(ts).view(np.uint64), want_s0)
    assert np.array_equal(host(a0).view(np.uint64), ((alpha & m32) - (r & m32)) & m32)


@pytest.mark.parametrize("kind", ["dif", "dpf"])
def test_fss_kernels_golden(cuda, golden_dir, kind):
    """Keys generated on the GPU from the fixture's (alpha, s0) equal the reference's keys, and
    evaluation reproduces the reference's output shares bit for bit (incl. x = alpha, alpha +- 1)."""
    g = np.load(os.path.join(golden_dir, "secure_fss.npz"))
    alpha, s0, x = g[f"{kind}.alpha"], g[f"{kind}.s0"], g[f"{kind}.x"]
    n = alpha.shape[0]
    _, okeys = (S.dif_keygen if kind == "dif" else S.dpf_keygen)(alpha, s0)
    bits = torch.empty(32, n, dtype=torch.uint8, device=cuda)
    cw_s = torch.empty(32, 2, n, dtype=I64, device=cuda)
    xm = dev(x.astype(np.uint32), cuda)
    if kind == "dif":
        cw_sigma = torch.empty(32, 2, n, dtype=I64, device=cuda)
        leaf = torch.empty(33, n, dtype=torch.int32, device=cuda)
        call("primia_dif_keygen", dev(alpha, cuda), dev(s0, cuda), bits, cw_sigma, cw_s, leaf, n)
        assert np.array_equal(host(leaf), g["dif.leaf"])
        assert np.array_equal(host(cw_sigma).view(np.uint64), okeys[0]["cw_sigma"])
        ob = okeys[0]["bits"]  # [32, 4, n] -> packed
        packed = ob[:, 0] | (ob[:, 1] << 1) | (ob[:, 2] << 2) | (ob[:, 3] << 3)
    else:
        cwn = torch.empty(n, dtype=I64, device=cuda)
        call("primia_dpf_keygen", dev(alpha, cuda), dev(s0, cuda), bits, cw_s, cwn, n)
        assert np.array_equal(host(cwn), g["dpf.leaf"])
        ob = okeys[0]["bits"]
        packed = ob[:, 0] | (ob[:, 1] << 1)
    assert np.array_equal(host(bits), packed.astype(np.uint8))
    assert np.array_equal(host(cw_s).view(np.uint64), okeys[0]["cw_s"])
    for b in range(2):
        out = torch.empty(n, dtype=I64, device=cuda)
        s0b = dev(s0[b], cuda)
        if kind == "dif":
            call("primia_dif_eval", b, xm, s0b, bits, cw_sigma, cw_s, leaf, out, n)
        else:
            call("primia_dpf_eval", b, xm, s0b, bits, cw_s, cwn, out, n)
        assert np.array_equal(host(out), g[f"{kind}.out{b}"]), (kind, b)


def shares_equal(gpu, ora):
    return all(np.array_equal(host(gpu[j]), ora[j]) for j in range(2))


@pytest.mark.parametrize("pf", [3, 16])
def test_protocol_ops_bit_exact_vs_oracle(cuda, pf):
    """relu, 9-window max tree, conv2d, Newton BN, avgpool, linear: GPU shares == oracle shares when
    the oracle replays the GPU dealer's stream."""
    dealer = Dealer(cuda, seed=11 + pf)
    dealer.log = []
    ctx = SecureContext(dealer, 10, pf)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 10, 10, generator=g) * 2
    w = torch.randn(8, 4, 3, 3, generator=g) * 0.3
    bn = dict(mean=torch.randn(8, generator=g) * 0.1, var=torch.rand(8, generator=g) + 0.5,
              weight=torch.rand(8, generator=g) + 0.5, bias=torch.randn(8, generator=g) * 0.1)
    fcw, fcb = torch.randn(3, 8, generator=g) * 0.2, torch.randn(3, generator=g) * 0.1

    def run(c, enc):
        xs = c.share(enc(x))
        ws = c.share(enc(w))
        bns = {k: c.share(enc(v)) for k, v in bn.items()}
        fw, fb = c.share(enc(fcw)), c.share(enc(fcb))
        outs = {}
        outs["relu"] = c.relu(xs)
        outs["pool"] = c.max_pool2d_3x3s2(xs)
        y = c.conv2d(xs, ws, 1, 1)
        outs["conv"] = y
        y = c.batch_norm_eval(y, bns["mean"], bns["var"], bns["weight"], bns["bias"])
        outs["bn"] = y
        y = c.relu(y)
        y = c.avg_pool2d(y, 10)
        outs["avg"] = y
        outs["fc"] = c.linear([t.reshape(1, -1) for t in y], fw, fb)
        return outs

    gout = run(ctx, lambda v: ctx.encode(v.to(cuda)))
    octx = S.OracleContext(S.ReplayDealer(dealer.log), 10, pf)
    oout = run(octx, lambda v: S.fix_encode(v.numpy(), 10, pf))
    assert octx.dealer.pos == len(dealer.log), "oracle consumed a different number of primitives"
    for k in gout:
        assert shares_equal(gout[k], oout[k]), f"{k} shares differ (pf={pf})"
    if pf == 3:  # sane precision: the decoded values mean something
        dec = ctx.decode(ctx.reconstruct(gout["relu"])).cpu()
        assert torch.allclose(dec, torch.relu((x * 1000).long().float() / 1000), atol=1e-6)


def mini_state_dict(gen):
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
    conv("layer2.0.conv1", 128, 64, 3)
    bn("layer2.0.bn1", 128)
    conv("layer2.0.conv2", 128, 128, 3)
    bn("layer2.0.bn2", 128)
    conv("layer2.0.downsample.0", 128, 64, 1)
    bn("layer2.0.downsample.1", 128)
    sd["fc.weight"] = torch.randn(3, 128, generator=gen) * 0.1
    sd["fc.bias"] = torch.randn(3, generator=gen) * 0.1
    return sd


@pytest.mark.parametrize("pf", [3, 16])
def test_mini_resnet_encrypted_forward_bit_exact(cuda, pf):
    """A 2-block ResNet with the full op mix (stem swap, identity + projection blocks, Newton BN,
    FSS relu / max-pool tree, avg-pool, fc) at 16x16 — output shares bit-exact vs the oracle."""
    gen = torch.Generator().manual_seed(21)
    sd = mini_state_dict(gen)
    image = torch.randn(1, 3, 16, 16, generator=gen)
    blocks = [("layer1.0", 1), ("layer2.0", 2)]
    dealer = Dealer(cuda, seed=5)
    dealer.log = []
    ctx = SecureContext(dealer, 10, pf)
    model = SecureResNet18(ctx, sd, input_size=16, blocks=blocks)
    xs = ctx.share(ctx.encode(image.to(cuda)))
    out = model.forward_shares(xs)
    octx = S.OracleContext(S.ReplayDealer(dealer.log), 10, pf)
    oout = S.secure_resnet_forward(octx, {k: v.numpy() for k, v in sd.items()}, image.numpy(), blocks)
    assert octx.dealer.pos == len(dealer.log)
    assert shares_equal(out, oout)
    dec = ctx.decode(ctx.reconstruct(out)).cpu().numpy()
    assert np.array_equal(dec, S.fix_decode(S.reconstruct(*oout), 10, pf))
    assert ctx.stats["dif_evals"] > 10000


def test_graphed_inference_matches_eager_and_refills(cuda):
    """The captured online phase replays bit-identically to the eager forward on the same primitives, and a
    refill (fresh dealer randomness, same buffers) decodes to the same logits up to fixed-point noise."""
    from primia_amd.secure import GraphedSecureInference, PreloadedDealer

    gen = torch.Generator().manual_seed(21)
    sd = mini_state_dict(gen)
    blocks = [("layer1.0", 1), ("layer2.0", 2)]
    g = GraphedSecureInference(sd, cuda, input_size=16, precision_fractional=16, seed=5, blocks=blocks)
    img = torch.randn(1, 3, 16, 16, generator=gen).to(cuda)
    out_g = g(img, refill=False).clone()
    ctx = SecureContext(PreloadedDealer(g.tape, cuda), 10, 16)
    out_e = SecureResNet18(ctx, sd, 16, blocks)(img)
    assert torch.equal(out_g, out_e)
    before = g.tape[-1].clone() if torch.is_tensor(g.tape[-1]) else None
    arena = g._arena.clone()
    out_r = g(img).clone()          # new primitives, same image
    assert torch.allclose(out_r, out_g, atol=1e-3)
    if before is not None:
        assert not torch.equal(before, g.tape[-1])
    # every replay of the captured refill draws FRESH keystream (the block counter is a device word the graph advances),
    # and the refilled primitives are complete: every triple multiplies out, on the tape the online graph reads
    assert g.refills == 2 and g._refill_g is not None
    assert int((g._arena == arena).sum()) <= 2          # (64-bit coincidences aside, no word repeats)
    arena2 = g._arena.clone()
    g.refill()
    assert int((g._arena == arena2).sum()) <= 2
    n_tr = 0
    for i in range(g._n_model, len(g.tape)):
        kind, args, _ = g.requests[i]
        if kind != "triple":
            continue
        op, xs, ys = args
        t = g.tape[i]
        aa, bb, cc = (S.radd(host(t[0][k]), host(t[1][k])) for k in range(3))
        want = S.rmul(aa, bb) if op == "mul" else S.rmatmul(aa.reshape(xs[-2], xs[-1]), bb).reshape(cc.shape)
        assert np.array_equal(cc, want), i
        n_tr += 1
    assert n_tr > 20
    # the serving form returns its STATIC output buffer (the graph writes it in place): a caller that keeps results copies them,
    # as inference.py --hip_graph does since round 6 (its logits dump used to hold the last image's row twice)
    # (three fractional digits: at the reference's literal 16 every product wraps in the 2^64 ring and the decoded logits of ANY
    # image are the fc bias — SURVEY.md §8c quirk (a), measured by tools/secure_sensitivity.py — so two images could not differ)
    g3 = GraphedSecureInference(sd, cuda, input_size=16, precision_fractional=3, seed=5, blocks=blocks)
    img2 = torch.randn(1, 3, 16, 16, generator=gen).to(cuda)
    r1 = g3(img)
    keep1 = r1.clone()
    r2 = g3(img2)
    assert r1 is r2 and not torch.allclose(keep1, r2, atol=1e-2)
    assert bool((keep1.cpu() - sd["fc.bias"]).abs().max() > 0.1)       # ... and they are not the bias
    # the eager form of the refill (refill_graph = False) hands out the same primitives: same logits, bit for bit
    outs = []
    for use_graph in (True, False):
        GraphedSecureInference.refill_graph = use_graph
        try:
            h = GraphedSecureInference(sd, cuda, input_size=16, precision_fractional=16, seed=5, blocks=blocks)
        finally:
            GraphedSecureInference.refill_graph = True
        assert (h._refill_g is not None) == use_graph
        outs.append([h(img).clone() for _ in range(3)])
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("pf", [16])
def test_three_role_deployment_bit_identical_to_in_process(cuda, tmp_path, pf):
    """model_owner / data_owner / crypto_provider as three ranks: party j holds only share j, the
    dealer ships each half of every primitive to its party, opens are 2-party all_reduces.  With the same
    dealer seed the decoded logits equal the in-process run bit for bit, on both parties."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = torch.Generator().manual_seed(21)
    sd = mini_state_dict(gen)
    images = torch.randn(2, 3, 16, 16, generator=gen)
    blocks = [("layer1.0", 1), ("layer2.0", 2)]
    ctx = SecureContext(Dealer(cuda, seed=5), 10, pf)
    model = SecureResNet18(ctx, sd, input_size=16, blocks=blocks)
    want = torch.cat([model(images[i:i + 1].to(cuda)) for i in range(2)]).cpu()
    out = str(tmp_path / "logits")
    from tests.conftest import free_port

    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "party_worker.py"),
           out, str(pf)]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for j in range(2):
        assert torch.equal(torch.load(f"{out}.{j}"), want), j


def test_eager_forwards_do_not_retain_newton_primitives(cuda):
    """inference.py's default path: one SecureContext with a live dealer, model(image) per image.  The fused Newton
    launch must not pin its 237 triples (or cache a pointer table) per call: table list and device memory stay flat."""
    gen = torch.Generator().manual_seed(21)
    sd = mini_state_dict(gen)
    blocks = [("layer1.0", 1), ("layer2.0", 2)]
    ctx = SecureContext(Dealer(cuda, seed=5), 10, 16)
    model = SecureResNet18(ctx, sd, input_size=16, blocks=blocks)
    img = torch.randn(1, 3, 16, 16, generator=gen).to(cuda)
    model(img)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for _ in range(5):
        out = model(img)
    torch.cuda.synchronize()
    del out
    assert len(ctx._newton_tables) == 0
    assert torch.cuda.memory_allocated() <= base + (1 << 20)


def test_pipelined_inference_hides_the_dealer_without_changing_a_bit(cuda):
    """PipelinedSecureInference: two graph slots, the dealer refilling one on its own stream while the other replays.
    A stream of images must come out (a) exactly as the single-slot serving form produces them from the same dealer
    seeds — image i on slot i % 2 after i // 2 refills — and (b) identically from run to run (a refill racing a replay
    would not); every result decodes to the plaintext logits up to fixed-point noise."""
    from primia_amd.secure import GraphedSecureInference, PipelinedSecureInference

    gen = torch.Generator().manual_seed(33)
    sd = mini_state_dict(gen)
    blocks = [("layer1.0", 1), ("layer2.0", 2)]
    imgs = [torch.randn(1, 3, 16, 16, generator=gen).to(cuda) for _ in range(6)]
    runs = []
    for _ in range(2):
        p = PipelinedSecureInference(sd, cuda, input_size=16, precision_fractional=16, seed=11, blocks=blocks)
        # no image runs on the primitives the all-zero warm-up image consumed: every slot was refilled once
        assert all(sl.refills == 1 for sl in p.slots)
        runs.append([p(im) for im in imgs])
        torch.cuda.synchronize()
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    # the serial form with the same seeds: slot k = GraphedSecureInference(seed 11 + 7919 k), refilled between its images
    serial = [GraphedSecureInference(sd, cuda, input_size=16, precision_fractional=16, seed=11 + 7919 * k, blocks=blocks)
              for k in range(2)]
    for i, im in enumerate(imgs):
        ref = serial[i % 2](im, refill=i >= 2).clone()
        assert torch.equal(runs[0][i], ref), i
    # fresh primitives only add fixed-point noise
    again = p(imgs[0])
    assert torch.allclose(again, runs[0][0], atol=1e-3)
