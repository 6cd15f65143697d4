"""GPU: the per-layer kernels of the in-process deployment (csrc/secure_local.hip, primia_dif_eval_local: both parties'
shares in one launch, opens as additions) against the step-by-step protocol chain (csrc/ring.hip / fss.hip, what a
three-role run executes) on the SAME dealer stream: every output share BIT-identical, no tolerance.  The chain itself is
held to the reference-minted fixtures by tests/test_gpu_secure_ref.py, which now also runs through these kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from primia_amd import resnet_spec as rs  # noqa: E402
from primia_amd.secure import Dealer, PreloadedDealer, SecureContext, SecureResNet18  # noqa: E402

I64 = torch.int64


def two_contexts(cuda, pf, record):
    """(fused context, chain context) fed the same primitives: `record(ctx)` runs once on a live dealer with a tape."""
    d = Dealer(cuda, seed=5)
    d.tape = []
    live = SecureContext(d, 10, pf)
    live.local_fused = False
    record(live)
    fused = SecureContext(PreloadedDealer(d.tape, cuda), 10, pf)
    chain = SecureContext(PreloadedDealer(d.tape, cuda), 10, pf)
    chain.local_fused = False
    assert fused._local and not chain._local
    return fused, chain


def shares_of(ctx, x):
    return ctx.share(ctx.encode(x), owner=1)


def same(a, b):
    return all(torch.equal(a[j], b[j]) for j in (0, 1))


@pytest.mark.parametrize("pf", [3, 16])
def test_every_layer_op_matches_the_chain(cuda, pf):
    g = torch.Generator().manual_seed(pf)
    C, H = 24, 10
    x = torch.randn(1, C, H, H, generator=g).to(cuda)
    w = (torch.randn(40, C, 3, 3, generator=g) * 0.1).to(cuda)
    wd = (torch.randn(40, C, 1, 1, generator=g) * 0.1).to(cuda)
    vec = [torch.randn(C, generator=g).to(cuda) for _ in range(3)] + [(torch.rand(C, generator=g) + 0.5).to(cuda)]
    fcw, fcb = torch.randn(3, C * 4, generator=g).to(cuda) * 0.1, torch.randn(3, generator=g).to(cuda)

    def program(ctx):
        xs, ws, wds = shares_of(ctx, x), shares_of(ctx, w), shares_of(ctx, wd)
        mean, bias, weight, var = [shares_of(ctx, v) for v in vec]
        out = {}
        out["add"], out["sub"] = ctx.add(xs, xs), ctx.sub(xs, ctx.add(xs, xs))
        out["relu"] = ctx.relu(xs)
        out["le"] = ctx.le(xs, out["relu"])
        out["fpt_mul"] = ctx.fpt_mul(xs, xs)
        out["mul_bcast"] = ctx.beaver_mul(ctx._each(lambda j: xs[j].permute(0, 2, 3, 1).reshape(-1, C).contiguous()), mean)
        out["conv3x3"] = ctx.conv2d(xs, ws, 1, 1)
        out["conv3x3s2"] = ctx.conv2d(xs, ws, 2, 1)
        out["conv1x1s2"] = ctx.conv2d(xs, wds, 2, 0)
        out["bn"] = ctx.batch_norm_eval(xs, mean, var, weight, bias)            # (Newton inside: inv = None)
        inv = ctx.reciprocal_newton(var)
        out["bn_inv"] = ctx.batch_norm_eval(xs, mean, var, weight, bias, inv=inv)
        out["maxpool"] = ctx.max_pool2d_3x3s2(xs)
        out["avgpool"] = ctx.avg_pool2d(xs, 5)
        flat = [t.reshape(1, -1) for t in out["avgpool"]]
        out["linear"] = ctx.linear(flat, shares_of(ctx, fcw), shares_of(ctx, fcb))
        return out

    fused, chain = two_contexts(cuda, pf, program)
    a, b = program(fused), program(chain)
    assert fused.dealer.pos == chain.dealer.pos == len(fused.dealer.tape)      # the same primitives, all of them
    for k in a:
        assert same(a[k], b[k]), k
    assert fused.stats == chain.stats


def test_ragged_sizes_and_split_k(cuda):
    """Shapes that do not fill tiles / blocks: odd channel counts and pixel counts (the 32 x 32 transposing tile of the
    BatchNorm kernel, the 64 x 64 GEMM tile, split-K with an uneven last slice)."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 37, 7, 9, generator=g).to(cuda)
    w = (torch.randn(70, 37, 3, 3, generator=g) * 0.1).to(cuda)
    v = [torch.randn(37, generator=g).to(cuda) for _ in range(3)] + [(torch.rand(37, generator=g) + 0.5).to(cuda)]

    def program(ctx):
        xs, ws = shares_of(ctx, x), shares_of(ctx, w)
        mean, bias, weight, var = [shares_of(ctx, t) for t in v]
        inv = ctx.reciprocal_newton(var)
        return {"conv": ctx.conv2d(xs, ws, 1, 1), "bn": ctx.batch_norm_eval(xs, mean, var, weight, bias, inv=inv),
                "pool": ctx.max_pool2d_3x3s2(xs), "relu": ctx.relu(xs)}

    fused, chain = two_contexts(cuda, 16, program)
    a, b = program(fused), program(chain)
    for k in a:
        assert same(a[k], b[k]), k


@pytest.mark.parametrize("pf", [3, 16])
def test_whole_network_matches_the_chain_and_launches_less(cuda, pf):
    """ResNet-18 (real widths, 64 x 64 image) through both forms on one dealer tape: logit shares bit-identical; the fused
    form issues fewer than 400 C-ABI calls per image where the chain issues about 1,000."""
    import primia_amd.secure as sec

    torch.manual_seed(3)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, 64, "max"))
    img = torch.randn(1, 3, 64, 64).to(cuda)
    d = Dealer(cuda, seed=9)
    d.tape = []
    live = SecureContext(d, 10, pf)
    live.local_fused = False
    m = SecureResNet18(live, sd, 64)
    ref = m.forward_shares(live.share(live.encode(img), owner=1))
    counts, outs = {}, {}
    orig = sec.call
    for name, fused in (("fused", True), ("chain", False)):
        ctx = SecureContext(PreloadedDealer(d.tape, cuda), 10, pf)
        ctx.local_fused = fused
        model = SecureResNet18(ctx, sd, 64)
        n = [0]

        def counting(fn, *a, **k):
            n[0] += 1
            return orig(fn, *a, **k)

        sec.call = counting
        try:
            outs[name] = model.forward_shares(ctx.share(ctx.encode(img), owner=1))
        finally:
            sec.call = orig
        counts[name] = n[0]
        assert ctx.dealer.pos == len(d.tape)
    assert same(outs["fused"], outs["chain"]) and same(outs["fused"], ref)
    print("C-ABI calls per image: fused", counts["fused"], "chain", counts["chain"])
    assert counts["fused"] < 400 < counts["chain"]


def test_transposed_weight_cache_follows_the_weights(cuda):
    """the fused conv2d / linear cache weight.reshape(O, -1).t() per model.  Shares edited in place, or new
    shares that the caching allocator puts at a freed model's address, must never meet a stale transpose."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 8, 6, 6, generator=g).to(cuda)
    w1 = (torch.randn(16, 8, 3, 3, generator=g) * 0.1).to(cuda)
    w2 = (torch.randn(16, 8, 3, 3, generator=g) * 0.1).to(cuda)

    def plain_conv(ctx, xs, ws):      # what the fused path must equal: the chain re-transposes on every call
        keep = ctx.local_fused
        ctx.local_fused = False
        try:
            return ctx.decode(ctx.reconstruct(ctx.conv2d(xs, ws, 1, 1)))
        finally:
            ctx.local_fused = keep

    ctx = SecureContext(Dealer(cuda, seed=9), 10, 3)      # (three fractional digits: products stay inside the ring)
    assert ctx._local
    xs, ws = shares_of(ctx, x), shares_of(ctx, w1)
    a = ctx.decode(ctx.reconstruct(ctx.conv2d(xs, ws, 1, 1)))
    assert a.abs().max() > 0.05
    assert torch.allclose(a, plain_conv(ctx, xs, ws), atol=5e-3)
    # in place: same tensors, new contents
    new = shares_of(ctx, w2)
    for j in (0, 1):
        ws[j].copy_(new[j])
    b = ctx.decode(ctx.reconstruct(ctx.conv2d(xs, ws, 1, 1)))
    assert torch.allclose(b, plain_conv(ctx, xs, ws), atol=5e-3)
    assert not torch.allclose(a, b, atol=2e-2)
    # freed and re-allocated at (very likely) the same address
    ptrs = (ws[0].data_ptr(), ws[1].data_ptr())
    del ws, new
    ws2 = shares_of(ctx, w1)
    c = ctx.decode(ctx.reconstruct(ctx.conv2d(xs, ws2, 1, 1)))
    assert torch.allclose(c, a, atol=5e-3), (ptrs, ws2[0].data_ptr())
    ctx.invalidate_weight_cache()
    assert not ctx._wt_cache
    # bounded, least recently used first out: a long-lived context fed changing weight objects does not grow
    ctx._wt_cache_max = 3
    kept = [shares_of(ctx, w1) for _ in range(5)]
    for ws_k in kept[:3]:
        ctx.conv2d(xs, ws_k, 1, 1)
    ctx.conv2d(xs, kept[0], 1, 1)                     # a hit: entry 0 becomes the most recently used
    ctx.conv2d(xs, kept[3], 1, 1)                     # evicts entry 1, not entry 0
    keys = list(ctx._wt_cache)
    assert len(keys) == 3 and (id(kept[0][0]), id(kept[0][1]), 16, 72) in keys
    assert (id(kept[1][0]), id(kept[1][1]), 16, 72) not in keys
