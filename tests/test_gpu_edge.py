"""GPU: boundary behaviour of the C ABI — empty inputs, ragged sizes, bad arguments (error codes,
not crashes), batch 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import secure_oracle as S  # noqa: E402
from primia_amd import _lib  # noqa: E402
from primia_amd._lib import ConvDesc, PrimiaError, call  # noqa: E402

I64 = torch.int64


def test_empty_inputs_are_no_ops(cuda):
    z = torch.empty(0, dtype=I64, device=cuda)
    one = torch.ones(1, dtype=I64, device=cuda)
    call("primia_ring_add", z, one, z, 0, 1)
    call("primia_trunc_div", z, 10, z, 0)
    call("primia_fss_mask", z, z, z, z, 0)
    call("primia_dif_eval", 0, z, z, z, z, z, z, z, 0)
    f = torch.empty(0, device=cuda)
    call("primia_sgd_step", torch.empty(4, device=cuda), torch.empty(4, device=cuda), 0, 0.1, 0.0)
    call("primia_fx_encode", f, z, 0, 1000.0)


def test_bad_arguments_return_error_codes(cuda):
    x = torch.zeros(64, dtype=torch.bfloat16, device=cuda)
    bad = ConvDesc.make(1, 8, 8, 48, 64, 3, 3, 1, 1)          # channels not a multiple of 64
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_ARG"):
        call("primia_conv2d_fwd", bad, x, x, x, _lib.PRIMIA_BF16)
    good = ConvDesc.make(1, 8, 8, 64, 64, 3, 3, 1, 1)
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_ARG"):
        call("primia_conv2d_fwd", good, None, x, x, _lib.PRIMIA_BF16)   # null pointer
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_ARG"):
        call("primia_conv2d_fwd", good, x, x, x, 7)                      # unknown dtype
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_ARG"):
        call("primia_trunc_div", torch.zeros(4, dtype=I64, device=cuda), 0, torch.zeros(4, dtype=I64, device=cuda), 4)
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_WORKSPACE"):
        y = torch.zeros(64, 64, device=cuda)
        v = torch.zeros(64, device=cuda)
        call("primia_bn_fwd_train", y, None, y, v, v, v, v, v, v, 64, 64, 1e-5, 0.1, 1,
             torch.empty(16, dtype=torch.uint8, device=cuda), 16, _lib.PRIMIA_F32)
    stem = ConvDesc.make(1, 32, 32, 4, 64, 7, 7, 2, 3)
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_ARG"):
        call("primia_conv2d_dgrad", stem, x, x, x, 0, _lib.PRIMIA_BF16)  # the stem has no data gradient


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 50001])
def test_dif_ragged_sizes(cuda, n):
    """Comparison counts that are not multiples of the wave / block size; 50001 crosses the
    reference's MULTI_LIMIT (fss.py:44) where it switches to a process pool."""
    g = torch.Generator(device=cuda).manual_seed(n)
    alpha = torch.randint(0, 2 ** 32, (n,), dtype=I64, device=cuda, generator=g)
    s0 = torch.randint(-2 ** 63, 2 ** 63 - 1, (2, 2, n), dtype=I64, device=cuda, generator=g)
    s0[:, 0] &= 0x7FFFFFFFFFFFFFFF
    x = torch.randint(0, 2 ** 32, (n,), dtype=I64, device=cuda, generator=g)
    x[: n // 3] = alpha[: n // 3]                     # exercise x == alpha
    bits = torch.empty(32, n, dtype=torch.uint8, device=cuda)
    cs = torch.empty(32, 2, n, dtype=I64, device=cuda)
    cg = torch.empty(32, 2, n, dtype=I64, device=cuda)
    leaf = torch.empty(33, n, dtype=torch.int32, device=cuda)
    call("primia_dif_keygen", alpha, s0, bits, cg, cs, leaf, n)
    xm = x.to(torch.int32)  # low 32 bits
    outs = []
    for b in range(2):
        o = torch.empty(n, dtype=I64, device=cuda)
        call("primia_dif_eval", b, xm, s0[b].contiguous(), bits, cg, cs, leaf, o, n)
        outs.append(o)
    bit = (outs[0] + outs[1]).cpu()
    assert torch.equal(bit, (x <= alpha).to(I64).cpu())
    if n <= 257:  # and bit-exact per share against the oracle
        _, keys = S.dif_keygen(alpha.cpu().numpy().astype("uint64"), s0.cpu().numpy().view("uint64"))
        for b in range(2):
            ref = S.dif_eval(b, x.cpu().numpy().astype("uint64"), keys[b])
            assert (outs[b].cpu().numpy() == ref).all()


def test_batch_one_training_step(cuda):
    from primia_amd.engine import ResNet18Engine

    eng = ResNet18Engine(1, 3, 3, 64, "max", dtype=torch.float32, device=cuda, norm="group")
    torch.manual_seed(0)
    eng.init_weights()
    x = torch.randn(1, 3, 64, 64)
    eng.forward(x.to(cuda))
    loss = eng.loss_backward(torch.tensor([2], device=cuda))
    eng.sgd_step(1e-3, 0.0)
    assert torch.isfinite(loss).all() and torch.isfinite(eng.flat).all()


def test_xent_out_of_range_label_poisons_instead_of_reading_out_of_bounds(cuda):
    """torch raises for a label outside [0, C); the kernel cannot, so the loss and that sample's gradient row are NaN
    (loud downstream) and no memory outside cw / logits is touched."""
    import torch

    from primia_amd._lib import call

    N, C = 5, 3
    logits = torch.randn(N, C, device=cuda)
    cw = torch.tensor([0.5, 1.0, 2.0], device=cuda)
    loss, dl = torch.zeros(1, device=cuda), torch.zeros(N, C, device=cuda)
    for bad in (3, -1, 2 ** 40):
        tgt = torch.tensor([0, 1, bad, 2, 1], device=cuda)
        call("primia_xent_hard", logits, tgt, cw, loss, dl, N, C)
        assert torch.isnan(loss).all() and torch.isnan(dl[2]).all()
    call("primia_xent_hard", logits, torch.tensor([0, 1, 2, 2, 1], device=cuda), cw, loss, dl, N, C)
    assert torch.isfinite(loss).all() and torch.isfinite(dl).all()


def test_round3_fused_entry_points_refuse_what_they_do_not_serve(cuda):
    """The fused forms added in round 3 answer with a query / an error code, never with a wrong result: the step tail
    for a layer the tiles do not cover, the paired transition forward on shapes that are not a transition pair, empty
    SGD range lists; primia_sgd_step_ranges equals primia_sgd_step on the ranges it is given and leaves the rest."""
    import ctypes

    from primia_amd._lib import query

    stem = ConvDesc.make(1, 32, 32, 4, 64, 7, 7, 2, 3)
    c33 = ConvDesc.make(2, 8, 8, 64, 128, 3, 3, 2, 1)
    c11 = ConvDesc.make(2, 8, 8, 64, 128, 1, 1, 2, 0)
    c33s1 = ConvDesc.make(2, 8, 8, 64, 128, 3, 3, 1, 1)
    assert query("primia_conv_sgd_fusable", stem, 3) == 0
    assert query("primia_conv_sgd_fusable", c33, 64) == 1 and query("primia_conv_sgd_fusable", c11, 64) == 1
    assert query("primia_conv_fwd_pair_ok", c33, c11, _lib.PRIMIA_BF16) == 1
    assert query("primia_conv_fwd_pair_ok", c33, c11, _lib.PRIMIA_F32) == 0      # bf16 only
    assert query("primia_conv_fwd_pair_ok", c33s1, c11, _lib.PRIMIA_BF16) == 0   # not a stride-2 pair
    x = torch.zeros(2 * 8 * 8 * 64, dtype=torch.bfloat16, device=cuda)
    w = torch.zeros(128 * 9 * 64, dtype=torch.bfloat16, device=cuda)
    y = torch.zeros(2 * 8 * 8 * 128, dtype=torch.bfloat16, device=cuda)
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_UNSUPPORTED"):
        call("primia_conv2d_fwd_stats_pair", c33s1, x, w, y, None, c11, w, y, None, _lib.PRIMIA_BF16)
    # the step tail refuses the stem filter (the caller asks primia_conv_sgd_fusable first)
    one = lambda t: (ctypes.c_void_p * 1)(ctypes.c_void_p(t.data_ptr()))
    f = torch.zeros(64 * 256, device=cuda)
    with pytest.raises(PrimiaError, match="PRIMIA_ERR_UNSUPPORTED"):
        call("primia_conv_sgd_step_many", (ConvDesc * 1)(stem), (ctypes.c_int * 1)(3), one(f), one(f), one(f), one(x), one(x), 1,
             0.1, 0.0, _lib.PRIMIA_BF16)
    # ranges: nothing to do is fine; two ranges = primia_sgd_step on each, the gap untouched
    g = torch.Generator(device=cuda).manual_seed(3)
    p0 = torch.randn(1000, device=cuda, generator=g)
    gr = torch.randn(1000, device=cuda, generator=g)
    call("primia_sgd_step_ranges", p0, gr, None, None, 0, 0.1, 0.0)
    a, b = p0.clone(), p0.clone()
    call("primia_sgd_step_ranges", a, gr, (ctypes.c_int64 * 2)(0, 500), (ctypes.c_int64 * 2)(128, 333), 2, 1e-2, 5e-4)
    call("primia_sgd_step", b[:128], gr[:128], 128, 1e-2, 5e-4)
    bb = b[500:833].clone(); gg = gr[500:833].clone()       # (16-byte aligned copies for the flat kernel)
    call("primia_sgd_step", bb, gg, 333, 1e-2, 5e-4)
    b[500:833] = bb
    assert torch.equal(a, b)
