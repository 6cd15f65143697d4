"""CPU: the secure-path oracle against the golden vectors minted from the reference's own
nn/functional.py and mpc/fss.py (tests/golden/make_secure_golden.py), plus protocol-level
properties (reconstruction = plaintext function) that hold for any randomness."""
import hashlib
import os

import numpy as np
import pytest

from oracle import secure_oracle as S


def test_sha_loop_matches_hashlib():
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (64, 16), dtype=np.uint8)
    for kind, h in ((256, hashlib.sha256), (512, hashlib.sha512)):
        ref = np.stack([np.frombuffer(h(r.tobytes()).digest(), dtype=np.uint8) for r in x])
        assert np.array_equal(S.sha_loop(x, kind), ref)


def test_layout_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "secure_layouts.npz"))
    for name in ("stem", "c3", "s2", "ds"):
        stride, pad = [int(v) for v in g[f"conv.{name}.meta"]]
        im, wr, (B, O, Ho, Wo) = S.pre_conv(g[f"conv.{name}.x"], g[f"conv.{name}.w"], stride, pad)
        assert int(im.astype(np.uint64).sum(dtype=np.uint64)) == int(g[f"conv.{name}.im_sum"][0])
        assert np.array_equal(im[:, :3, :], g[f"conv.{name}.im_head"])
        post = S.post_conv(g[f"conv.{name}.bias"], g[f"conv.{name}.res"], B, O, Ho, Wo)
        assert np.array_equal(post, g[f"conv.{name}.post"])
    for name in ("p3", "p7", "p2"):
        k, stride, pad = [int(v) for v in g[f"pool.{name}.meta"]]
        im, _ = S.pre_pool(g[f"pool.{name}.x"], k, stride, pad)
        assert np.array_equal(im, g[f"pool.{name}.im"])


@pytest.mark.parametrize("kind", ["dif", "dpf"])
def test_fss_golden(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, "secure_fss.npz"))
    alpha, s0, x = g[f"{kind}.alpha"], g[f"{kind}.s0"], g[f"{kind}.x"]
    _, keys = (S.dif_keygen if kind == "dif" else S.dpf_keygen)(alpha, s0)
    assert np.array_equal(keys[0]["cw_leaf" if kind == "dif" else "cw_n"], g[f"{kind}.leaf"])
    assert int(keys[0]["cw_s"].sum(dtype=np.uint64)) == int(g[f"{kind}.cw_s_sum"][0])
    ev = S.dif_eval if kind == "dif" else S.dpf_eval
    outs = [ev(b, x, keys[b]) for b in range(2)]
    assert np.array_equal(outs[0], g[f"{kind}.out0"]) and np.array_equal(outs[1], g[f"{kind}.out1"])
    want = (x <= alpha) if kind == "dif" else (x == alpha)
    assert np.array_equal(S.radd(*outs), want.astype(np.int64))


def test_restated_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "secure_restated.npz"))
    for d in (10 ** 16, 10 ** 3, 20, 49):
        assert np.array_equal(S.trunc_div(g["trunc.x"], d), g[f"trunc.d{d}"])
    for tag, op in (("beaver.mul.5x7", "mul"), ("beaver.mul.6x4", "mul"), ("beaver.matmul.1x9x12", "matmul")):
        x = [g[f"{tag}.x{j}"] for j in range(2)]
        y = [g[f"{tag}.y{j}"] for j in range(2)]
        t = [(g[f"{tag}.a{j}"], g[f"{tag}.b{j}"], g[f"{tag}.c{j}"]) for j in range(2)]
        z = S.beaver(op, x, y, t)
        for j in range(2):
            assert np.array_equal(z[j], g[f"{tag}.z{j}"])


def test_fix_precision_matches_torch():
    import torch

    x = torch.randn(1000) * 3
    for pf in (16, 3):
        ref = (x * 10 ** pf).long()
        assert np.array_equal(S.fix_encode(x.numpy(), 10, pf), ref.numpy())
        assert np.array_equal(S.fix_decode(ref.numpy(), 10, pf), (ref.float() / 10 ** pf).numpy())
