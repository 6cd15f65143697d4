"""Mint the secure-path golden vectors from the REFERENCE's own code (build container only).

    python tests/golden/make_secure_golden.py

Loads, straight from /root/reference:
  * syft/frameworks/torch/nn/functional.py  (stubs for `syft`, `syft.generic.*`): _pre_conv,
    _post_conv, _pre_pool, _post_pool — the im2col / pool-unroll layouts;
  * syft/frameworks/torch/mpc/fss.py with the three shims of SURVEY.md §8c: a `shaloop` stand-in
    (independent SHA-256/512 of each 16-byte row), `np.bool = np.bool_`, and int64 +-1 in place of
    `(-1) ** <uint64 array>` at the three keygen sites NumPy 2 refuses: DIF/DPF keygen + eval.
It checks oracle/secure_oracle.py against them and writes small fixtures.  Pieces that cannot be
loaded (spdz.py, beaver.py, additive_shared.py, precision.py need the whole syft package) are
restated in the oracle from source; their fixtures are produced by the oracle and marked so.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF = "/root/reference/syft/frameworks/torch"

from oracle import secure_oracle as S  # noqa: E402


def _stub_syft():
    syft = types.ModuleType("syft")
    syft.Plan = object
    generic = types.ModuleType("syft.generic")
    utils = types.ModuleType("syft.generic.utils")
    utils.allow_command = lambda f: f
    utils.remote = lambda f, location=None: f
    fw = types.ModuleType("syft.generic.frameworks")
    ftypes = types.ModuleType("syft.generic.frameworks.types")
    ftypes.FrameworkTensor = torch.Tensor
    exc = types.ModuleType("syft.exceptions")
    exc.EmptyCryptoPrimitiveStoreError = type("EmptyCryptoPrimitiveStoreError", (Exception,), {})
    workers = types.ModuleType("syft.workers")
    wsc = types.ModuleType("syft.workers.websocket_client")
    wsc.WebsocketClientWorker = type("WebsocketClientWorker", (), {})
    for name, mod in [("syft", syft), ("syft.generic", generic), ("syft.generic.utils", utils),
                      ("syft.generic.frameworks", fw), ("syft.generic.frameworks.types", ftypes),
                      ("syft.exceptions", exc), ("syft.workers", workers),
                      ("syft.workers.websocket_client", wsc)]:
        sys.modules[name] = mod
    shaloop = types.ModuleType("shaloop")

    def sha256_loop_func(x, out):
        out[...] = S.sha_loop(np.ascontiguousarray(x), 256)

    def sha512_loop_func(x, out):
        out[...] = S.sha_loop(np.ascontiguousarray(x), 512)

    shaloop.sha256_loop_func, shaloop.sha512_loop_func = sha256_loop_func, sha512_loop_func
    sys.modules["shaloop"] = shaloop


def load_functional():
    spec = importlib.util.spec_from_file_location("ref_functional", f"{REF}/nn/functional.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_fss():
    src = open(f"{REF}/mpc/fss.py", encoding="utf-8").read()
    shims = [
        ("CW_n = (-1) ** t[n, 1] * (", "CW_n = (1 - 2 * t[n, 1].astype(np.int64)) * ("),
        ("CW_leaf[i] = (-1) ** τ[i + 1, 1] * (", "CW_leaf[i] = (1 - 2 * τ[i + 1, 1].astype(np.int64)) * ("),
        ("CW_leaf[n] = (-1) ** t[n, 1] * (", "CW_leaf[n] = (1 - 2 * t[n, 1].astype(np.int64)) * ("),
    ]
    for old, new in shims:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    if not hasattr(np, "bool"):
        np.bool = np.bool_
    m = types.ModuleType("ref_fss")
    exec(compile(src, f"{REF}/mpc/fss.py", "exec"), m.__dict__)
    return m


def draw_fss_randomness(n):
    """The draws DIF/DPF.keygen make from np.random, in order (fss.py:346,357 / :288,298;
    randbit :495-501)."""
    alpha = np.random.randint(0, 2 ** 32, size=(n,), dtype=np.uint64)
    s0 = np.random.randint(0, 2 ** 64, size=(2, 2, n), dtype=np.uint64)
    s0[:, 0] = s0[:, 0] % 2 ** 63
    return alpha, s0


def ref_key_to_dict(kind, key_b):
    """Reference key tuple (alpha, s0_b, *_CW, leaf) -> the oracle's dict layout."""
    alpha, s0, *cw, leaf = key_b
    if kind == "dif":
        bits = np.stack([np.stack([c[0], c[1], c[2], c[3]]).astype(np.uint8) for c in cw])
        return dict(s0=s0, bits=bits, cw_sigma=np.stack([c[4] for c in cw]), cw_s=np.stack([c[5] for c in cw]),
                    cw_leaf=leaf)
    bits = np.stack([np.stack([c[0], c[1]]).astype(np.uint8) for c in cw])
    return dict(s0=s0, bits=bits, cw_s=np.stack([c[2] for c in cw]), cw_n=leaf)


def mint_fss(fss):
    out = {}
    n = 256
    for kind in ("dif", "dpf"):
        np.random.seed(1234 if kind == "dif" else 4321)
        ref_keys = (fss.DIF if kind == "dif" else fss.DPF).keygen(n_values=n)
        alpha_ref, s00, s01, *rest = ref_keys
        np.random.seed(1234 if kind == "dif" else 4321)
        alpha, s0 = draw_fss_randomness(n)
        assert np.array_equal(alpha, alpha_ref) and np.array_equal(s0[0], s00) and np.array_equal(s0[1], s01)
        _, keys = (S.dif_keygen if kind == "dif" else S.dpf_keygen)(alpha, s0)
        for b in range(2):
            rk = ref_key_to_dict(kind, (alpha_ref, [s00, s01][b], *rest))
            for f in rk:
                assert np.array_equal(np.asarray(rk[f]).astype(np.int64), np.asarray(keys[b][f]).astype(np.int64)), (kind, b, f)
        # inputs: random, plus the edges x = alpha, alpha +- 1 (mod 2^32), 0, 2^32 - 1
        rng = np.random.default_rng(7)
        x = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64)
        x[:40] = alpha[:40]
        x[40:80] = (alpha[40:80] + 1) % 2 ** 32
        x[80:120] = (alpha[80:120] - 1) % 2 ** 32
        x[120] = 0
        x[121] = 2 ** 32 - 1
        ev = fss.DIF.eval if kind == "dif" else fss.DPF.eval
        res = []
        for b in range(2):
            kb = (ref_keys[1 + b], *ref_keys[3:])
            ref_out = ev(b, x.copy(), *kb)
            mine = (S.dif_eval if kind == "dif" else S.dpf_eval)(b, x, keys[b])
            assert np.array_equal(ref_out.astype(np.int64), mine), (kind, b)
            res.append(mine)
        bit = S.radd(res[0], res[1])
        want = (x <= alpha) if kind == "dif" else (x == alpha)
        assert np.array_equal(bit, want.astype(np.int64)), kind
        out[f"{kind}.alpha"] = alpha
        out[f"{kind}.s0"] = s0
        out[f"{kind}.x"] = x
        out[f"{kind}.out0"], out[f"{kind}.out1"] = res
        out[f"{kind}.leaf"] = keys[0]["cw_leaf"] if kind == "dif" else keys[0]["cw_n"]
        out[f"{kind}.cw_s_sum"] = np.array([int(keys[0]["cw_s"].astype(np.uint64).sum(dtype=np.uint64))], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "secure_fss.npz"), **out)
    print("fss fixtures ok")


def mint_layouts(F):
    out = {}
    g = torch.Generator().manual_seed(3)
    cases = {"stem": ((1, 3, 16, 16), (8, 3, 7, 7), 2, 3), "c3": ((1, 4, 9, 9), (6, 4, 3, 3), 1, 1),
             "s2": ((2, 4, 8, 8), (5, 4, 3, 3), 2, 1), "ds": ((1, 4, 8, 8), (8, 4, 1, 1), 2, 0)}
    for name, (xs, ws, stride, pad) in cases.items():
        x = torch.randint(-2 ** 62, 2 ** 62, xs, generator=g, dtype=torch.int64)
        w = torch.randint(-2 ** 62, 2 ** 62, ws, generator=g, dtype=torch.int64)
        bias = torch.randint(-2 ** 62, 2 ** 62, (ws[0],), generator=g, dtype=torch.int64)
        im, wr, *params = F._pre_conv(x, w, None, stride, pad)
        oim, owr, (B, O, Ho, Wo) = S.pre_conv(x.numpy(), w.numpy(), stride, pad)
        assert np.array_equal(im.numpy(), oim) and np.array_equal(wr.numpy(), owr)
        assert [int(p) for p in params] == [B, O, Ho, Wo]
        res = torch.randint(-2 ** 62, 2 ** 62, (B, Ho * Wo, O), generator=g, dtype=torch.int64)
        # bias needs .is_wrapper in the reference; a plain tensor takes the `res += bias` branch
        bias_t = bias.clone()
        bias_t.is_wrapper = False
        res_t = res.clone()
        res_t.is_wrapper = False
        post = F._post_conv(bias_t, res_t, *params)
        opost = S.post_conv(bias.numpy(), res.numpy(), B, O, Ho, Wo)
        assert np.array_equal(post.numpy(), opost)
        out[f"conv.{name}.x"], out[f"conv.{name}.w"] = x.numpy(), w.numpy()
        out[f"conv.{name}.im_sum"] = np.array([int(oim.astype(np.uint64).sum(dtype=np.uint64))], dtype=np.uint64)
        out[f"conv.{name}.im_head"] = oim[:, :3, :]
        out[f"conv.{name}.res"], out[f"conv.{name}.bias"], out[f"conv.{name}.post"] = res.numpy(), bias.numpy(), opost
        out[f"conv.{name}.meta"] = np.array([stride, pad])
    for name, (xs, k, stride, pad) in {"p3": ((1, 2, 8, 8), 3, 2, 1), "p7": ((1, 4, 7, 7), 7, 7, 0),
                                       "p2": ((2, 3, 6, 6), 2, 2, 0)}.items():
        x = torch.randint(-2 ** 62, 2 ** 62, xs, generator=g, dtype=torch.int64)
        im, *params = F._pre_pool(x, k, stride, pad)
        oim, (B, C, Ho, Wo) = S.pre_pool(x.numpy(), k, stride, pad)
        assert np.array_equal(im.numpy(), oim)
        assert [int(p) for p in params[:4]] == [B, C, Ho, Wo]
        out[f"pool.{name}.x"], out[f"pool.{name}.im"] = x.numpy(), oim
        out[f"pool.{name}.meta"] = np.array([k, stride, pad])
    np.savez_compressed(os.path.join(HERE, "secure_layouts.npz"), **out)
    print("layout fixtures ok")


def mint_restated():
    """Fixtures for the pieces restated from source (spdz.py / beaver.py / additive_shared.py /
    precision.py cannot be imported): produced by the oracle itself, randomness included."""
    rng = np.random.default_rng(11)

    def r64(*shape):
        return rng.integers(-2 ** 63, 2 ** 63 - 1, size=shape, dtype=np.int64)

    out = {}
    # truncation toward zero incl. the edge cases
    x = np.concatenate([r64(64), np.array([0, 1, -1, 10 ** 16, -10 ** 16, 10 ** 16 - 1, -(10 ** 16) + 1,
                                           np.iinfo(np.int64).max, np.iinfo(np.int64).min, 49, -49, 48, -48])])
    out["trunc.x"] = x
    for d in (10 ** 16, 10 ** 3, 20, 49):
        out[f"trunc.d{d}"] = S.trunc_div(x, d)
        ref = torch.div(torch.from_numpy(x), d, rounding_mode="trunc").numpy()
        assert np.array_equal(out[f"trunc.d{d}"], ref)
    # Beaver mul / matmul
    for op, xs, ys in (("mul", (5, 7), (5, 7)), ("mul", (6, 4), (4,)), ("matmul", (1, 9, 12), (12, 5))):
        x, y = [r64(*xs), r64(*xs)], [r64(*ys), r64(*ys)]
        a, b = r64(*xs), r64(*ys)
        triple = S.build_triple(op, a, b, r64(*xs), r64(*ys), r64(*np.broadcast_shapes(xs, ys)) if op == "mul"
                                else r64(*(xs[:-1] + ys[-1:])))
        z = S.beaver(op, x, y, triple)
        xv, yv = S.radd(*x), S.radd(*y)
        want = S.rmul(xv, yv) if op == "mul" else S.rmatmul(xv, yv)
        assert np.array_equal(S.radd(*z), want)
        tag = f"beaver.{op}.{'x'.join(map(str, xs))}"
        for j in range(2):
            out[f"{tag}.x{j}"], out[f"{tag}.y{j}"], out[f"{tag}.z{j}"] = x[j], y[j], z[j]
            out[f"{tag}.a{j}"], out[f"{tag}.b{j}"], out[f"{tag}.c{j}"] = triple[j]
    np.savez_compressed(os.path.join(HERE, "secure_restated.npz"), **out)
    print("restated fixtures ok")


def main():
    _stub_syft()
    mint_layouts(load_functional())
    mint_fss(load_fss())
    mint_restated()


if __name__ == "__main__":
    main()
