"""Mint the encrypted-inference golden vectors by RUNNING the reference's own code (build container only).

    python tests/golden/make_secure_ref_golden.py

`ref_runtime.Runtime` loads PySyft's spdz.py / beaver.py / primitives.py / fss.py / additive_shared.py /
precision.py / nn/functional.py and PriMIA's torchlib/models.py from /root/reference and executes them
(stand-ins only for workers, pointers and the hook — see its docstring).  Two fixtures are written:

  secure_ref_ops.npz      one small case per operation of SURVEY.md §8a S2-S11 (fresh sharing, Beaver mul /
                          broadcast mul / matmul + per-share truncation, FSS ReLU, the 80-step Newton
                          reciprocal, eval BatchNorm, conv2d in its four ResNet shapes, 3x3/2 max pool,
                          AvgPool2d, linear), each at precision_fractional 3 and 16: plaintext inputs, the
                          provider's complete primitive stream in request order, the output shares.
  secure_ref_forward.npz  S12: the reference's full ResNet-18 (8 blocks, real widths) at 32x32 through
                          model.fix_precision().share() -> stem swap -> model(data) (inference.py:279-321) at
                          precision_fractional 3 and 16: seeds, a (descriptor, checksum) row per primitive and
                          the output shares.  tests/ref_stream.py re-draws the stream from the seeds; every
                          re-drawn primitive is compared with the reference's HERE before the file is written.

oracle/secure_oracle.py is checked bit for bit against every case while minting.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_runtime as R  # noqa: E402
import ref_stream as RS  # noqa: E402
import secure_cases as C  # noqa: E402
from oracle import secure_oracle as S  # noqa: E402


def wrap_fpt(rt, like, ast_):
    return rt.FPT(**like.get_class_attributes()).on(ast_, wrap=False)


def run_reference_op(rt, name, xs):
    """The reference call for each case of tests/secure_cases.py."""
    F = rt.F
    if name in ("mul", "mul_bcast"):
        return xs[0] * xs[1]                                   # precision.py:264-366
    if name == "matmul":
        return xs[0].matmul(xs[1])                             # precision.py:419-463
    if name == "relu":
        return wrap_fpt(rt, xs[0], xs[0].child.relu())         # additive_shared.py:922-925
    if name == "newton":
        return xs[0].reciprocal(method="newton")               # precision.py:507-518
    if name == "bn_eval":
        x, mean, var, w, b = xs
        return F.batch_norm(x, mean, var, w, b, False, 0.1, 1e-5)   # nn/functional.py:44-75
    if name.startswith("conv_"):
        _, stride, pad = C.CONV_SHAPES[name]
        return F.conv2d(xs[0], xs[1], None, stride, pad, 1, 1)      # nn/functional.py:204-308
    if name == "maxpool":
        return F.max_pool2d(xs[0], 3, 2, 1, 1, False, False)        # nn/functional.py:419-508
    if name == "avgpool":
        return F.avg_pool2d(xs[0], 7, 7, 0, False, True, None)
    if name == "linear":
        return rt.FPT.torch.addmm(xs[2], xs[0], xs[1].t())          # nn/functional.py:10-14 -> precision.py:822-825
    raise KeyError(name)


def mint_ops():
    out = {}
    for pf in (3, 16):
        for name in C.CASES:
            rt = R.Runtime()
            inputs = C.make_inputs(name)
            torch.manual_seed(1000 + pf)
            np.random.seed(2000 + pf)
            xs = [rt.fix_share(torch.from_numpy(x), pf) for x in inputs]
            res = run_reference_op(rt, name, xs)
            shares = rt.shares_of(res)
            # the oracle on the reference's stream
            ctx = S.OracleContext(S.ReplayDealer(rt.log), 10, pf)
            mine = C.run_case(ctx, name, inputs, S.fix_encode)
            assert ctx.dealer.pos == len(rt.log), (name, ctx.dealer.pos, len(rt.log))
            for j in range(2):
                assert np.array_equal(np.asarray(mine[j]).reshape(shares[j].shape), shares[j]), (name, pf, j)
            tag = f"{name}.p{pf}"
            RS.pack_log(tag, rt.log, out)
            for j in range(2):
                out[f"{tag}/out{j}"] = shares[j]
            out[f"{tag}/decoded"] = rt.decode(res).numpy()
            print(f"  {tag}: {len(rt.log)} primitives, oracle bit-identical")
    np.savez_compressed(os.path.join(HERE, "secure_ref_ops.npz"), **out)
    print("secure_ref_ops.npz written")


def mint_forward():
    out = {}
    M = R.load_reference_models()
    for pf in (3, 16):
        rt = R.Runtime()
        sd, image = C.forward_model_and_image()
        model = M.resnet18(pretrained=False, num_classes=3, in_channels=3, adptpool=False, input_size=C.FWD_SIZE,
                           pooling="max")
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.eval()
        with torch.no_grad():
            model.pool, model.relu = model.relu, model.pool
            plain = model(torch.from_numpy(image)).numpy()
            model.pool, model.relu = model.relu, model.pool
        tseed, nseed = 31000 + pf, 32000 + pf
        torch.manual_seed(tseed)
        np.random.seed(nseed)
        rt.share_model(model, pf)                                  # inference.py:279-287
        model.pool, model.relu = model.relu, model.pool            # :289
        data = rt.fix_share(torch.from_numpy(image), pf)           # :303-308
        with rt.hooked(), torch.no_grad():
            res = model(data)                                      # :311
        shares = rt.shares_of(res)
        log = rt.log
        # (1) the stream re-drawn from the seeds equals what the reference produced, primitive by primitive
        stream = RS.RefStream(tseed, nseed)
        for i, e in enumerate(log):
            if e[0] == "mask":
                mine = stream.mask(e[1].shape)
                assert np.array_equal(mine[1], e[1]), i
            elif e[0] == "triple":
                mine = stream.triple(e[1], e[2][0][0].shape, e[2][0][1].shape)
                for j in range(2):
                    for u, v in zip(mine[2][j], e[2][j]):
                        assert np.array_equal(u, v), (i, j)
            else:
                mine = stream.dif(e[1])
                for u, v in zip(mine[2:], e[2:]):
                    assert np.array_equal(np.asarray(u).reshape(-1), np.asarray(v).reshape(-1)), i
        # (2) the oracle, in the reference's primitive order, reproduces the output shares
        ctx = S.OracleContext(S.ReplayDealer(log), 10, pf)
        mine = S.secure_resnet_forward(ctx, sd, image, batched_newton=False)
        assert ctx.dealer.pos == len(log)
        for j in range(2):
            assert np.array_equal(mine[j], shares[j]), (pf, j)
        tag = f"fwd.p{pf}"
        out[f"{tag}/seeds"] = np.array([tseed, nseed])
        out[f"{tag}/desc"] = np.array([RS.entry_descriptor(e) for e in log], dtype=np.int64)
        out[f"{tag}/sums"] = np.array([RS.entry_checksum(e) for e in log], dtype=np.uint64)
        for j in range(2):
            out[f"{tag}/out{j}"] = shares[j]
        out[f"{tag}/decoded"] = rt.decode(res).numpy()
        out[f"{tag}/plain"] = plain
        print(f"  {tag}: {len(log)} primitives re-drawn identically, oracle bit-identical; decoded "
              f"{out[f'{tag}/decoded'].ravel()} plaintext {plain.ravel()}")
    np.savez_compressed(os.path.join(HERE, "secure_ref_forward.npz"), **out)
    print("secure_ref_forward.npz written")


def fedavg_inputs(K=3):
    """K tiny client state dicts with the key mix of a real model (weights, BN buffers, num_batches_tracked)."""
    from collections import OrderedDict

    g = torch.Generator().manual_seed(77)
    sds = []
    for k in range(K):
        sds.append(OrderedDict([
            ("conv.weight", torch.randn(4, 3, 3, 3, generator=g) * 0.2),
            ("bn.weight", torch.rand(4, generator=g) + 0.5),
            ("bn.bias", torch.randn(4, generator=g) * 0.1),
            ("bn.running_mean", torch.randn(4, generator=g) * 0.1),
            ("bn.running_var", torch.rand(4, generator=g) + 0.5),
            ("bn.num_batches_tracked", torch.tensor(5 + k)),
            ("fc.weight", torch.randn(3, 4, generator=g) * 5.0),       # larger magnitudes too
            ("fc.bias", torch.tensor([0.0, -1e-4, 123.456])),
        ]))
    return sds


class TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 4, 3, bias=False)
        self.bn = torch.nn.BatchNorm2d(4)
        self.fc = torch.nn.Linear(4, 3)


def mint_fedavg():
    """aggregation() of torchlib/utils.py:1000-1092, executed from its file: plain and secure (fix_prec -> share
    between the K workers -> stack -> sum -> get -> float_prec), unweighted and weighted."""
    from collections import Counter
    from types import SimpleNamespace

    from oracle import train_oracle as O

    out = {}
    ids = ["alice", "bob", "charlie"]
    sds = fedavg_inputs(len(ids))
    for k, sd in enumerate(sds):
        for key, v in sd.items():
            out[f"in{k}/{key}"] = v.numpy()
    wts = {"alice": 0.2, "bob": 0.3, "charlie": 0.5}
    for secure in (False, True):
        for weighted in (False, True):
            for pf in ((3, 16) if secure else (0,)):
                rt = R.Runtime(parties=tuple(ids))
                g = {"torch": R.torch_namespace(rt), "np": np, "Counter": Counter}
                R._extract("/root/reference/torchlib/utils.py", ["aggregation"], g)
                torch.manual_seed(1)
                np.random.seed(1)
                local = TinyNet()
                local.bn.num_batches_tracked.fill_(9)
                models = {i: R.RemoteModel(rt, sd) for i, sd in zip(ids, sds)}
                args = SimpleNamespace(precision_fractional=pf)
                res = g["aggregation"](local, models, ids, rt.local_worker, args, None,
                                       weights=wts if weighted else None, secure=secure)
                got = res.state_dict()
                mine = (O.fedavg_secure(sds, [wts[i] for i in ids] if weighted else None, pf) if secure
                        else O.fedavg_plain(sds, [wts[i] for i in ids] if weighted else None))
                tag = f"{'secure' if secure else 'plain'}.{'w' if weighted else 'u'}.p{pf}"
                for key, v in got.items():
                    if key.endswith("num_batches_tracked"):
                        # load_state_dict of a plain dict without the key: BatchNorm's version-1 path fills it in —
                        # with 0 under the reference's torch 1.4, with the module's current value under torch 2
                        assert int(v) in (0, 9)
                        continue
                    assert torch.equal(mine[key], v), (tag, key)
                    out[f"{tag}/{key}"] = v.numpy()
                print(f"  fedavg {tag}: oracle bit-identical")
    np.savez_compressed(os.path.join(HERE, "fedavg_ref.npz"), **out)
    print("fedavg_ref.npz written")


def mint_mean_std():
    """The secure mean / std exchange of setup_pysyft (torchlib/utils.py:764-794), with the reference's own tensor
    classes doing the work: every worker's statistic goes through FixedPrecisionTensor.fix_precision() (defaults:
    base 10, 3 fractional digits), .share() between the workers, AdditiveSharingTensor `+=`, .get(),
    .float_precision(), and is divided by the number of workers.  (Only grid.search and the pointer .get() that
    fetches the shared tensor to the orchestrator are not executed — they move tensors, they do not change them.)"""
    from oracle import datapipe_oracle as D

    out = {}
    g = torch.Generator().manual_seed(21)
    for nw, ch in ((2, 1), (3, 3), (5, 3)):
        ids = tuple("w%d" % i for i in range(nw))
        rt = R.Runtime(parties=ids)
        torch.manual_seed(2)
        np.random.seed(2)
        means = [torch.rand(ch, generator=g) * 0.6 + 0.2 for _ in range(nw)]
        stds = [torch.rand(ch, generator=g) * 0.3 + 0.05 for _ in range(nw)]
        mean = rt.fix_share(means[0], precision_fractional=3)
        std = rt.fix_share(stds[0], precision_fractional=3)
        for m, s_ in zip(means[1:], stds[1:]):
            mean += rt.fix_share(m, precision_fractional=3)
            std += rt.fix_share(s_, precision_fractional=3)
        mean = rt.decode(mean) / len(stds)
        std = rt.decode(std) / len(stds)
        om, os_ = D.exchange_mean_std(means, stds)
        assert torch.equal(om, mean) and torch.equal(os_, std), (nw, ch)
        tag = f"w{nw}c{ch}"
        out[tag + "/means"], out[tag + "/stds"] = torch.stack(means).numpy(), torch.stack(stds).numpy()
        out[tag + "/mean"], out[tag + "/std"] = mean.numpy(), std.numpy()
        print(f"  mean/std exchange {tag}: oracle bit-identical")
    np.savez_compressed(os.path.join(HERE, "mean_std_ref.npz"), **out)
    print("mean_std_ref.npz written")


if __name__ == "__main__":
    todo = sys.argv[1:] or ["fedavg", "mean_std", "ops", "forward"]
    for name in todo:
        {"fedavg": mint_fedavg, "mean_std": mint_mean_std, "ops": mint_ops, "forward": mint_forward}[name]()
