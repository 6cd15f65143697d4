"""Mint the data-path golden vectors from the REFERENCE's own code.

Run in the build container only (needs /root/reference):
    python tests/golden/make_datapipe_golden.py

torchlib/utils.py and torchlib/dataloader.py import syft / albumentations / torchvision, which this image
does not have, so the modules cannot be imported whole.  The script parses the two files where they lie
under /root/reference, takes the definitions of MixUp, To_one_hot and calc_mean_std out of the syntax tree
and executes exactly those (with torch / random / tqdm supplied as globals) — the reference's own code,
run from its own files; nothing of it is copied into this repository.  Every output is checked bit for
bit against oracle/datapipe_oracle.py and stored, with its inputs, in tests/golden/datapipe.npz.
"""
import ast
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import datapipe_oracle as D  # noqa: E402


def extract(path, names, glob):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, path, "exec"), glob)
    return glob


def reference_split(n_items, num_workers):
    """The statements of data/server_simulation/distribute_data.py that deal the images to the workers (from the
    `worker_dirs = ...` assignment to the `for i in range(args.num_workers)` loop), executed from the file with
    ImageFolder replaced by an object of the requested length.  Returns the per-worker index lists."""
    import ast
    from random import seed, shuffle
    from types import SimpleNamespace

    path = "/root/reference/data/server_simulation/distribute_data.py"
    tree = ast.parse(open(path).read())
    main_if = [n for n in tree.body if isinstance(n, ast.If)][-1]
    names = [getattr(n.targets[0], "id", None) if isinstance(n, ast.Assign) else None for n in main_if.body]
    start = names.index("worker_dirs")
    stop = next(i for i in range(start, len(main_if.body))
                if isinstance(main_if.body[i], ast.For) and getattr(main_if.body[i].target, "id", "") == "i")
    body = main_if.body[start:stop + 1]

    class Folder:
        classes = ["a", "b", "c"]

        def __init__(self, src):
            pass

        def __len__(self):
            return n_items

    g = {"args": SimpleNamespace(num_workers=num_workers, train_data_src="."), "ImageFolder": Folder,
         "seed": seed, "shuffle": shuffle}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), g)
    return [g["worker_imgs"]["worker{:d}".format(i + 1)] for i in range(num_workers)]


def registration_loop(glob):
    """The statements of setup_pysyft that register one worker's dataset (torchlib/utils.py:694-734): from the
    `data, targets = [], []` assignment to the `del data, targets` statement, taken out of the function's syntax tree
    and executed with `args`, `dataset`, `MixUp`, `torch`, `tqdm`, `worker` supplied.  torch.utils.data.DataLoader is
    replaced by an object that replays given permutations (torch 1.4's shuffling cannot be reproduced under torch 2)."""
    path = "/root/reference/torchlib/utils.py"
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "setup_pysyft"][0]
    found = []

    class V(ast.NodeVisitor):
        def visit_For(self, node):
            for i, st in enumerate(node.body):
                if (isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Tuple)
                        and [getattr(e, "id", None) for e in st.targets[0].elts] == ["data", "targets"]):
                    j = next(k for k in range(i, len(node.body)) if isinstance(node.body[k], ast.Delete))
                    found.append(node.body[i:j + 1])
            self.generic_visit(node)

    V().visit(fn)
    assert len(found) == 1, len(found)
    exec(compile(ast.Module(body=found[0], type_ignores=[]), path, "exec"), glob)
    return glob["selected_data"], glob["selected_targets"]


def mint_registration(out, MixUp, To_one_hot, gen):
    from types import SimpleNamespace

    import tqdm as _tqdm

    class ReplayLoader:
        """Stands in for DataLoader(dataset, batch_size=1, shuffle=True): one given permutation per pass."""

        def __init__(self, dataset, orders):
            self.dataset, self.orders, self.k = dataset, orders, 0

        def __len__(self):
            return len(self.dataset)

        def __iter__(self):
            order = self.orders[self.k]
            self.k += 1
            for i in order:
                d, t = self.dataset[i]
                yield d.unsqueeze(0), t.unsqueeze(0)

    cases = [("mix_fixed", 7, 2, True, 0.3, 0.9, 11), ("mix_rand_lambda", 6, 3, True, None, 0.9, 12),
             ("mix_always", 5, 2, True, 0.499, 1.0, 13), ("onehot_only", 5, 2, False, None, 0.9, 14)]
    for name, n, reps, mix, lam, p, seed in cases:
        xs = torch.randn(n, 3, 6, 5, generator=gen)
        labels = torch.randint(0, 3, (n,), generator=gen)
        ys = To_one_hot(3)(labels)
        dataset = [(xs[i], ys[i]) for i in range(n)]
        pr = random.Random(seed)
        orders = [pr.sample(range(n), n) for _ in range(reps)] if mix else [list(range(n))] * reps
        args = SimpleNamespace(mixup=mix, weight_classes=True, mixup_lambda=lam, mixup_prob=p, repetitions_dataset=reps,
                               num_threads=0)
        holder = {}

        class FakeDL:       # torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=True, num_workers=...)
            def __new__(cls, ds, **kw):
                assert kw["batch_size"] == 1 and kw["shuffle"] is True
                return ReplayLoader(ds, orders)

        fake_torch = SimpleNamespace(**{k: getattr(torch, k) for k in ("stack", "tensor")},
                                     utils=SimpleNamespace(data=SimpleNamespace(DataLoader=FakeDL)))
        glob = {"args": args, "dataset": dataset, "MixUp": MixUp, "torch": fake_torch, "tqdm": SimpleNamespace(tqdm=lambda it, **kw: it),
                "worker": SimpleNamespace(id="alice", load_data=lambda x: None), "j": 0}
        # Tensor.tag exists only under PySyft's hook: the statements `selected_data.tag(...)` run against a no-op
        torch.Tensor.tag = lambda self, *a: self
        random.seed(seed)
        try:
            data, tg = registration_loop(glob)
        finally:
            del torch.Tensor.tag
        od, ot = D.register_federated(list(xs), list(ys), orders, mix, lam, p, rng=random.Random(seed))
        assert torch.equal(data, od) and torch.equal(tg, ot), name
        out[f"reg.{name}.x"], out[f"reg.{name}.labels"] = xs.numpy(), labels.numpy()
        out[f"reg.{name}.orders"] = np.array(orders)
        out[f"reg.{name}.meta"] = np.array([n, reps, int(mix), -1.0 if lam is None else lam, p, seed], dtype=np.float64)
        out[f"reg.{name}.data"], out[f"reg.{name}.targets"] = data.numpy(), tg.numpy()
    # ---- calc_class_weights: the reference function on worker-keyed loaders (federated, soft targets) and on one
    #      plain loader (vanilla, hard labels) -----------------------------------------------------------------------
    class Worker:
        def __init__(self, wid):
            self.id = wid

    g_cw = {"torch": torch, "tqdm": _tqdm, "warn": lambda *a, **k: None}
    extract("/root/reference/torchlib/utils.py", ["calc_class_weights"], g_cw)
    ref_cw = g_cw["calc_class_weights"]
    torch.Tensor.send = lambda self, *a: self          # PySyft pointers: the arithmetic is the tensors' own
    torch.Tensor.get = lambda self: self
    try:
        for name, fed, bs, sizes in [("fed_soft", True, 4, [(3, 4), (2, 4), (1, 3)]), ("vanilla_hard", False, 5, [(4, 5)]),
                                     ("fed_empty", True, 4, [])]:
            loaders = {}
            for wi, (nb, b) in enumerate(sizes):
                batches = []
                for _ in range(nb):
                    lab = torch.randint(0, 3, (b,), generator=gen)
                    if fed:
                        soft = To_one_hot(3)(lab) * 0.7 + 0.1          # soft rows whose arg-max is the label
                        batches.append((torch.zeros(b, 1), soft))
                    else:
                        batches.append((torch.zeros(b, 1), lab))
                loaders[Worker(f"w{wi}")] = batches
            args = SimpleNamespace(batch_size=bs, train_federated=fed, mixup=fed, weight_classes=True)
            cw = ref_cw(args, loaders if fed else (list(loaders.values())[0] if loaders else []), 3)
            ocw = D.calc_class_weights(list(loaders.values()), bs, 3, fed)
            assert torch.equal(cw, ocw), name
            for wi, batches in enumerate(loaders.values()):
                out[f"cw.{name}.w{wi}"] = np.stack([t.numpy() for _, t in batches])
            out[f"cw.{name}.n"] = np.array([len(loaders)])
            out[f"cw.{name}.cw"] = cw.numpy()
    finally:
        del torch.Tensor.send, torch.Tensor.get


def main():
    from typing import List, Optional, Tuple, Union

    import torch.utils.data as torchdata

    g_utils = {"torch": torch, "random": random.random, "List": List, "Optional": Optional, "Tuple": Tuple,
               "Union": Union}
    extract("/root/reference/torchlib/utils.py", ["MixUp", "To_one_hot"], g_utils)
    g_dl = {"torchdata": torchdata, "stack": torch.stack, "cat": torch.cat, "std_mean": torch.std_mean,
            "tqdm": lambda it, **kw: it, "os": os, "save": torch.save}
    extract("/root/reference/torchlib/dataloader.py", ["calc_mean_std"], g_dl)
    MixUp, To_one_hot, calc_mean_std = g_utils["MixUp"], g_utils["To_one_hot"], g_dl["calc_mean_std"]

    out = {}
    gen = torch.Generator().manual_seed(2024)
    # ---- MixUp: (name, L, fixed lambda, p, seed of random, tuple form) ------------------------------------
    # (the reference's constructor asserts 0 <= p <= 1, so p is always a number: 1.0 = always mix after drawing
    # one random(), 0.0 = always mix without drawing)
    cases = [("even", 6, 0.3, 1.0, 1, False), ("odd", 7, 0.499, 0.0, 2, False), ("rand_lam", 4, None, 1.0, 3, False),
             ("p_skip_or_mix_a", 4, 0.25, 0.5, 4, False), ("p_skip_or_mix_b", 4, 0.25, 0.5, 7, False),
             ("pair_tuple", 2, 0.7, 0.9, 5, True), ("pair_tuple_skip", 2, 0.7, 0.1, 8, True), ("single", 1, 0.3, 1.0, 6, False)]
    for name, L, lam, p, seed, as_tuple in cases:
        x = torch.randn(L, 3, 5, 4, generator=gen)
        y = To_one_hot(3)(torch.randint(0, 3, (L,), generator=gen))
        xin = tuple(t.unsqueeze(0) for t in x) if as_tuple else x
        yin = tuple(t.unsqueeze(0) for t in y) if as_tuple else y
        random.seed(seed)
        rx, ry = MixUp(λ=lam, p=p)((xin, yin))
        rs = random.Random(seed)
        ox, oy = D.mixup(xin, yin, lam, p, rng=rs)
        assert torch.equal(rx, ox) and torch.equal(ry, oy), name
        out[f"mixup.{name}.x"], out[f"mixup.{name}.y"] = x.numpy(), y.numpy()
        out[f"mixup.{name}.meta"] = np.array([L, -1.0 if lam is None else lam, p, seed, int(as_tuple)], dtype=np.float64)
        out[f"mixup.{name}.out_x"], out[f"mixup.{name}.out_y"] = rx.numpy(), ry.numpy()
    # ---- To_one_hot ------------------------------------------------------------------------------------------
    for name, arg in [("int", 2), ("list", [0, 2, 1, 1]), ("scalar_tensor", torch.tensor(1)),
                      ("vector", torch.tensor([2, 0, 0, 1, 2]))]:
        r = To_one_hot(3)(arg)
        assert torch.equal(r, D.to_one_hot(arg, 3)), name
        out[f"onehot.{name}.in"] = np.array(arg if not torch.is_tensor(arg) else arg.numpy())
        out[f"onehot.{name}.out"] = r.numpy()
    # ---- calc_mean_std ----------------------------------------------------------------------------------------
    for name, shape in [("rgb", (10, 3, 9, 7)), ("gray", (6, 1, 8, 8)), ("other", (5, 2, 4, 4))]:
        data = torch.randn(*shape, generator=gen) * 1.7 + 0.4
        ds = torchdata.TensorDataset(data, torch.zeros(shape[0]))
        m, s = calc_mean_std(ds)
        om, os_ = D.calc_mean_std(data)
        assert torch.equal(m, om) and torch.equal(s, os_), name
        out[f"meanstd.{name}.data"] = data.numpy()
        out[f"meanstd.{name}.mean"], out[f"meanstd.{name}.std"] = m.numpy(), s.numpy()
    # the IID round-robin split (distribute_data.py:58-70), executed from the reference's script
    for n_items, nw in ((5163, 3), (100, 8), (7, 2)):
        ref_split = reference_split(n_items, nw)
        assert ref_split == D.iid_round_robin_split(n_items, nw), (n_items, nw)
        out[f"split.{n_items}.{nw}"] = np.array([len(s) for s in ref_split] + [v for s in ref_split for v in s[:8]])
    # ---- federated registration with MixUp (setup_pysyft, utils.py:694-734) and class weights (utils.py:469-513),
    #      executed from the reference's file ----------------------------------------------------------------------
    mint_registration(out, MixUp, To_one_hot, gen)
    np.savez_compressed(os.path.join(HERE, "datapipe.npz"), **out)
    print("wrote", len(out), "arrays; the oracle reproduces the reference bit for bit")


if __name__ == "__main__":
    main()
