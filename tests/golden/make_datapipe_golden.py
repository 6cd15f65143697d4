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
    np.savez_compressed(os.path.join(HERE, "datapipe.npz"), **out)
    print("wrote", len(out), "arrays; the oracle reproduces the reference bit for bit")


if __name__ == "__main__":
    main()
