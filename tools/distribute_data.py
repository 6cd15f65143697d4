#!/usr/bin/env python3
"""Deal an image folder to simulated clients — the job of the reference's data/server_simulation/distribute_data.py:
`<train_data_src>/<class>/<image>` becomes `<out>/worker<i>/<class>/<image>` (and the test folder `<out>/validation/...`),
the layout `train.py --train_federated --data_dir <out>` reads.

    python tools/distribute_data.py --train_data_src data/train --test_data_src data/test --num_workers 8 \
        [--out data/server_simulation] [-s] [--label_skew ALPHA] [--seed 0]

Default split = the reference's: indices shuffled with random.seed(0), dealt i::num_workers (IID;
primia_amd.datapipe.iid_round_robin_split, pinned against the reference script in tests/golden/datapipe.npz).
`--label_skew ALPHA` produces the non-IID shards BASELINE.json configs[2] names instead (every class is divided between
the clients in Dirichlet(ALPHA) proportions; primia_amd.datapipe.label_skew_split).  -s links instead of copying."""
import argparse
import os
import sys
from shutil import copyfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def scan(root):
    """torchvision.datasets.ImageFolder's listing (same function the training loader uses)."""
    ext = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    samples = []
    for ci, c in enumerate(classes):
        for dirpath, _, files in sorted(os.walk(os.path.join(root, c))):
            for f in sorted(files):
                if f.lower().endswith(ext) and not f.startswith("._"):
                    samples.append((os.path.join(dirpath, f), ci))
    return classes, samples


def place(src, dst, symbolic):
    if os.path.lexists(dst):
        os.remove(dst)
    if symbolic:
        os.symlink(os.path.abspath(src), dst)
    else:
        copyfile(src, dst)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-s", "--symbolic", action="store_true", help="Create symlinks instead of copying files.")
    ap.add_argument("--num_workers", default=3, type=int, help="How many servers should be simulated.")
    ap.add_argument("--train_data_src", default="../train", type=str, help="Source data folder for training data.")
    ap.add_argument("--test_data_src", default="../test", type=str, help="Source data folder for test data.")
    ap.add_argument("--out", default=".", type=str, help="where worker<i>/ and validation/ are created")
    ap.add_argument("--label_skew", default=None, type=float, help="Dirichlet alpha of a non-IID split (default: IID)")
    ap.add_argument("--seed", default=0, type=int)
    a = ap.parse_args(argv)
    # the splitters are host-side index arithmetic; importing the module does not need the GPU library
    from primia_amd.datapipe import iid_round_robin_split, label_skew_split

    classes, samples = scan(a.train_data_src)
    if a.label_skew is None:
        shards = iid_round_robin_split(len(samples), a.num_workers, a.seed)
    else:
        shards = label_skew_split([c for _, c in samples], a.num_workers, a.label_skew, a.seed)
    def rel(src, root, cls):
        """Path below the class directory (files of nested sub-folders keep their sub-path: equal basenames do not
        overwrite each other)."""
        return os.path.relpath(src, os.path.join(root, cls))

    def put(src, dst):
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if os.path.lexists(dst) and dst in written:
            raise SystemExit("two source images map to {:s}".format(dst))
        written.add(dst)
        place(src, dst, a.symbolic)

    written = set()
    for i, idcs in enumerate(shards):
        for c in classes:
            os.makedirs(os.path.join(a.out, "worker{:d}".format(i + 1), c), exist_ok=True)
        for idx in idcs:
            src, ci = samples[idx]
            put(src, os.path.join(a.out, "worker{:d}".format(i + 1), classes[ci], rel(src, a.train_data_src, classes[ci])))
    n_val = 0
    if a.test_data_src and os.path.isdir(a.test_data_src):
        tclasses, tsamples = scan(a.test_data_src)
        for c in tclasses:
            os.makedirs(os.path.join(a.out, "validation", c), exist_ok=True)
        for src, ci in tsamples:
            put(src, os.path.join(a.out, "validation", tclasses[ci], rel(src, a.test_data_src, tclasses[ci])))
        n_val = len(tsamples)
    print("{:d} training images -> {:s}; {:d} validation images".format(
        len(samples), ", ".join("worker{:d}: {:d}".format(i + 1, len(s)) for i, s in enumerate(shards)), n_val))
    return shards


if __name__ == "__main__":
    main()
