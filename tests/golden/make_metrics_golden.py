"""Mint the validation-table golden text from the REFERENCE's own stats_table (torchlib/utils.py:1295-1351).

Run in the build container only (needs /root/reference):
    python tests/golden/make_metrics_golden.py

Like make_datapipe_golden.py it takes the function definition out of the reference file's syntax tree and
executes it in place (torchlib/utils.py cannot be imported whole: syft / albumentations are absent).  The
inputs are sklearn's own confusion matrix and classification report for fixed label vectors; the rendered
table is stored in tests/golden/metrics.npz together with those label vectors.
"""
import ast
import os
import sys

import numpy as np
from sklearn import metrics as mt
from tabulate import tabulate

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))


def main():
    path = "/root/reference/torchlib/utils.py"
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "stats_table"]
    glob = {"tabulate": tabulate}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), glob)
    ref_table = glob["stats_table"]
    from primia_amd.torchlib_compat import stats_table

    out = {}
    rng = np.random.default_rng(4)
    for name, n, names in (("named", 60, ["normal", "bacterial", "viral"]), ("numbered", 25, None)):
        t = rng.integers(0, 3, size=n)
        p = np.where(rng.random(n) < 0.7, t, rng.integers(0, 3, size=n))
        cm = mt.confusion_matrix(t, p)
        rep = mt.classification_report(t, p, output_dict=True, zero_division=0)
        text = ref_table(cm, rep, roc_auc=0.8123, matthews_coeff=0.5678, class_names=names, epoch=7)
        assert text == stats_table(cm, rep, roc_auc=0.8123, matthews_coeff=0.5678, class_names=names, epoch=7), name
        out[f"{name}.target"], out[f"{name}.pred"] = t, p
        out[f"{name}.table"] = np.frombuffer(text.encode("utf-8"), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("wrote", list(out))


if __name__ == "__main__":
    main()
