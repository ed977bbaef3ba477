#!/usr/bin/env python
"""Generates tests/golden/lfw_roc.npz by RUNNING the reference's own LFW-protocol functions.

Only runs in the build container (needs /root/reference); the resulting vectors are plain data and travel with the repo.
The reference module feature_extraction/facenet_version/facenet.py cannot be imported as a whole (it imports tensorflow at
line 32), so the three pure-NumPy functions of the protocol — distance (412-426), calculate_roc (428-459) and
calculate_accuracy (461-471) — are taken from its source text with `ast` and executed unchanged, with the names they use
(np, math, KFold) supplied from the installed numpy / sklearn.  No reference text is stored in this repo.
"""
import ast
import math
import os

import numpy as np
from sklearn.model_selection import KFold

SRC = "/root/reference/feature_extraction/facenet_version/facenet.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_functions():
    tree = ast.parse(open(SRC).read())
    wanted = {"distance", "calculate_roc", "calculate_accuracy"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
    ns = {"np": np, "math": math, "KFold": KFold}
    exec(compile(mod, SRC, "exec"), ns)
    return ns


def main():
    ref = load_reference_functions()
    rng = np.random.default_rng(20261005)
    n, d = 600, 32
    e1 = rng.normal(size=(n, d))
    issame = rng.random(n) < 0.5
    e2 = np.where(issame[:, None], e1 + 2.5 * rng.normal(size=(n, d)), rng.normal(size=(n, d)))
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 /= np.linalg.norm(e2, axis=1, keepdims=True)
    out = {"emb1": e1, "emb2": e2, "issame": issame}
    for metric, thresholds in ((0, np.arange(0, 4, 0.01)), (1, np.arange(0, 1, 0.0025))):
        for sub in (False, True):
            tpr, fpr, acc = ref["calculate_roc"](thresholds, e1, e2, issame, nrof_folds=10, distance_metric=metric, subtract_mean=sub)
            key = "m%d_s%d" % (metric, int(sub))
            out[key + "_thresholds"] = thresholds
            out[key + "_tpr"], out[key + "_fpr"], out[key + "_acc"] = tpr, fpr, acc
    np.savez_compressed(os.path.join(HERE, "lfw_roc.npz"), **out)
    print("wrote lfw_roc.npz; accuracies:", {k: float(v.mean()) for k, v in out.items() if k.endswith("_acc")})


if __name__ == "__main__":
    main()
