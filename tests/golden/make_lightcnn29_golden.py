"""Generates tests/golden/lightcnn29_step.npz: one training step of the network entry point 1 trains — the Gluon LightCNN_29
(ref: lightcnn.py:6-133) under the train_efm.py loss (ref: train_efm.py:229-245: whole-matrix norm, CE on the anchors +
0.1 * TripletLoss(0.2), ones head-gradient) followed by one Adam update (ref: train_efm.py:212-214,245) — from the NumPy fp64
oracle (oracle/efm_oracle.py), after checking it against the independent torch-CPU restatement.

PARITY UNPINNED: MXNet cannot be installed here and the reference holds no fixture for this path, so the vectors come from the
restatement, not from the reference itself.  Inputs and weights are regenerated from the portable splitmix64 generator by
`inputs()` (never committed); the .npz holds only outputs (~60 KB).

    python tests/golden/make_lightcnn29_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import efm_oracle as O  # noqa: E402

BATCH, IMAGE, CLASSES = 4, 32, 16          # 4 anchors + 4 positives of 1x32x32
LR, WD, MARGIN, ALPHA = 0.00024, 0.00001, 0.2, 0.1   # ref: train_efm.py:200-204,213
FULL = ["g3_res_conv0_bias", "g5_res_conv1_bias", "g1_conv1_weight", "batchnorm0_gamma", "batchnorm0_beta", "dense1_bias"]
SLICED = ["g2_res_conv1_weight", "g4_res_conv0_weight", "g5_res_conv0_weight"]   # first 2 output channels of the SHARED convolutions


def inputs():
    shapes = O.lightcnn29_param_shapes(1, IMAGE, CLASSES)
    params = O.init_lightcnn29_params(shapes, 42)
    k = 0
    for name in params:   # non-trivial biases / BatchNorm affine so that every gradient path carries signal
        if name.endswith("_bias") or name.startswith("batchnorm0_"):
            k += 1
            params[name] = params[name] + O.uniform_pm(params[name].shape, 9000 + k, 0.1)
    x = O.uniform01(2 * BATCH * IMAGE * IMAGE, 1234).reshape(2 * BATCH, 1, IMAGE, IMAGE)
    labels = np.array([0, 1, 2, 3, 0, 1, 2, 3], dtype=np.int64)   # labels duplicated for the positives (ref: train_efm.py:95-100)
    neg = np.array([1, 0, 3, 2], dtype=np.int32)
    return params, x, labels, neg


def main():
    import torch
    from oracle import efm_oracle_torch as OT
    params, x, labels, neg = inputs()
    r = O.train_efm_step(params, x, labels, neg, MARGIN, ALPHA)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    out, fc, tl, idl, loss = OT.train_efm_step(tp, torch.tensor(x), torch.tensor(labels), torch.tensor(neg.astype(np.int64)), MARGIN, ALPHA)
    assert np.abs(out.numpy() - r["out"]).max() < 1e-10 and np.abs(fc.numpy() - r["fc1_out"]).max() < 1e-9
    for k in params:
        g = r["grads"][k]
        assert np.abs(tp[k].grad.numpy() - g).max() <= 1e-9 * np.abs(g).max(), k
    names = sorted(params)
    fix = {"out": r["out"], "fc1_out": r["fc1_out"], "tl": r["tl"], "id": r["id"], "loss": r["loss"],
           "bn_mean": r["bn_mean"], "bn_var": r["bn_var"], "names": np.array(names),
           "grad_abs_sums": np.array([np.abs(r["grads"][k]).sum() for k in names]),
           "grad_sums": np.array([r["grads"][k].sum() for k in names])}
    for k in FULL:
        fix["grad_" + k] = r["grads"][k]
        w, _, _ = O.adam_step(params[k], r["grads"][k], 0.0, 0.0, 1, LR, WD, 1.0 / BATCH)
        fix["adam_" + k] = w
    for k in SLICED:
        fix["grad2_" + k] = r["grads"][k][:2]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lightcnn29_step.npz")
    np.savez_compressed(path, **{k: (v if v.dtype.kind in "US" else v.astype(np.float64)) for k, v in fix.items()})
    print("wrote", path, os.path.getsize(path), "bytes; loss", r["loss"])


if __name__ == "__main__":
    main()
