#!/usr/bin/env python
"""Freezes BASELINE configs[0]: EFM-29 on 64 synthetic 3x112x112 faces in the reference batch layout
[32 anchors ; 32 positives], 128-d head on row-normalised 342-d features, TripletLoss(0.2) with a fixed negative index vector,
cosine rows as train_efm.py:26-34 logs them — embeddings + loss only (forward), from the torch-CPU fp64 restatement.
Inputs and weights come from the portable splitmix64 generator; only the OUTPUTS are stored (64x128 embeddings as float32,
32 losses, 32 cosine pairs, per-image feature checksums: ~36 KB).  The reference has no fixture for this path (parity
unpinned): this file pins the oracle against drift and gives the HIP path a file-based target at the BASELINE geometry.
Takes ~1-2 minutes on 8 cores."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import efm_oracle as O  # noqa: E402
from oracle import efm_oracle_torch as OT  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BATCH, IMAGE = 64, 112


def inputs(rows=None):
    params = O.init_params(O.efm29_param_shapes(3, IMAGE), 42)
    w_head = O.uniform_pm((128, 342), 777, O.xavier_uniform_scale((128, 342)))
    x = O.uniform01(BATCH * 3 * IMAGE * IMAGE, 1234).reshape(BATCH, 3, IMAGE, IMAGE)
    neg = ((np.arange(BATCH // 2) + 5) % (BATCH // 2)).astype(np.int32)  # anchor i's negative = anchor (i+5)%32: another identity
    return params, w_head, (x if rows is None else x[rows]), neg


def forward(params, w_head, x):
    """(feat 342-d, emb 128-d) of a set of images; every row depends on its own image only."""
    tp = {k: torch.tensor(v) for k, v in params.items()}
    with torch.no_grad():
        feat = OT.efm29_forward(tp, torch.tensor(x))
        yn = feat / feat.norm(dim=1, keepdim=True)
        emb = yn @ torch.tensor(w_head).T
    return feat.numpy(), emb.numpy()


def loss_and_cosines(emb, neg, margin=0.2):
    h = emb.shape[0] // 2
    a, p = emb[:h], emb[h:]
    n = a[neg]
    loss = np.maximum(((p - a) ** 2).sum(1) - ((n - a) ** 2).sum(1) + margin, 0.0)
    cos = lambda u, v: (u * v).sum(1) / (np.linalg.norm(u, axis=1) * np.linalg.norm(v, axis=1))  # noqa: E731
    return loss, np.stack([cos(a, p), cos(a, n)], axis=1)


def main():
    params, w_head, x, neg = inputs()
    feat, emb = forward(params, w_head, x)
    loss, cosines = loss_and_cosines(emb, neg)
    np.savez_compressed(os.path.join(HERE, "config1_efm112.npz"), emb=emb.astype(np.float32), loss=loss, cosines=cosines,
                        feat_sum=feat.sum(1), feat_abs=np.abs(feat).sum(1), neg=neg)
    print("wrote config1_efm112.npz: loss mean %.6f, active %d/32" % (loss.mean(), int((loss > 0).sum())))


if __name__ == "__main__":
    main()
