"""LFW verification protocol on device embeddings.

Restates the 10-fold best-threshold accuracy of the reference's vendored facenet.py (`calculate_roc` :428-459,
`calculate_accuracy` :461-471, `distance` :412-426): the per-pair distances come from one HIP kernel launch per fold
(`efm_pair_distance`), the fold / threshold bookkeeping is host code as in the reference.  Pinned by golden vectors
produced by running the reference's own functions (tests/golden/lfw_roc.npz).
"""
import math

import numpy as np
import torch

from . import ops


def _accuracy(threshold, dist, issame):
    pred = dist < threshold
    tp = np.sum(pred & issame)
    fp = np.sum(pred & ~issame)
    tn = np.sum(~pred & ~issame)
    fn = np.sum(~pred & issame)
    tpr = 0 if tp + fn == 0 else float(tp) / float(tp + fn)
    fpr = 0 if fp + tn == 0 else float(fp) / float(fp + tn)
    return tpr, fpr, float(tp + tn) / dist.size


def distance(emb1, emb2, mean=None, distance_metric=0):
    sq, cs = ops.pair_distance(emb1, emb2, mean)
    if distance_metric == 0:
        return sq.cpu().numpy().astype(np.float64)
    if distance_metric == 1:
        return np.arccos(np.clip(cs.cpu().numpy().astype(np.float64), -1.0, 1.0)) / math.pi
    raise ValueError("Undefined distance metric %d" % distance_metric)


def calculate_roc(thresholds, embeddings1, embeddings2, actual_issame, nrof_folds=10, distance_metric=0, subtract_mean=False):
    """embeddings: (N, D) fp32 device tensors; returns (tpr, fpr, accuracy[nrof_folds]) like the reference."""
    assert embeddings1.shape == embeddings2.shape
    issame = np.asarray(actual_issame, dtype=bool)
    n = min(len(issame), embeddings1.shape[0])
    folds = np.array_split(np.arange(n), nrof_folds)  # KFold(shuffle=False): contiguous folds, first n % k one longer
    tprs = np.zeros((nrof_folds, len(thresholds)))
    fprs = np.zeros((nrof_folds, len(thresholds)))
    accuracy = np.zeros(nrof_folds)
    e1, e2 = embeddings1.contiguous(), embeddings2.contiguous()
    dist_all = None if subtract_mean else distance(e1, e2, None, distance_metric)
    for f, test in enumerate(folds):
        train = np.concatenate([folds[k] for k in range(nrof_folds) if k != f])
        if subtract_mean:
            idx = torch.as_tensor(train, device=e1.device)
            mean = torch.cat([e1[idx], e2[idx]]).mean(dim=0).contiguous()
            dist = distance(e1, e2, mean, distance_metric)
        else:
            dist = dist_all
        acc_train = np.array([_accuracy(t, dist[train], issame[train])[2] for t in thresholds])
        best = int(np.argmax(acc_train))
        for ti, t in enumerate(thresholds):
            tprs[f, ti], fprs[f, ti], _ = _accuracy(t, dist[test], issame[test])
        accuracy[f] = _accuracy(thresholds[best], dist[test], issame[test])[2]
    return tprs.mean(0), fprs.mean(0), accuracy


def evaluate(embeddings1, embeddings2, actual_issame, nrof_folds=10, distance_metric=0, subtract_mean=False):
    """Mean / std of the 10-fold accuracy — the number BASELINE.json calls 'LFW acc'."""
    thresholds = np.arange(0, 4, 0.01) if distance_metric == 0 else np.arange(0, 1, 0.0025)
    tpr, fpr, acc = calculate_roc(thresholds, embeddings1, embeddings2, actual_issame, nrof_folds, distance_metric, subtract_mean)
    return float(acc.mean()), float(acc.std()), tpr, fpr
