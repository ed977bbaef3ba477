"""TEST / BASELINE INFRASTRUCTURE — never imported by the product path.

CPU restatement of the two per-sample HOST loops the reference runs every training step, kept literal (one Python iteration,
one scalar read-back and one row copy per anchor) so that bench.py's `cpu_baseline` leg can time what the reference's step pays
for them beside the vectorised device path (`data.pick_negatives` + `efm_gather_rows`, `efm_cosine_pairs`):

  * negative pick, ref: train_efm.py:234-239 and pre-trained_efm_v3.py:202-207 — for each anchor i draw j ~ U{0..B-1} with
    `random.randint` until `int(label[j].asscalar()) != int(label[i].asscalar())`, then `neg.append(fc[j].asnumpy())`, and finally
    `mx.nd.array(neg)`: 2 blocking scalar read-backs per draw, one row copy per anchor, one (B, D) re-upload;
  * `cosine_dist`, ref: train_efm.py:26-34 — per anchor two `dot / (norm * norm)` on single rows, and (train_efm.py:251-255) one
    `.asscalar()` per value when the CSV rows are written.

MXNet is absent from this image, so the NDArray calls are restated on torch CPU tensors: `.item()` for
`.asscalar()`, `.numpy()` for `.asnumpy()`, `torch.dot` / `torch.linalg.vector_norm` for `mx.nd.dot` / `mx.nd.norm`.  Parity
unpinned like the rest of the oracle; the RESULTS are checked against the vectorised forms in tests/test_oracle.py.
"""
import random

import numpy as np
import torch


def pick_negatives_loop(label, fc, batch_size, rng=None):
    """ref: train_efm.py:234-239.  label: (>= batch_size,) tensor, fc: (>= batch_size, D) tensor.  -> (neg (B, D) tensor, idx list)."""
    rng = rng or random
    neg, idx = [], []
    for i in range(batch_size):
        j = rng.randint(0, batch_size - 1)
        while int(label[j].item()) == int(label[i].item()):
            j = rng.randint(0, batch_size - 1)
        neg.append(fc[j].numpy())
        idx.append(j)
    return torch.from_numpy(np.array(neg)), idx


def cosine_dist_loop(anc, pos, neg, batch_size):
    """ref: train_efm.py:26-34.  -> (pos_dist, neg_dist): lists of 0-d tensors."""
    pos_dist, neg_dist = [], []
    for i in range(batch_size):
        pos_dist.append(torch.dot(anc[i], pos[i]) / (torch.linalg.vector_norm(anc[i]) * torch.linalg.vector_norm(pos[i])))
        neg_dist.append(torch.dot(anc[i], neg[i]) / (torch.linalg.vector_norm(anc[i]) * torch.linalg.vector_norm(neg[i])))
    return pos_dist, neg_dist


def csv_rows(pos_dist, neg_dist, batch_size):
    """ref: train_efm.py:254-255 — the per-value `.asscalar()` read-backs of the CSV writer (no file is written here)."""
    return [(pos_dist[v].item(), neg_dist[v].item()) for v in range(batch_size)]
