"""Pair iterator of the reference scripts (define_pos / Batch / DataIter, ref: train_efm.py:37-114 and the streaming
variant pre-trained_efm_v3.py:59-111) over simple host-side sources: synthetic faces/features and CSV feature files
(the format extract_feacture_v2.py writes and mx.io.CSVIter reads, ref: pre-trained_efm_v3.py:155-156).

A batch is [B anchors ; B positives] with the labels duplicated; the positive of identity k is the FIRST sample of k
seen in the source (it may be the anchor itself, exactly as in the reference).
"""
import numpy as np
import torch

from . import synth


class Batch(object):
    def __init__(self, data_names, data, label_names, label):
        self.data, self.label = data, label
        self.data_names, self.label_names = data_names, label_names

    @property
    def provide_data(self):
        return [(n, tuple(x.shape)) for n, x in zip(self.data_names, self.data)]

    @property
    def provide_label(self):
        return [(n, tuple(x.shape)) for n, x in zip(self.label_names, self.label)]


class ArrayIter:
    """Minimal mx.io.NDArrayIter / CSVIter stand-in: fixed-size batches over in-memory arrays, last partial batch dropped."""

    def __init__(self, data, label, batch_size):
        self.data_arr, self.label_arr, self.batch_size = data, label, batch_size
        self.pos = 0

    def __iter__(self):
        self.reset()
        return self

    def __next__(self):
        if self.pos + self.batch_size > len(self.data_arr):
            raise StopIteration
        s = slice(self.pos, self.pos + self.batch_size)
        self.pos += self.batch_size
        return Batch(["data"], [self.data_arr[s]], ["label"], [self.label_arr[s]])

    next = __next__

    def reset(self):
        self.pos = 0


def CSVIter(data_csv, label_csv, batch_size, feature_dim):
    """Rows of `feature_dim` comma-separated floats (a trailing comma is tolerated, as extract_feacture_v2.py:67-79
    writes one); labels one float per line."""
    rows = []
    with open(data_csv) as f:
        for line in f:
            vals = [v for v in line.strip().split(",") if v != ""]
            if vals:
                rows.append(np.asarray(vals[:feature_dim], dtype=np.float32))
    data = torch.from_numpy(np.stack(rows))
    label = torch.from_numpy(np.loadtxt(label_csv, dtype=np.float32).reshape(-1))
    return ArrayIter(data, label, batch_size)


def synthetic_source(n, shape, identities, seed, batch_size):
    """n samples of `shape` U[0,1) (splitmix64 stream) with labels cycling over `identities` ids."""
    count = int(np.prod(shape))
    data = synth.uniform01(n * count, seed, device="cpu").view((n,) + tuple(shape))
    label = (torch.arange(n) % identities).to(torch.float32)
    return ArrayIter(data, label, batch_size)


def define_pos(data_iter, length, batch_size):
    """{identity: first sample seen} over one pass of the source (ref: train_efm.py:37-45)."""
    pos_img = {}
    for batch in data_iter:
        lab = batch.label[0]
        for i in range(batch_size):
            k = int(lab[i])
            if k not in pos_img:
                pos_img[k] = batch.data[0][i].clone()
    return pos_img


class DataIter:
    """Streams [anchors ; positives] batches (ref: pre-trained_efm_v3.py:84-107; train_efm.py materialises all pairs
    in host memory first, :74-85 — same batches, O(dataset) less memory)."""

    def __init__(self, data_iter, length, pos_img, batch_size, dshape=None):
        self.data_iter, self.length, self.pos_img, self.batch_size = data_iter, length, pos_img, batch_size
        self.provide_data = [("data", dshape)]
        self.provide_label = [("label", (batch_size, 1))]

    def __iter__(self):
        self.data_iter.reset()
        for _ in range(self.length):
            try:
                b = self.data_iter.next()
            except StopIteration:
                return
            data, lab = b.data[0], b.label[0]
            pos = torch.stack([self.pos_img[int(lab[i])] for i in range(self.batch_size)])
            yield Batch(["data"], [torch.cat([data, pos])], ["label"], [torch.cat([lab, lab])])

    def reset(self):
        self.data_iter.reset()


def pick_negatives(labels, batch_size, pool, rng):
    """The reference's rejection sampling, vectorised per draw round (ref: train_efm.py:234-239; validation draws from
    all 2B rows, :268-273).  Raises instead of spinning forever when the pool holds one identity."""
    lab = labels.cpu()
    a = lab[:batch_size]
    if bool((lab[:pool] == a[0]).all()) and bool((a == a[0]).all()):
        raise ValueError("a batch with a single identity has no negative (the reference would loop forever)")
    idx = torch.from_numpy(rng.integers(0, pool, size=batch_size))
    bad = lab[idx] == a
    while bool(bad.any()):
        idx[bad] = torch.from_numpy(rng.integers(0, pool, size=int(bad.sum())))
        bad = lab[idx] == a
    return idx.to(torch.int32)
