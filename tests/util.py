"""Helpers shared by the parity tests."""
import numpy as np
import torch

from improving_face_recognition_performance_using_triplet_loss_amd import ops


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def to_nhwc(x_nchw):
    """numpy NCHW -> device NHWC(pad4) through the library's own kernel."""
    return ops.nchw_to_nhwc(dev(x_nchw))


def from_nhwc(x, c):
    return ops.nhwc_to_nchw(x, c).cpu().numpy().astype(np.float64)


def rand(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return rng.uniform(-scale, scale, size=shape)
