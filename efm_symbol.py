"""Drop-in for the reference's efm_symbol.py (network-builder half): res_block / group / multi_gpu / get_net with the
reference's signatures, building `graph.Sym` networks that run on the MI355X HIP kernels.

    from efm_symbol import get_net
    logits, feature = get_net(classes=8398)
"""
from improving_face_recognition_performance_using_triplet_loss_amd.efm_symbol import (  # noqa: F401
    efm_feature, embedding_net, get_net, group, multi_gpu, res_block)
from improving_face_recognition_performance_using_triplet_loss_amd.functional import cosine_dist as _cosine_pairs  # noqa: F401
from improving_face_recognition_performance_using_triplet_loss_amd.data import Batch, DataIter, define_pos  # noqa: F401


def cosine_dist(a, b):
    """Cosine similarity of two flattened tensors (ref: efm_symbol.py:125-136)."""
    s, _ = _cosine_pairs(a.reshape(1, -1), b.reshape(1, -1), b.reshape(1, -1))
    return s[0]
