#!/usr/bin/env python
"""Freezes the 'mini-EFM' end-to-end fixture from the NumPy fp64 oracle: B=4 (2 anchors + 2 positives),
3x32x32, all 29 convolutions at real width, weights / inputs from the portable splitmix64 generator (nothing but the outputs
is stored).  The reference has no fixture for this path (parity unpinned); this one pins the ORACLE against drift and gives
the HIP path a second, file-based target."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import efm_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def inputs():
    params = O.init_params(O.efm29_param_shapes(3, 32), 42)
    w_head = O.uniform_pm((128, 342), 777, O.xavier_uniform_scale((128, 342)))
    x = O.uniform01(4 * 3 * 32 * 32, 1234).reshape(4, 3, 32, 32)
    neg = np.array([1, 0], dtype=np.int32)
    demb = (O.uniform01(4 * 128, 99) * 2 - 1).reshape(4, 128)
    return params, w_head, x, neg, demb


def main():
    params, w_head, x, neg, demb = inputs()
    loss, emb, feat, grads, g_head = O.train_step_loss(params, w_head, x, neg, 0.2, demb=demb)
    out = {"loss": loss, "emb": emb, "feat": feat, "g_head_sum": g_head.sum(), "g_head_abs": np.abs(g_head).sum()}
    for k in ("conv1_weight", "conv3_res_weight", "fc1_weight"):
        out["grad_" + k] = grads[k] if grads[k].size < 20000 else grads[k].reshape(-1)[:: max(grads[k].size // 4096, 1)]
    out["grad_abs_sums"] = np.array([np.abs(grads[k]).sum() for k in sorted(grads)])
    np.savez_compressed(os.path.join(HERE, "mini_efm.npz"), **out)
    print("wrote mini_efm.npz", {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
