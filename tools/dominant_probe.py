#!/usr/bin/env python
"""Launch the step's dominant kernel (conv2 forward, fused epilogue) a few times with the tiling / kernel the bench run chose —
the target of the rocprofv3 --pmc passes in tools/profile_round.sh (no autotune here, so no other launches of that kernel family).

    python tools/dominant_probe.py <winograd 0|1> <tune_fwd> [iters]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from improving_face_recognition_performance_using_triplet_loss_amd import ops  # noqa: E402


def main():
    wino, tune = int(sys.argv[1]), int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    d = ops.conv_desc(256, 56, 56, 66, 198, 3, 3, 1, 1)
    d.tune_fwd = tune
    x = torch.rand((d.batch, d.hin, d.win, d.cin_p), device="cuda")
    x[..., d.cin:] = 0
    w = (torch.rand((d.n_pad16, d.k_pad), device="cuda") - 0.5) * 0.1
    b = torch.zeros(d.n_pad16, device="cuda")
    if wino:
        u = ops.wino_mfm_make_u(d, w, 3)
        run = lambda: ops.wino_mfm_fwd(d, x, u, b, 3, 0, True)  # noqa: E731
    else:
        run = lambda: ops.conv_mfm_fwd(d, x, w, b, 3, 0, True)  # noqa: E731
    for _ in range(iters + 1):
        run()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
