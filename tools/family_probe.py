#!/usr/bin/env python
"""Run only the launches of ONE kernel instance of the benchmark step (e.g. "wino4_k<3>"), with the committed tuning table —
the program the rocprofv3 --pmc passes of tools/profile_round2.sh wrap.   usage: family_probe.py "<kernel name prefix>" [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from improving_face_recognition_performance_using_triplet_loss_amd import tuning  # noqa: E402
from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer  # noqa: E402

name, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
table, _ = tuning.load("efm", 256, 112, "f32")
tr = TripletTrainer(256, image=112, tuning=table, autotune=table is None)
fam = bench.kernel_families(tr, torch, iters=iters, only=name, dedup=False)
torch.cuda.synchronize()
for k, f in fam.items():
    print(k, f["launches"], "launches/step", round(f["ms"], 3), "ms/step")
