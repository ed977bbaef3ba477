#!/usr/bin/env python
"""Run Plan.autotune() for a BASELINE configuration on the GPU box and freeze the selection under
improving_face_recognition_performance_using_triplet_loss_amd/tuning/ (see that package's docstring).

    python tools/make_tuning.py [--workload efm] [--batch 256] [--image 112] [--rounds 3] [--out FILE]

--rounds N repeats the timing N times and keeps, per layer and direction, the majority choice (ties -> the first round's)."""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="efm", choices=["efm", "lightcnn9", "deepcnn"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--image", type=int, default=112)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, tuning
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    outs = {"efm": efm_symbol.embedding_net, "lightcnn9": efm_symbol.lightcnn9_embedding_net, "deepcnn": efm_symbol.deepcnn_embedding_net}[args.workload]()
    tables = []
    for r in range(args.rounds):
        plan = Plan(outs, (args.batch, 3, args.image, args.image), "cuda")
        plan.autotune()
        tables.append(plan.tuning_table())
        print("round", r, json.dumps(plan.chosen), flush=True)
    final = collections.OrderedDict()
    for name in tables[0]:
        entry = {}
        # forward and data gradient: (kernel family, tile) travel together
        for keys in (("tune_fwd", "wino_fwd"), ("tune_dgrad", "wino_dgrad"), ("tune_wgrad",)):
            votes = collections.Counter(tuple(t[name][k] for k in keys) for t in tables)
            best = max(votes.items(), key=lambda kv: (kv[1], -[tuple(t[name][k] for k in keys) for t in tables].index(kv[0])))[0]
            entry.update(dict(zip(keys, best)))
        final[name] = entry
    f = tuning.save(final, args.workload, args.batch, args.image, "f32", file=args.out,
                    note="Plan.autotune() on %s, majority of %d rounds" % (torch.cuda.get_device_name(0), args.rounds))
    print("wrote", f)


if __name__ == "__main__":
    main()
