#!/usr/bin/env python
"""Drop-in for the reference's extract_feacture_v2.py: run the trained EFM-29 over the train / test sets and dump the per-row
L2-normalised 342-d features and the labels as the CSV files `pre-trained_efm_v3.py` trains on.

    python extract_feacture_v2.py <root> <model_dir>                 # <root>/{train,test}.rec, <model_dir>/EFM_RES.params
    python extract_feacture_v2.py <root> <model_dir> --synthetic 256  # no dataset / checkpoint on disk

Same observable outputs as the reference (ref: extract_feacture_v2.py:56-110): `feature_vector_train.csv` /
`feature_vector_valid.csv` — one row per image, 342 floats each followed by a comma (:67-72) — and `label_train.csv` /
`label_valid.csv` — one float per line (:75-78); both are appended to, batch by batch, and a
"[batch N]: train acc A, in T sec" line is printed per batch (:80).  The forward runs on the HIP kernels (fused conv + MFM + pool
plan, `efm_l2norm_fwd` in row mode); the identity-head accuracy needs the checkpoint's fc2 and is reported as nan without one.
The checkpoint is an MXNet .params file (mxio.load_params: data only, nothing is executed); without it (--synthetic) the
weights are Xavier-initialised, which makes the dump a format / plumbing run.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, ops
from improving_face_recognition_performance_using_triplet_loss_amd.data import synthetic_source
from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan


def load_split(root, name, args, seed):
    rec = os.path.join(root, name + ".rec")
    if not args.synthetic and os.path.exists(rec):
        from improving_face_recognition_performance_using_triplet_loss_amd.mxio import ImageRecordIter
        return ImageRecordIter(path_imgrec=rec, shuffle=True, scale=1. / 255, rand_crop=True, rand_mirror=True,
                               data_shape=(args.channels, args.image_size, args.image_size), batch_size=args.batch_size, seed=seed,
                               device=torch.device("cuda", 0))
    if args.synthetic:
        return synthetic_source(args.synthetic, (args.channels, args.image_size, args.image_size), max(args.synthetic // 4, 2), seed,
                                args.batch_size)
    raise SystemExit("no %s found — pass --synthetic N" % rec)


def dump(plan, flat, src, tag, batch_size, fc2):
    """One pass over `src`: append normalised features / labels to the two CSV files, print the reference's per-batch line."""
    cnt = 0
    for batch in src:
        tic = time.time()
        data = batch.data[0].cuda().float().contiguous()
        label = batch.label[0]
        if data.shape[0] != batch_size:
            break  # ImageRecordIter pads / drops the tail like the reference's iterator; a short batch ends the pass
        feat = plan.forward(data, flat, train=False)[0]
        c = plan.outputs[0].shape[0]
        fmat = feat.view(batch_size, -1)[:, :c].contiguous()               # drop the pad channels of the NHWC buffer
        fc, _ = ops.l2norm_fwd(fmat)                                       # fc[v] / norm(fc[v])  (ref :69), row mode
        rows = fc.cpu().numpy()
        acc = float("nan")
        if fc2 is not None:
            logits = fmat @ fc2[0].T + fc2[1]
            acc = float((logits.argmax(dim=1).cpu() == label.to(torch.int64)).float().mean())
        with open("feature_vector_%s.csv" % tag, "a+", newline="") as f:
            for v in range(batch_size):
                f.write("".join("{},".format(float(e)) for e in rows[v]))
                f.write("\n")
        with open("label_%s.csv" % tag, "a+", newline="") as f:
            for v in range(batch_size):
                f.write("{}".format(float(label[v])))
                f.write("\n")
        print("[batch {}]: {} acc {:g}, in {:.1f} sec".format(cnt, "train" if tag == "train" else "valid", acc, time.time() - tic), flush=True)
        cnt += 1
    return cnt


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("model_dir", nargs="?", default=".")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic images per split (no dataset needed)")
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--channels", type=int, default=1)
    args = ap.parse_args(argv)

    train_src = load_split(args.root, "train", args, 1234)
    test_src = load_split(args.root, "test", args, 4321)
    print("Load model...", flush=True)
    js = os.path.join(args.model_dir, "EFM_RES.json")
    if os.path.exists(js):  # the checkpoint's own graph (ref :47-51 takes the internal `concat29_output` = the 342-d feature)
        from improving_face_recognition_performance_using_triplet_loss_amd import mxio
        try:
            feat = mxio.load_symbol(js, outputs=["concat29_output"])[0]
        except KeyError:  # a graph written by this package names the node differently: the 342-d feature is what feeds fc2
            feat = mxio.load_symbol(js)[-1]
            if feat.op == "fc" and feat.name == "fc2":
                feat = feat.inputs[0]
    else:
        data = efm_symbol.G.Variable("data")
        feat, _ = efm_symbol.efm_feature(data)
    plan = Plan([feat], (args.batch_size, args.channels, args.image_size, args.image_size))
    if args.batch_size >= 64 and os.environ.get("EFM_AUTOTUNE", "1") != "0":
        plan.autotune()                                                # per-layer kernel selection, timed once
    flat = plan.new_flat()
    fc2 = None
    ckpt = os.path.join(args.model_dir, "EFM_RES.params")
    if os.path.exists(ckpt):
        from improving_face_recognition_performance_using_triplet_loss_amd import mxio
        params = mxio.load_params(ckpt)
        plan.load_params(flat, {k: v for k, v in params.items() if k in plan.params})
        if "fc2_weight" in params:
            fc2 = (torch.as_tensor(params["fc2_weight"]).cuda(), torch.as_tensor(params["fc2_bias"]).cuda())
    else:
        print("no %s: Xavier-initialised weights (format / plumbing run)" % ckpt, flush=True)
        plan.init_xavier(flat, 42)
    print("Processing...", flush=True)
    dump(plan, flat, train_src, "train", args.batch_size, fc2)
    dump(plan, flat, test_src, "valid", args.batch_size, fc2)


if __name__ == "__main__":
    main()
