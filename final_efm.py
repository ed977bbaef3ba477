#!/usr/bin/env python
"""Drop-in for the reference's final_efm.py: triplet fine-tuning of a Dense(342, use_bias=False) head on top of the FROZEN
pre-trained EFM-29 (the images go through the backbone every step; LFW is the validation set).

    python final_efm.py <root> <model_dir>                   # <root>/{test,lfw_data}.{rec,lst}, <model_dir>/EFM_RES.{json,params}
    python final_efm.py <root> <model_dir> --synthetic 320    # no dataset / checkpoint on disk

Same constants, loop and outputs as the reference (ref: final_efm.py:131-318): 1x128x128 inputs, batch 40 anchors + their positives,
per-row L2 normalisation of the 342-d backbone features (:238-243), Dense(342, no bias) with Xavier init (:218-221), random negative
of another identity among the anchors (:250-254), TripletLoss(0.2), SGD lr 2.4e-4 wd 1e-5, "s_ap s_an" rows appended to
cosine_similarity.csv (:266-270), fc_efm_res-%04d.params per epoch, the "Epoch N: train loss ..." line (:317-318).
Deliberate deviations from the file as committed (it cannot run: SURVEY.md appendix): `loss = id_loss + alpha * TL_loss` names two
undefined variables (:261,301) — the frozen backbone gives the identity loss no trainable parameter, so the step trains on the
triplet loss; `train_acc, ... = 0., 0.` unpacks 4 names from 2 values (:231); the per-epoch file holds the trained head, not the
frozen net (:316); the matplotlib figures (:118-128, 321-322) are written as `curves.csv` instead.
The backbone forward is the fused HIP plan without activation storage (train=False); head / loss / cosine kernels as in
pre-trained_efm_v3.py.
"""
import argparse
import csv
import os
import time

import numpy as np
import torch

from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, mxio, ops
from improving_face_recognition_performance_using_triplet_loss_amd import functional as F_
from improving_face_recognition_performance_using_triplet_loss_amd.data import DataIter, define_pos, pick_negatives, synthetic_source
from improving_face_recognition_performance_using_triplet_loss_amd.nn import Dense, Trainer, TripletLoss
from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan


def cosine_dist(anc, pos, neg, batch_size):
    s_ap, s_an = F_.cosine_dist(anc[:batch_size], pos[:batch_size], neg[:batch_size])
    return s_ap.cpu().tolist(), s_an.cpu().tolist()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("model_dir", nargs="?", default=".")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--batch-size", type=int, default=40)
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    batch_size = args.batch_size
    shape = (args.channels, args.image_size, args.image_size)

    def split(name, n, seed):
        rec, lst = os.path.join(args.root, name + ".rec"), os.path.join(args.root, name + ".lst")
        if not args.synthetic and os.path.exists(rec):
            it = mxio.ImageRecordIter(path_imgrec=rec, shuffle=True, scale=1. / 255, rand_crop=True, rand_mirror=True, data_shape=shape,
                                      batch_size=batch_size, seed=seed, device=torch.device("cuda", 0))
            return it, (len(open(lst).readlines()) if os.path.exists(lst) else len(it))
        if not args.synthetic:
            raise SystemExit("no %s found — pass --synthetic N" % rec)
        return synthetic_source(n, shape, max(n // 4, 2), seed, batch_size), n
    train_dataiter, Training_IMG_number = split("test", args.synthetic, 1234)                       # sic: the reference trains on test.rec
    test_dataiter, Testing_IMG_number = split("lfw_data", max(args.synthetic // 2, 2 * batch_size), 4321)
    print("Totoal number of training samples = ", Training_IMG_number, flush=True)
    print("Totoal number of testing samples = ", Testing_IMG_number, flush=True)
    epoch_size = Training_IMG_number / batch_size
    dshape = (batch_size,) + shape

    print("defining positive image...", flush=True)
    pos_img_train = define_pos(train_dataiter, int(epoch_size), batch_size)
    pos_img_test = define_pos(test_dataiter, int(Testing_IMG_number / batch_size), batch_size)
    train_dataiter.reset()
    test_dataiter.reset()
    print("making training pairs...", flush=True)
    data_train = DataIter(train_dataiter, int(epoch_size), pos_img_train, batch_size, dshape)
    print("making testing pairs...", flush=True)
    data_test = DataIter(test_dataiter, int(Testing_IMG_number / batch_size), pos_img_test, batch_size, dshape)

    lr, MARGIN = 0.00024, 0.2
    devs = torch.device("cuda", 0)
    print("load efm model...", flush=True)
    js, ck = os.path.join(args.model_dir, "EFM_RES.json"), os.path.join(args.model_dir, "EFM_RES.params")
    if os.path.exists(js):
        try:
            feat_sym = mxio.load_symbol(js, outputs=["concat29_output"])[0]
        except KeyError:
            feat_sym = mxio.load_symbol(js)[-1]
            if feat_sym.op == "fc" and feat_sym.name == "fc2":
                feat_sym = feat_sym.inputs[0]
    else:
        feat_sym, _ = efm_symbol.efm_feature(efm_symbol.G.Variable("data"))
    plan = Plan([feat_sym], (2 * batch_size,) + shape, devs)          # a batch = B anchors followed by their B positives
    if 2 * batch_size >= 64 and os.environ.get("EFM_AUTOTUNE", "1") != "0":
        plan.autotune()                                                # per-layer kernel selection, timed once
    flat = plan.new_flat()
    if os.path.exists(ck):
        params = mxio.load_params(ck)
        plan.load_params(flat, {k: v for k, v in params.items() if k in plan.params})
    else:
        print("no %s: Xavier-initialised backbone (plumbing run)" % ck, flush=True)
        plan.init_xavier(flat, 42)
    c = plan.outputs[0].shape[0]

    print("build network...", flush=True)
    model = Dense(342, use_bias=False, in_units=c)
    model._materialise(c, devs)
    triplet_loss = TripletLoss(margin=MARGIN)
    trainer = Trainer(model.parameters(), "sgd", learning_rate=lr, wd=0.00001)
    rng = np.random.default_rng(args.seed)

    def run(batch):
        data = batch.data[0].to(devs).float().contiguous()
        label = batch.label[0]
        fc = plan.forward(data, flat, train=False)[0].view(2 * batch_size, -1)[:, :c].contiguous()   # frozen backbone
        n_data, _ = ops.l2norm_fwd(fc)                                                                 # fc[i] / norm(fc[i])
        Wnx = model(n_data)
        anc, pos = Wnx[0:batch_size], Wnx[batch_size:batch_size * 2]
        neg = F_.gather_negatives(Wnx, pick_negatives(label, batch_size, batch_size, rng).to(devs))   # detached copy of Wnx[j]
        return triplet_loss(anc, pos, neg), (anc, pos, neg)

    curves = []
    print("start training...", flush=True)
    for epoch in range(args.epochs):
        train_loss, valid_loss = 0., 0.
        tic = time.time()
        for batch in data_train:
            loss, (anc, pos, neg) = run(batch)
            loss.sum().backward()
            trainer.step(batch_size)
            train_loss += loss.mean().item()
            pos_dist, neg_dist = cosine_dist(anc, pos, neg, batch_size)
            with open("cosine_similarity.csv", "a+", newline="") as csvfile:
                csvwriter = csv.writer(csvfile, delimiter=" ")
                for v in range(batch_size):
                    csvwriter.writerow([pos_dist[v], neg_dist[v]])
        with torch.no_grad():
            for batch in data_test:
                loss, _ = run(batch)
                valid_loss += loss.mean().item()
        data_train.reset()
        data_test.reset()
        tl, vl = train_loss / epoch_size, valid_loss / (Testing_IMG_number / batch_size)
        curves.append((epoch, tl, vl))
        mxio.save_params("fc_efm_res-%04d.params" % epoch, {"dense0_weight": model.weight_mx().cpu().numpy()})
        print("Epoch {}: train loss {:g}, valid loss {:g}, in {:.1f} sec".format(epoch, tl, vl, time.time() - tic), flush=True)
    with open("curves.csv", "w", newline="") as f:
        csv.writer(f).writerows([("epoch", "train_loss", "valid_loss")] + curves)


if __name__ == "__main__":
    main()
