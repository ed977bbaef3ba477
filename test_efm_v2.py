#!/usr/bin/env python
"""Drop-in for the reference's test_efm_v2.py: the cosine-similarity test of pre-extracted 342-d EFM features.

    python test_efm_v2.py                      # reads train_img.csv / train_id.csv from CWD (extract_feacture_v2.py writes them)
    python test_efm_v2.py --synthetic 65536    # splitmix64 features instead

Same flow and outputs as the reference (ref: test_efm_v2.py:104-180): count the samples, keep one positive per identity
(`define_pos`), batches of 16384 anchors followed by their positives, per-row L2 normalisation of all 2B rows (:150-153), one
random negative of another identity per anchor drawn from the 2B rows (:160-165), `cosine_dist`, rows "s_ap s_an" appended to
cosine_similarity.csv (:172-176), one "[batch N]: in T sec" line per batch (:179).  The 2B per-row norm launches, the B-iteration
negative loop with a device sync per sample and the 2B cosine launches of the reference are one `efm_l2norm_fwd`, one vectorised
draw + `efm_gather_rows`, and one `efm_cosine_pairs` launch.  (Not a pytest module despite its name: the test suite lives in tests/.)
"""
import argparse
import csv
import time

import numpy as np

from improving_face_recognition_performance_using_triplet_loss_amd import functional as F_
from improving_face_recognition_performance_using_triplet_loss_amd import ops
from improving_face_recognition_performance_using_triplet_loss_amd.data import CSVIter, DataIter, define_pos, pick_negatives, synthetic_source


def cosine_dist(anc, pos, neg, batch_size):
    s_ap, s_an = F_.cosine_dist(anc[:batch_size], pos[:batch_size], neg[:batch_size])
    return s_ap.cpu().tolist(), s_an.cpu().tolist()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--batch-size", type=int, default=4096 * 4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    feature_dim, batch_size = 342, args.batch_size
    if args.synthetic:
        IMG_number = args.synthetic
        test_dataiter = synthetic_source(IMG_number, (feature_dim,), max(IMG_number // 8, 2), 1234, batch_size)
    else:
        with open("train_id.csv", "r") as file:
            IMG_number = len(file.readlines())
        test_dataiter = CSVIter("train_img.csv", "train_id.csv", batch_size, feature_dim)
    print("Totoal number of samples = ", IMG_number, flush=True)
    epoch_size = IMG_number / batch_size
    dshape = (batch_size, feature_dim)
    print("epoch_size: {}".format(epoch_size), flush=True)
    print("defining positive image...", flush=True)
    pos_img = define_pos(test_dataiter, int(epoch_size), batch_size)
    test_dataiter.reset()
    print("making pairs...", flush=True)
    data_test = DataIter(test_dataiter, int(epoch_size), pos_img, batch_size, dshape)
    rng = np.random.default_rng(args.seed)

    print("start testing...", flush=True)
    cnt = 0
    for batch in data_test:
        tic = time.time()
        data = batch.data[0].cuda().float().contiguous()
        label = batch.label[0]
        n_data, _ = ops.l2norm_fwd(data)                                       # data[i] / norm(data[i]) for all 2B rows
        anc, pos = n_data[0:batch_size], n_data[batch_size:batch_size * 2]
        neg = ops.gather_rows(n_data, pick_negatives(label, batch_size, batch_size * 2, rng).cuda())
        pos_dist, neg_dist = cosine_dist(anc, pos, neg, batch_size)
        with open("cosine_similarity.csv", "a+", newline="") as csvfile:
            csvwriter = csv.writer(csvfile, delimiter=" ")
            for v in range(batch_size):
                csvwriter.writerow([pos_dist[v], neg_dist[v]])
        cnt += 1
        print("[batch {}]: in {:.1f} sec".format(cnt, time.time() - tic), flush=True)


if __name__ == "__main__":
    main()
