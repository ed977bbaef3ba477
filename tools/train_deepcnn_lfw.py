#!/usr/bin/env python
"""BASELINE configs[4]: the deeper 512-d CNN (or LightCNN-9) trained with in-batch semi-hard triplets in bf16 (or fp32), data
parallel over the GPUs of one node, with an LFW-protocol pair evaluation every N steps.

    python tools/train_deepcnn_lfw.py --steps 200 --eval-every 50                      # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_deepcnn_lfw.py --steps 200

No dataset exists in this environment: training batches are synthetic identities (synth.identity_faces: P = batch/4
identities x 4 images per step, identity ids sharded by rank like Celeb1M shards) and the evaluation set is `--pairs`
matched + `--pairs` mismatched pairs of HELD-OUT identities, scored by the LFW 10-fold best-threshold protocol
(lfw.evaluate, restating the reference's vendored facenet.calculate_roc; golden-tested).  Every rank evaluates the same
pairs (weights are identical), rank 0 prints.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", choices=["deepcnn", "lightcnn9"], default="deepcnn")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (BASELINE configs[4]: 128)")
    ap.add_argument("--image", type=int, default=112)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--identities", type=int, default=1024, help="training identities per rank")
    ap.add_argument("--pairs", type=int, default=300, help="matched (and mismatched) evaluation pairs")
    ap.add_argument("--optimizer", choices=["adam", "sgd"], default="adam")
    ap.add_argument("--lr", type=float, default=2.4e-4)
    ap.add_argument("--margin", type=float, default=0.2)
    ap.add_argument("--noise", type=float, default=0.25)
    ap.add_argument("--negatives", choices=["semihard", "random"], default="semihard",
                    help="semihard: mined on device from the batch cosine matrix; random: the reference's rule (train_efm.py:234-239), "
                         "any image of another identity")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("EFM_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("EFM_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, device_id=device) if backend == "nccl" else dist.init_process_group(backend)

    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, lfw, synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import MiningTripletTrainer
    net = efm_symbol.deepcnn_embedding_net() if args.net == "deepcnn" else efm_symbol.lightcnn9_embedding_net()
    tr = MiningTripletTrainer(args.batch, image=args.image, optimizer=args.optimizer, lr=args.lr, wd=1e-5, margin=args.margin,
                              device=device, seed=42, outputs=net, dtype=args.dtype)
    per = args.batch // 4
    tr.set_labels(np.repeat(np.arange(per), 4))  # P identities x 4 images; the actual ids change every step, the layout does not

    # evaluation pairs from held-out identities (ids above every rank's training range)
    first_eval = world * args.identities
    n_eval = 2 * args.pairs
    ids1 = np.concatenate([first_eval + np.arange(args.pairs), first_eval + np.arange(args.pairs)])
    ids2 = np.concatenate([first_eval + np.arange(args.pairs), first_eval + args.pairs + np.arange(args.pairs)])
    issame = np.concatenate([np.ones(args.pairs, bool), np.zeros(args.pairs, bool)])
    order = np.random.default_rng(0).permutation(n_eval)  # interleave matched / mismatched pairs over the 10 folds
    ids1, ids2, issame = ids1[order], ids2[order], issame[order]

    def embed(ids, seed):
        out = []
        for s in range(0, len(ids), args.batch):
            chunk = ids[s:s + args.batch]
            pad = np.concatenate([chunk, np.full(args.batch - len(chunk), chunk[-1])])
            x = synth.identity_faces(pad, 3, args.image, seed + s, args.noise, device=device)
            emb, _ = tr.plan.forward(x, tr.flat, train=False)
            out.append(emb[:len(chunk), :emb.shape[1]].clone())
        return torch.cat(out).contiguous()

    def evaluate(step):
        e1, e2 = embed(ids1, 900001), embed(ids2, 900002)
        acc, std, _, _ = lfw.evaluate(e1, e2, issame, nrof_folds=10, distance_metric=0)
        if rank == 0:
            print("step %d  LFW-protocol accuracy %.4f +- %.4f  (%d held-out pairs)" % (step, acc, std, n_eval), flush=True)
        return acc

    rng = np.random.default_rng(1000 + rank)
    evaluate(0)
    t0 = time.perf_counter()
    for step in range(1, args.steps + 1):
        ids = rank * args.identities + rng.choice(args.identities, size=per, replace=False)
        x = synth.identity_faces(np.repeat(ids, 4), 3, args.image, 7 * step + rank, args.noise, device=device)
        neg = None
        if args.negatives == "random":  # layout is P identities x 4 images: shift by a random non-zero number of identities
            shift = 4 * rng.integers(1, per, size=args.batch) + rng.integers(0, 4, size=args.batch) - (np.arange(args.batch) % 4)
            neg = torch.as_tensor(((np.arange(args.batch) + shift) % args.batch).astype(np.int32)).to(device)
        loss = tr.step(x, neg)
        if step % 10 == 0 and rank == 0:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("step %d  loss %.5f  %.1f triplets/s (incl. data generation and evaluation)"
                  % (step, float(loss.mean()), world * args.batch * step / dt), flush=True)
        if step % args.eval_every == 0:
            evaluate(step)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
