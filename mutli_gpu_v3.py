#!/usr/bin/env python
"""Drop-in for the reference's mutli_gpu_v3.py: data-parallel softmax (identity) pre-training of the Symbol EFM-29, the run that
produces the `try2_efm_light_29-symbol.json` / `-%04d.params` checkpoints the other scripts start from.

    python mutli_gpu_v3.py --synthetic 800 --epochs 2                          # one GPU, no dataset on disk
    python mutli_gpu_v3.py --train-rec trainImg.rec --test-rec testImg.rec       # RecordIO as the reference (:138-141)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 mutli_gpu_v3.py ...   # [gpu0, gpu1] of the reference

Same constants and observable behaviour as the reference (ref: mutli_gpu_v3.py:103-162): 3x128x128 inputs, global batch 100 split
evenly over the devices (`mx.mod.Module(context=devs)`, :153), Xavier(factor_type="in", magnitude=2.34) (:156), Adam lr 2.4e-4
beta1 0.9 wd 1e-5 with rescale_grad = 1/batch and FactorScheduler(6 epochs, 0.88, stop 5e-15) (:159), gradient SUM across devices
(`kvstore local`, :158), accuracy metric, a Speedometer line every 100 batches and a checkpoint per epoch (:160-162).
`mx.mod.Module` / kvstore / memonger are replaced by one process per GPU + an all-reduce of the flat gradient (RCCL); the backbone
runs on the HIP kernels, Dropout(0.7) -> FullyConnected(classes) -> softmax cross-entropy of the id head are torch ops (not on the
north-star kernel list).  The Python-2 `xrange` of the reference is not reproduced.
"""
import argparse
import datetime
import logging
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, mxio
from improving_face_recognition_performance_using_triplet_loss_amd.data import synthetic_source
from improving_face_recognition_performance_using_triplet_loss_amd.nn import FactorScheduler, SymbolNet, Trainer


def ensure_dir(f):
    d = os.path.dirname(f)
    if d and not os.path.exists(d):
        os.makedirs(d)


def mutli_gpu(classes):
    """(softmax head input = fc2 logits symbol, the 342-d feature symbol): the reference's builder name, see efm_symbol.multi_gpu."""
    data = efm_symbol.G.Variable("data")
    fc2, fc1 = efm_symbol.multi_gpu(data, classes)
    return fc2, fc1.feature


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-rec", default="")
    ap.add_argument("--test-rec", default="")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic training images (test = a quarter of it)")
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=280)
    ap.add_argument("--batch-size", type=int, default=100, help="GLOBAL batch, split evenly over the ranks")
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--out-dir", default="try2_efm_light_29_134")
    args = ap.parse_args(argv)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("EFM_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    devs = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("EFM_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, device_id=devs) if backend == "nccl" else dist.init_process_group(backend)
    if args.batch_size % world:
        raise SystemExit("the global batch %d does not split over %d devices" % (args.batch_size, world))
    batch_size, local_batch = args.batch_size, args.batch_size // world
    Training_IMG_channel, Training_IMG_size = 3, args.image_size

    def source(path, n, seed):
        if path and os.path.exists(path):
            it = mxio.ImageRecordIter(path_imgrec=path, shuffle=True, scale=1. / 255, rand_crop=True, rand_mirror=True,
                                      data_shape=(Training_IMG_channel, Training_IMG_size, Training_IMG_size), batch_size=local_batch,
                                      seed=seed + rank, part_index=rank, num_parts=world, device=devs)   # each rank reads ITS share of the records
            return it, it.num_total
        if not n:
            raise SystemExit("no RecordIO file given — pass --train-rec/--test-rec or --synthetic N")
        return synthetic_source(n // world, (Training_IMG_channel, Training_IMG_size, Training_IMG_size), args.classes, seed + rank, local_batch), n
    train_dataiter, Training_IMG_number = source(args.train_rec, args.synthetic, 1234)
    test_dataiter, _ = source(args.test_rec, max(args.synthetic // 4, batch_size), 4321)
    epoch_size = Training_IMG_number / batch_size
    lr = 0.00024

    Log_save_dir = os.path.join(args.out_dir, "log") + "/"
    Model_save_dir = os.path.join(args.out_dir, "model") + "/"
    Model_save_name = "try2_efm_light_29"
    if rank == 0:
        ensure_dir(Log_save_dir)
        ensure_dir(Model_save_dir)
        logging.basicConfig(filename=Log_save_dir + Model_save_name + datetime.datetime.now().strftime("%Y-%m-%d_%H%M%S") + ".log", level=logging.INFO)
        root_logger = logging.getLogger()
        root_logger.addHandler(logging.StreamHandler(sys.stdout))
        root_logger.setLevel(logging.INFO)

    fc2_sym, feat_sym = mutli_gpu(args.classes)
    net = SymbolNet([feat_sym], Training_IMG_channel, Training_IMG_size, device=devs, seed=42, init=None,
                    autotune=os.environ.get("EFM_AUTOTUNE", "1") != "0")   # per-layer kernel selection, timed once (batches >= 64)
    net.plan(2).init_xavier(net.flat.data, 42, magnitude=2.34, factor_type="in")          # mx.init.Xavier(factor_type="in", magnitude=2.34)
    head = torch.nn.Sequential(torch.nn.Dropout(0.7), torch.nn.Linear(342, args.classes)).to(devs)
    with torch.no_grad():
        bound = float(np.sqrt(2.34 / 342))
        head[1].weight.uniform_(-bound, bound)
        head[1].bias.zero_()
    params = list(net.parameters()) + list(head.parameters())
    sched = FactorScheduler(step=int(epoch_size * 6), factor=0.88, stop_factor_lr=5e-15)
    trainer = Trainer(params, "adam", learning_rate=lr, wd=0.00001, lr_scheduler=sched, beta1=0.9)
    ce = torch.nn.CrossEntropyLoss(reduction="sum")

    def run(it, train):
        correct, seen, nbatch, tic, speed_tic = 0, 0, 0, time.time(), time.time()
        net.train(train)
        head.train(train)
        for batch in it:
            data, label = batch.data[0].to(devs).float(), batch.label[0].to(devs).long()
            if data.shape[0] != local_batch:
                break
            with torch.set_grad_enabled(train):
                (feat,) = net(data)
                out = head(feat)
                if train:
                    ce(out, label).backward()
                    if world > 1:  # kvstore 'local': sum of the device gradients; rescale_grad = 1/global batch is the optimiser's
                        for p in params:
                            dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    trainer.step(batch_size)
            correct += int((out.argmax(dim=1) == label).sum())
            seen += local_batch
            nbatch += 1
            if train and nbatch % 100 == 0 and rank == 0:  # mx.callback.Speedometer(batch_size, 100)
                logging.info("Epoch[%d] Batch [%d]\tSpeed: %.2f samples/sec\taccuracy=%f", run.epoch, nbatch,
                             100 * batch_size / (time.time() - speed_tic), correct / max(seen, 1))
                speed_tic = time.time()
        it.reset()
        stat = torch.tensor([correct, seen], dtype=torch.float64, device=devs)
        if world > 1:
            dist.all_reduce(stat)
        return float(stat[0] / max(float(stat[1]), 1.0)), time.time() - tic

    for epoch in range(args.epochs):
        run.epoch = epoch
        acc, cost = run(train_dataiter, True)
        if rank == 0:
            logging.info("Epoch[%d] Train-accuracy=%f", epoch, acc)
            logging.info("Epoch[%d] Time cost=%.3f", epoch, cost)
            # mx.callback.do_checkpoint(prefix): prefix-symbol.json + prefix-%04d.params ("arg:" keys), epoch numbers start at 1
            prefix = Model_save_dir + Model_save_name
            mxio.save_symbol(prefix + "-symbol.json", [fc2_sym])
            p = {"arg:" + k: v.cpu().numpy() for k, v in net.export_params().items()}
            p["arg:fc2_weight"], p["arg:fc2_bias"] = head[1].weight.detach().cpu().numpy(), head[1].bias.detach().cpu().numpy()
            mxio.save_params("%s-%04d.params" % (prefix, epoch + 1), p)
            logging.info('Saved checkpoint to "%s-%04d.params"', prefix, epoch + 1)
        vacc, _ = run(test_dataiter, False)
        if rank == 0:
            logging.info("Epoch[%d] Validation-accuracy=%f", epoch, vacc)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
