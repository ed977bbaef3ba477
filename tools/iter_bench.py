#!/usr/bin/env python
"""Images/s of mxio.ImageRecordIter ALONE on a synthetic .rec (the input stage of train_efm.py:179-181 / mutli_gpu_v3.py), by
number of decode threads, next to the training step's rate — is the real-data path input-bound?

    python tools/iter_bench.py [--n 4096] [--size 144] [--crop 128] [--batch 256] [--fmt JPEG] [--threads 0,1,4,8,16] [--device cuda]

Writes N random-texture images of size x size (gray) as IRHeader + JPEG records (what im2rec writes), then times one epoch per
thread count: decode (PIL, pool of `preprocess_threads`) -> pinned staging -> H2D on the copy stream -> efm_crop_mirror_u8.
Prints one JSON line."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--size", type=int, default=144)
    ap.add_argument("--crop", type=int, default=128)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--fmt", default="JPEG")
    ap.add_argument("--threads", default="0,1,2,4,8,16")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    import torch
    from improving_face_recognition_performance_using_triplet_loss_amd import mxio
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp(prefix="efm_iter_")
    path = os.path.join(tmp, "synthetic.rec")
    shape = (args.size, args.size) if args.channels == 1 else (args.size, args.size, 3)
    base = [np.clip(np.kron(rng.integers(0, 256, size=(9, 9) + shape[2:]), np.ones((16, 16) + (1,) * (len(shape) - 2)))[: args.size, : args.size]
                    + rng.integers(-20, 20, size=shape), 0, 255).astype(np.uint8) for _ in range(64)]
    mxio.write_records(path, [mxio.pack_img(float(i % 1000), i, base[i % 64], fmt=args.fmt) for i in range(args.n)])
    dev = torch.device(args.device) if args.device != "cpu" else None
    res = {"records": args.n, "encoded": args.fmt, "source": "%dx%dx%d" % (args.size, args.size, args.channels), "crop": args.crop,
           "batch": args.batch, "file_MB": round(os.path.getsize(path) / 1e6, 1), "host_cpus": os.cpu_count(), "images_per_s": {}}
    for t in [int(v) for v in args.threads.split(",")]:
        it = mxio.ImageRecordIter(path, (args.channels, args.crop, args.crop), batch_size=args.batch, scale=1. / 255, rand_crop=True,
                                  rand_mirror=True, shuffle=True, seed=1, device=dev, preprocess_threads=t)
        for _ in it:      # warm-up epoch (page cache, pinned buffers, thread pool)
            pass
        if dev is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for b in it:
            n += b.data[0].shape[0]
        if dev is not None:
            torch.cuda.synchronize()
        res["images_per_s"]["threads=%d" % t] = round(n / (time.perf_counter() - t0), 1)
        it.close()
    print(json.dumps(res))
    os.remove(path)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
