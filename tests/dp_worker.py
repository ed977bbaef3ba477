"""One rank of the 2-process data-parallel rehearsal (started by tests/test_dp_gpu.py through torch.distributed.run).

Both ranks share cuda:0 (the GPU box has one card) and exchange gradients over gloo — RCCL refuses two ranks on one device; the
code path is the product's: TripletTrainer.step -> Plan.backward -> BucketReducer.ready (collectives issued from the weight-
gradient stream) -> finish() -> optimiser with rescale = 1/(global anchors) (ref: mutli_gpu_v3.py:153-159).  Each rank writes its
updated parameters, loss and bucket launch order to <out>/rank<r>.pt."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard_inputs(rank, batch, image):
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    x = synth.images(batch, 3, image, 100 + rank)
    neg = synth.negative_indices(synth.parity_labels(batch, images_per_identity=2), 5 + rank).cuda()
    return x, neg


def main():
    out, batch, image, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    tr = TripletTrainer(batch, image=image, seed=3, optimizer="sgd", lr=0.05, wd=1e-5)
    assert tr.world == 2
    x, neg = shard_inputs(rank, batch, image)
    losses, flats, grads = [], [], []
    for _ in range(steps):
        losses.append(tr.step(x, neg).clone())
        flats.append(tr.flat.cpu())
        grads.append(tr.grad.cpu())   # the all-reduced (summed over ranks) flat gradient of this step
    torch.cuda.synchronize()
    torch.save({"flat": tr.flat.cpu(), "flats": torch.stack(flats), "grad": tr.grad.cpu(), "grads": torch.stack(grads), "loss": torch.stack(losses).cpu(),
                "order": tr.reducer.last_launch_order, "nbuckets": len(tr.reducer.bounds) - 1}, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
