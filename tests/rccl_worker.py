"""One rank of the RCCL rehearsal (started by tests/test_rccl_gpu.py as a child process, one per run).

The product's data-parallel step — TripletTrainer.step -> Plan.backward (three streams: data gradients on the main stream,
weight-gradient slabs on the side stream, slab reductions + the collectives on the reduce stream) -> BucketReducer.ready ->
dist.all_reduce(async_op=True) on the `nccl` backend (= RCCL) -> finish() -> optimiser with rescale = 1/(global anchors)
(ref: mutli_gpu_v3.py:117,153-162) — at the benchmark's own size with the COMMITTED kernel selection.

    rccl_worker.py <out.pt> <batch> <image> <steps> <mode>      mode = "rccl" | "plain"

"rccl": a 1-rank process group on the nccl backend with EFM_FORCE_ALLREDUCE=1, so every bucket really goes through
ProcessGroupNCCL (its own streams, events and stream-ordering rules); "plain": no process group, no collective.  A 1-rank sum is
the identity, so both runs must leave bit-identical parameters."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, batch, image, steps, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    if mode == "rccl":
        assert os.environ.get("EFM_FORCE_ALLREDUCE") == "1"
        dist.init_process_group("nccl", device_id=device)   # exactly bench.py's call for N > 1
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from improving_face_recognition_performance_using_triplet_loss_amd import synth, tuning
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    table, _ = tuning.load("efm", batch, image, "f32")
    tr = TripletTrainer(batch, image=image, optimizer="sgd", lr=2.4e-4, wd=1e-5, margin=0.2, device=device, seed=42, tuning=table)
    assert tr.plan.two_streams and tr.plan.reduce_stream          # the three-stream backward
    assert tr.reducer.force == (mode == "rccl")
    labels = synth.parity_labels(batch)
    batches = [(synth.images(batch, 3, image, 1234 + s, device), synth.negative_indices(labels, 77 + s).to(device)) for s in range(2)]
    n_coll, orders, losses = 0, [], []
    for i in range(steps):
        losses.append(tr.step(*batches[i % 2]).clone())
        orders.append(list(tr.reducer.last_launch_order))
        n_coll += tr.reducer.last_collectives
    torch.cuda.synchronize()
    torch.save({"flat": tr.flat.cpu(), "grad": tr.grad.cpu(), "loss": torch.stack(losses).cpu(), "orders": orders,
                "nbuckets": len(tr.reducer.bounds) - 1, "collectives": n_coll, "tuned": table is not None,
                "wino": sum(bool(getattr(s, "wino_fwd", False)) for s in tr.plan.steps)}, out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
