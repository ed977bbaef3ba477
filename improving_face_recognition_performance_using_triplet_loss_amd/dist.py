"""Data-parallel gradient exchange: bucketed all-reduce of the flat gradient buffer, overlapped with backward.

Replaces `mx.mod.Module(context=[gpu0, gpu1])` + `kvstore='local'` (ref: mutli_gpu_v3.py:117,153,158): one
process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  The payload
is one flat fp32 buffer (36 MB for EFM-29), cut into a few contiguous buckets; `Plan.backward` reports each
parameter slice as soon as its gradient is final (late layers first) and a bucket's all-reduce is launched the
moment its last slice arrives, so the exchange of the deep layers hides under the backward of the shallow
ones.  xGMI is point-to-point: few large messages beat many small ones, hence 4-8 buckets, not per-layer.
Sum, not mean: the 1/global_batch scale is the optimiser's `rescale` (ref: mutli_gpu_v3.py:159).
"""
import os

import torch
import torch.distributed as dist


class BucketReducer:
    def __init__(self, grad_flat, boundaries, process_group=None):
        """boundaries: sorted offsets [0, ..., numel] of the buckets inside `grad_flat`."""
        self.grad = grad_flat
        self.bounds = list(boundaries)
        assert self.bounds[0] == 0 and self.bounds[-1] == grad_flat.numel()
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # rehearsal knob: issue the collectives even with a single rank (exercises the RCCL stream semantics on a 1-GPU box)
        self.force = bool(os.environ.get("EFM_FORCE_ALLREDUCE")) and dist.is_initialized()
        self._pending = None
        self._handles = []
        self.launch_order = []
        self.last_launch_order = []
        self.last_collectives = 0   # all-reduces actually issued by the last finished step (0 with one rank and no force)
        self.reset()

    @staticmethod
    def make_boundaries(param_ranges, numel, n_buckets):
        """Cut [0, numel) at parameter boundaries into ~equal buckets."""
        target = numel / float(n_buckets)
        cuts, nxt = [0], target
        for lo, hi in param_ranges:
            if hi >= nxt and hi < numel:
                cuts.append(hi)
                nxt = hi + target
        cuts.append(numel)
        return sorted(set(cuts))

    def reset(self):
        self._covered = [0] * (len(self.bounds) - 1)
        self._handles = []
        self.launch_order = []

    def _bucket_of(self, off):
        for k in range(len(self.bounds) - 1):
            if self.bounds[k] <= off < self.bounds[k + 1]:
                return k
        raise ValueError(off)

    def ready(self, lo, hi):
        """Slice [lo, hi) of the flat gradient is final."""
        while lo < hi:
            k = self._bucket_of(lo)
            end = min(hi, self.bounds[k + 1])
            self._covered[k] += end - lo
            if self._covered[k] == self.bounds[k + 1] - self.bounds[k]:
                self._launch(k)
            lo = end

    def _launch(self, k):
        self.launch_order.append(k)
        if self.world == 1 and not self.force:
            return
        view = self.grad.narrow(0, self.bounds[k], self.bounds[k + 1] - self.bounds[k])
        self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """Make the current stream wait for every launched all-reduce; checks that every bucket was launched."""
        missing = [k for k in range(len(self.bounds) - 1) if k not in self.launch_order]
        if missing:
            raise RuntimeError("buckets %s never became ready" % missing)
        for h in self._handles:
            h.wait()
        self.last_collectives = len(self._handles)
        self.last_launch_order = list(self.launch_order)
        self.reset()
