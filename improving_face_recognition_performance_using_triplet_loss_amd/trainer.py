"""The triplet training step of BASELINE config 2/4, without autograd in the loop.

One step = what the reference's hot loop does (ref: train_efm.py:225-245, pre-trained_efm_v3.py:193-212):
forward of [B/2 anchors ; B/2 positives] -> negatives picked from the anchor half and DETACHED ->
TripletLoss(margin) per anchor -> backward -> gradient sum across GPUs -> optimiser step with
rescale = 1/(global anchors) — as a fixed sequence of HIP launches on one stream.
"""
import torch

from . import efm_symbol, ops
from .dist import BucketReducer
from .plan import Plan


class TripletTrainer:
    def __init__(self, batch, image=112, in_channels=3, embed_dim=128, margin=0.2, optimizer="sgd", lr=2.4e-4, wd=1e-5,
                 device="cuda", seed=42, process_group=None, n_buckets=6, normalize=True, outputs=None, fuse=None,
                 autotune=False, dtype="f32", tuning=None):
        """tuning: a kernel-selection table (plan.tuning_table() / tuning.load()) applied instead of timing candidates — the
        reproducible form of `autotune=True`."""
        self.batch, self.half = batch, batch // 2
        self.margin, self.lr, self.wd = margin, lr, wd
        self.optimizer = optimizer
        self.device = torch.device(device)
        self.plan = Plan(outputs if outputs is not None else efm_symbol.embedding_net(embed_dim, normalize),
                         (batch, in_channels, image, image), device, fuse=fuse, dtype=dtype)
        if tuning is not None:
            self.plan.apply_tuning(tuning)
        elif autotune:
            self.plan.autotune()
        self.flat = self.plan.new_flat()
        self.plan.init_xavier(self.flat, seed)
        self.grad = torch.zeros_like(self.flat)
        self.t = 0
        if optimizer == "adam":
            self.m = torch.zeros_like(self.flat)
            self.v = torch.zeros_like(self.flat)
        elif optimizer != "sgd":
            raise ValueError(optimizer)
        ranges = []
        for ps in self.plan.params.values():
            if ps.kind == "weight":
                ranges.append([ps.offset, ps.offset + ps.numel])
            else:
                ranges[-1][1] = ps.offset + ps.numel
        bounds = BucketReducer.make_boundaries(ranges, self.flat.numel(), n_buckets)
        self.reducer = BucketReducer(self.grad, bounds, process_group)
        self.world = self.reducer.world
        self._ones = torch.ones(self.half, dtype=torch.float32, device=self.device)
        self.last = {}

    def forward_loss(self, x, neg_idx):
        emb, feat = self.plan.forward(x, self.flat, train=True)
        a, p = emb[: self.half], emb[self.half:]
        n = ops.gather_rows(emb, neg_idx)  # a copy: no gradient reaches the negatives (ref: train_efm.py:238-239)
        loss = ops.triplet_fwd(a, p, n, self.margin)
        self.last = {"emb": emb, "feat": feat, "a": a, "p": p, "n": n, "loss": loss}
        return loss

    def backward(self, demb=None):
        """`demb` (B, D) overrides the loss's own upstream gradient (used by the parity tests as a well-conditioned probe)."""
        L = self.last
        if demb is None:
            demb = torch.zeros_like(L["emb"])
            # vector backward = ones head-gradient (Gluon loss.backward()); the 1/B mean is the optimiser's rescale
            ops.triplet_bwd(L["a"], L["p"], L["n"], L["loss"], self._ones, da=demb[: self.half], dp=demb[self.half:])
        self.plan.backward([demb, None], self.flat, self.grad, ready_cb=self.reducer.ready)
        self.reducer.finish()

    def update(self):
        self.t += 1
        rescale = 1.0 / (self.half * self.world)  # Trainer.step(batch_size) / rescale_grad (ref: mutli_gpu_v3.py:159)
        if self.optimizer == "sgd":
            ops.sgd_update(self.flat, self.grad, self.lr, self.wd, rescale)
        else:
            ops.adam_update(self.flat, self.grad, self.m, self.v, self.lr, self.t, wd=self.wd, rescale=rescale)

    def step(self, x, neg_idx):
        loss = self.forward_loss(x, neg_idx)
        self.backward()
        self.update()
        return loss

    def cosine_log(self):
        """(s_ap, s_an) of the last step — the rows the reference appends to cosine_similarity.csv (train_efm.py:251-255)."""
        L = self.last
        return ops.cosine_pairs(L["a"], L["p"], L["n"])


class MiningTripletTrainer(TripletTrainer):
    """Batch-all anchors with in-batch semi-hard negatives (north star; BASELINE configs[2]): every image is an anchor,
    its positive is the next image of the same identity, its negative comes from the batch cosine matrix
    (`efm_gram_cosine` + `efm_mine_semihard`), detached like the reference's negatives.  All on device, no host sync."""

    def set_labels(self, labels):
        import numpy as np
        lab = np.asarray(labels.cpu() if torch.is_tensor(labels) else labels)
        pos = np.arange(len(lab), dtype=np.int32)
        for v in np.unique(lab):
            idx = np.nonzero(lab == v)[0]
            pos[idx] = np.roll(idx, -1)
        inv = np.empty_like(pos)
        inv[pos] = np.arange(len(lab), dtype=np.int32)
        self.labels = torch.as_tensor(lab.astype(np.int32)).to(self.device)
        self.pos = torch.as_tensor(pos).to(self.device)
        self.inv_pos = torch.as_tensor(inv).to(self.device)
        self.anchor = torch.arange(len(lab), dtype=torch.int32, device=self.device)
        self._ones = torch.ones(len(lab), dtype=torch.float32, device=self.device)
        self.n_anchor = len(lab)

    def forward_loss(self, x, neg_idx=None):
        emb, feat = self.plan.forward(x, self.flat, train=True)
        g = ops.gram_cosine(emb)
        neg = ops.mine_semihard(g, self.labels, self.anchor, self.pos) if neg_idx is None else neg_idx
        loss = ops.triplet_indexed_fwd(emb, self.pos, neg, self.margin)
        self.last = {"emb": emb, "feat": feat, "neg": neg, "loss": loss, "gram": g}
        return loss

    def backward(self, demb=None):
        L = self.last
        if demb is None:
            demb = ops.triplet_indexed_bwd(L["emb"], self.pos, L["neg"], self.inv_pos, L["loss"], self._ones)
        self.plan.backward([demb, None], self.flat, self.grad, ready_cb=self.reducer.ready)
        self.reducer.finish()

    def update(self):
        self.t += 1
        rescale = 1.0 / (self.n_anchor * self.world)
        if self.optimizer == "sgd":
            ops.sgd_update(self.flat, self.grad, self.lr, self.wd, rescale)
        else:
            ops.adam_update(self.flat, self.grad, self.m, self.v, self.lr, self.t, wd=self.wd, rescale=rescale)
