"""torch.nn modules over the HIP kernels — the Gluon-shaped host API the reference's scripts are written against
(nn.Dense, gluon.loss.TripletLoss, a HybridBlock-like network wrapper, a Trainer)."""
import math

import numpy as np
import os

import torch

from . import functional as F_
from . import ops
from .plan import Plan


class TripletLoss(torch.nn.Module):
    """gluon.loss.TripletLoss(margin)(anchor, positive, negative) -> (B,)  (ref: train_efm.py:210)."""

    def __init__(self, margin=1.0):
        super().__init__()
        self.margin = margin

    def forward(self, anchor, positive, negative):
        return F_.triplet_loss(anchor, positive, negative, self.margin)


class Dense(torch.nn.Module):
    """nn.Dense(units, use_bias) with deferred input size and Gluon-Xavier init (ref: pre-trained_efm_v3.py:180-184).
    The weight lives in the kernels' packed layout; `.weight_mx()` returns MXNet's (units, in_units)."""

    def __init__(self, units, use_bias=True, in_units=None):
        super().__init__()
        self.units, self.use_bias, self.in_units = units, use_bias, in_units
        self.weight = None
        self.bias = None
        self._desc = {}

    def _materialise(self, in_units, device):
        self.in_units = in_units
        d = ops.conv_desc(1, 1, 1, in_units, self.units, 1, 1, 0, 0)
        scale = math.sqrt(3.0 / ((in_units + self.units) / 2.0))  # init.Xavier(): uniform, factor avg, magnitude 3
        w = (torch.rand((self.units, in_units, 1, 1), device=device) * 2 - 1) * scale
        self.weight = torch.nn.Parameter(ops.conv_pack_weights(d, w))
        if self.use_bias:
            self.bias = torch.nn.Parameter(torch.zeros(d.n_pad16, device=device))

    def desc(self, rows):
        d = self._desc.get(rows)
        if d is None:
            d = self._desc[rows] = ops.conv_desc(rows, 1, 1, self.in_units, self.units, 1, 1, 0, 0)
        return d

    def forward(self, x):
        if self.weight is None:
            self._materialise(x.shape[1], x.device)
        return F_.dense(x, self.weight, self.bias, self.desc(x.shape[0]))

    def weight_mx(self):
        return ops.conv_unpack_weights(self.desc(1), self.weight.detach()).view(self.units, self.in_units)

    def load_weight_mx(self, w):
        w = torch.as_tensor(np.asarray(w), dtype=torch.float32)
        if self.weight is None:
            self._materialise(w.shape[1], torch.device("cuda"))
        with torch.no_grad():
            ops.conv_pack_weights_into(self.desc(1), w.to(self.weight.device).reshape(self.units, self.in_units, 1, 1).contiguous(), self.weight)


class BatchNorm(torch.nn.Module):
    """Gluon nn.BatchNorm() on a (N, C) feature (ref: lightcnn.py:113-114,130): momentum 0.9, eps 1e-5, gamma ones / beta zeros.
    Training mode normalises with the batch mean and the BIASED batch variance and tracks
    `running = 0.9*running + 0.1*batch` with that same biased variance — MXNet's rule; torch.nn.BatchNorm1d tracks the unbiased
    one, which would make the `running_var` written by save_parameters differ from the reference's by N/(N-1)."""

    def __init__(self, in_channels, momentum=0.9, epsilon=1e-5):
        super().__init__()
        self.momentum, self.eps = momentum, epsilon
        self.gamma = torch.nn.Parameter(torch.ones(in_channels))
        self.beta = torch.nn.Parameter(torch.zeros(in_channels))
        self.register_buffer("running_mean", torch.zeros(in_channels))
        self.register_buffer("running_var", torch.ones(in_channels))

    def forward(self, x):
        if not self.training:
            return torch.nn.functional.batch_norm(x, self.running_mean, self.running_var, self.gamma, self.beta, False, 0.0, self.eps)
        with torch.no_grad():
            mean = x.mean(dim=0)
            var = x.var(dim=0, unbiased=False)
            self.running_mean.mul_(self.momentum).add_(mean, alpha=1 - self.momentum)
            self.running_var.mul_(self.momentum).add_(var, alpha=1 - self.momentum)
        return torch.nn.functional.batch_norm(x, None, None, self.gamma, self.beta, True, 0.0, self.eps)


class SymbolNet(torch.nn.Module):
    """A compiled `graph.Sym` network as a torch module (the role of gluon.SymbolBlock / HybridBlock.hybridize()).
    One flat nn.Parameter holds every weight in packed layout; plans are compiled per batch size on first use.
    autotune: time the kernel candidates of every layer once per batch size (Plan.autotune(): Winograd vs direct kernels, tilings;
    a few seconds at training batch sizes) — None = the EFM_AUTOTUNE environment variable ("1" turns it on); batches below 64 (the
    2-image plan that only initialises the parameters, smoke runs) are never tuned."""

    def __init__(self, outputs, in_channels, image, device="cuda", seed=42, init="xavier", fuse=None, autotune=None):
        super().__init__()
        self._outputs, self.in_channels, self.image = outputs, in_channels, image
        self._fuse = fuse
        self._autotune = (os.environ.get("EFM_AUTOTUNE", "0") == "1") if autotune is None else bool(autotune)
        self._plans = {}
        self._dev = torch.device(device)
        p0 = self.plan(2)
        flat = p0.new_flat()
        if init == "xavier":
            p0.init_xavier(flat, seed)
        self.flat = torch.nn.Parameter(flat)

    def plan(self, batch):
        p = self._plans.get(batch)
        if p is None:
            p = self._plans[batch] = Plan(self._outputs, (batch, self.in_channels, self.image, self.image), self._dev, fuse=self._fuse)
            if self._autotune and batch >= 64 and self._dev.type == "cuda":
                p.autotune()
        return p

    def forward(self, x):
        return F_.plan_apply(self.plan(x.shape[0]), x, self.flat, self.training or torch.is_grad_enabled())

    def export_params(self):
        return self.plan(2).export_params(self.flat.detach())

    def load_params(self, params):
        with torch.no_grad():
            self.plan(2).load_params(self.flat, params)


class FactorScheduler:
    """mx.lr_scheduler.FactorScheduler(step, factor, stop_factor_lr) (ref: train_efm.py:212)."""

    def __init__(self, step, factor=1.0, stop_factor_lr=1e-8, base_lr=0.01):
        self.step, self.factor, self.stop_factor_lr, self.base_lr = max(int(step), 1), factor, stop_factor_lr, base_lr
        self.count = 0

    def __call__(self, num_update):
        while num_update > self.count + self.step:
            self.count += self.step
            self.base_lr *= self.factor
            if self.base_lr < self.stop_factor_lr:
                self.base_lr = self.stop_factor_lr
        return self.base_lr


class Trainer:
    """gluon.Trainer(params, 'sgd' | 'adam', {...}).step(batch_size) on the fused HIP optimiser kernels
    (ref: train_efm.py:213-214,245; pre-trained_efm_v3.py:185,212).  Parameters whose element count is not a multiple
    of 4 (none of this package's packed buffers) fall back to a torch update."""

    def __init__(self, params, optimizer="sgd", learning_rate=0.01, wd=0.0, lr_scheduler=None, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.params = [p for p in params]
        self.optimizer, self.lr, self.wd = optimizer, learning_rate, wd
        self.sched = lr_scheduler
        if self.sched is not None:
            self.sched.base_lr = learning_rate
        self.beta1, self.beta2, self.eps = beta1, beta2, epsilon
        self.t = 0
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, batch_size, ignore_stale_grad=False):
        self.t += 1
        lr = self.sched(self.t) if self.sched is not None else self.lr
        rescale = 1.0 / batch_size
        for p in self.params:
            if p.grad is None:
                continue
            w, g = p.data, p.grad.contiguous()
            fused = w.is_cuda and w.is_contiguous() and w.numel() % 4 == 0 and w.dtype == torch.float32
            if self.optimizer == "sgd":
                if fused:
                    ops.sgd_update(w, g, lr, self.wd, rescale)
                else:
                    w.sub_(lr * (rescale * g + self.wd * w))
            else:
                st = self.state.setdefault(id(p), (torch.zeros_like(w), torch.zeros_like(w)))
                if fused:
                    ops.adam_update(w, g, st[0], st[1], lr, self.t, self.beta1, self.beta2, self.eps, self.wd, rescale)
                else:
                    gr = rescale * g + self.wd * w
                    st[0].mul_(self.beta1).add_(gr, alpha=1 - self.beta1)
                    st[1].mul_(self.beta2).addcmul_(gr, gr, value=1 - self.beta2)
                    lr_t = lr * math.sqrt(1 - self.beta2 ** self.t) / (1 - self.beta1 ** self.t)
                    w.sub_(lr_t * st[0] / (st[1].sqrt() + self.eps))
        self.zero_grad()
