"""torch.autograd bridges: the Python host drives PyTorch autograd, every differentiable op below is a HIP kernel.

`plan_apply` runs a whole compiled network (plan.Plan) as ONE autograd node; the small head / loss ops are
individual Functions so that the reference's training scripts can compose them freely
(ref: train_efm.py:229-245, pre-trained_efm_v3.py:197-212).
"""
import torch

from . import _lib, ops
from ._lib import pad4


def _pad_cols(t, width):
    """(rows, d) -> contiguous (rows, width) with zero pad columns (plumbing: a device memcpy)."""
    if t.shape[1] == width and t.is_contiguous():
        return t
    out = torch.zeros((t.shape[0], width), dtype=torch.float32, device=t.device)
    out[:, : t.shape[1]].copy_(t)
    return out


class _PlanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flat, plan, train):
        outs = plan.forward(x, flat, train=train)
        ctx.plan, ctx.flat = plan, flat
        ctx.widths = [st.shape[0] for st in plan.outputs]
        ctx.vec = [st.shape[1:] == (1, 1) for st in plan.outputs]
        res = []
        for o, w, v in zip(outs, ctx.widths, ctx.vec):
            res.append(o[:, :w] if v else ops.nhwc_to_nchw(o, w))
        return tuple(res)

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        grads = []
        for g, w, v, st in zip(gouts, ctx.widths, ctx.vec, plan.outputs):
            if g is None:
                grads.append(None)
            elif v:
                grads.append(_pad_cols(g.to(torch.float32), pad4(w)))
            else:
                grads.append(ops.nchw_to_nhwc(g.contiguous()))
        gflat = torch.empty_like(ctx.flat)
        plan.backward(grads, ctx.flat, gflat)
        return None, gflat, None, None


def plan_apply(plan, x, flat, train=True):
    """Run `plan` on NCHW input `x` with packed parameters `flat`; returns one tensor per plan output
    ((B, C) for vector outputs, NCHW otherwise).  Differentiable w.r.t. `flat`."""
    return _PlanFn.apply(x, flat, plan, train)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = x.contiguous()
        y, n = ops.l2norm_fwd(x, mode)
        ctx.save_for_backward(y, n)
        ctx.mode = mode
        return y

    @staticmethod
    def backward(ctx, g):
        y, n = ctx.saved_tensors
        return ops.l2norm_bwd(y, n, g.contiguous(), ctx.mode), None


def l2_normalize(x, mode="row"):
    """mode 'row': x[i]/||x[i]|| (ref: final_efm.py:240-243); 'frobenius': x/||x||_F — what `anc/mx.nd.norm(anc)`
    computes in train_efm.py:241."""
    return _L2Norm.apply(x, _lib.L2_ROW if mode == "row" else _lib.L2_FROBENIUS)


class _Triplet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, p, n, margin):
        a, p, n = a.contiguous(), p.contiguous(), n.contiguous()
        loss = ops.triplet_fwd(a, p, n, margin)
        ctx.save_for_backward(a, p, n, loss)
        ctx.need_n = ctx.needs_input_grad[2]
        return loss

    @staticmethod
    def backward(ctx, g):
        a, p, n, loss = ctx.saved_tensors
        da, dp, dn = ops.triplet_bwd(a, p, n, loss, g.contiguous(), need_dn=ctx.need_n)
        return da, dp, dn, None


def triplet_loss(anchor, positive, negative, margin):
    """gluon.loss.TripletLoss: relu(sum (p-a)^2 - (n-a)^2 + margin), shape (B,) (ref: train_efm.py:210,241)."""
    return _Triplet.apply(anchor, positive, negative, float(margin))


def gather_negatives(emb, idx):
    """neg[i] = emb[idx[i]] as a COPY without gradient — the reference round-trips through NumPy
    (train_efm.py:238-239), which detaches; the stop-gradient is kept, the round trip is not."""
    return ops.gather_rows(emb.detach().contiguous(), idx.to(torch.int32).contiguous())


def cosine_dist(anc, pos, neg, batch_size=None):
    """Per-row cosine similarities (s_ap, s_an) as two device vectors (ref: train_efm.py:26-34 returns two lists)."""
    return ops.cosine_pairs(anc.detach().contiguous(), pos.detach().contiguous(), neg.detach().contiguous())


class _Dense(torch.autograd.Function):
    """y = x W^T (+ b) through the implicit-GEMM kernels (a 1x1 convolution over a (rows,1,1,C) map)."""

    @staticmethod
    def forward(ctx, x, wp, bias, desc):
        xp = _pad_cols(x, desc.cin_p)
        y = ops.conv_fwd(desc, xp, wp, bias)
        ctx.save_for_backward(xp, wp)
        ctx.desc, ctx.has_bias, ctx.d_in = desc, bias is not None, x.shape[1]
        return y.view(desc.batch, desc.cout_p)[:, : desc.cout]

    @staticmethod
    def backward(ctx, g):
        xp, wp = ctx.saved_tensors
        d = ctx.desc
        gp = _pad_cols(g, d.cout_p)
        dw, db = ops.conv_bwd_weight(d, xp, gp, want_bias=ctx.has_bias)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_bwd_data(d, gp, ops.conv_make_dgrad_weights(d, wp)).view(d.batch, d.cin_p)[:, : ctx.d_in]
        return dx, dw, db, None


def dense(x, w_packed, bias, desc):
    return _Dense.apply(x, w_packed, bias, desc)
