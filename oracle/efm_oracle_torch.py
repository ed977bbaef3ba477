"""CPU ORACLE #2 (test infrastructure, NOT product code) — torch-CPU restatement of the reference graph.

Independent of oracle/efm_oracle.py: built from torch.nn.functional ops (oneDNN convolutions) and torch
autograd, so the two restatements protect each other against typos (they cannot protect against a shared
wrong [MX-assumed] reading of MXNet — parity with MXNet itself stays UNPINNED, see efm_oracle.py).
It is also the `cpu_baseline` ("port") that bench.py times on the GPU box's host cores.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F

GROUPS = [(0, 99, 5, 2, "1", 0), (99, 198, 3, 1, "2", 1), (198, 387, 3, 1, "3", 2), (387, 261, 3, 1, "4", 3),
          (261, 261, 3, 1, "5", 4)]  # ref: efm_symbol.py:84-92


def mfm3(x):
    # ref: efm_symbol.py:25-30 — torch.maximum/minimum split a tie gradient 50/50, MXNet gives it to the lhs;
    # ties have measure zero on continuous data, tie behaviour is pinned in the NumPy oracle's KATs instead.
    s0, s1, s2 = torch.chunk(x, 3, dim=1)
    return torch.cat([torch.maximum(torch.maximum(s0, s1), s2), torch.minimum(torch.minimum(s0, s1), s2)], dim=1)


def mfm2(x):
    s0, s1 = torch.chunk(x, 2, dim=1)
    return torch.maximum(s0, s1)


def res_block(p, data, lname):
    # ref: efm_symbol.py:22-44
    e = mfm3(data)
    c = F.conv2d(e, p["conv%s_res_weight" % lname], p["conv%s_res_bias" % lname], padding=1)
    e = mfm3(c)
    c = F.conv2d(e, p["conv%s_res_r_weight" % lname], p["conv%s_res_r_bias" % lname], padding=1)
    return data + c


def efm29_forward(p, x):
    """(B,C,H,W) -> 342-d feature; `p` maps MXNet parameter names to tensors."""
    cur = x
    for num_r, num, k, pad, layer, tar in GROUPS:
        if num_r > 0:
            for i in range(tar):
                cur = res_block(p, cur, layer if i == 0 else layer + str(i))
            cur = mfm3(F.conv2d(cur, p["conv%s_r_weight" % layer], p["conv%s_r_bias" % layer]))
        cur = F.conv2d(cur, p["conv%s_weight" % layer], p["conv%s_bias" % layer], padding=pad)
        cur = F.max_pool2d(mfm3(cur), 2, 2)  # ceil_mode False = MXNet 'valid'
    fc1 = F.linear(cur.flatten(1), p["fc1_weight"], p["fc1_bias"])
    return mfm3(fc1)


def triplet_loss(a, p, n, margin):
    return F.relu(((p - a) ** 2 - (n - a) ** 2).sum(dim=1) + margin)


def train_step(p, w_head, x, neg_idx, margin, demb=None):
    """Reference-layout step (see efm_oracle.train_step_loss): returns (loss vector, emb, feat); gradients land in
    .grad of every tensor of `p` and of `w_head`.  `demb` (optional) replaces the loss's own upstream gradient."""
    feat = efm29_forward(p, x)
    yn = feat / feat.norm(dim=1, keepdim=True)
    emb = yn @ w_head.t()
    h = x.shape[0] // 2
    a, pos = emb[:h], emb[h:]
    n = emb[neg_idx].detach()  # negatives are copied through NumPy in the reference (train_efm.py:238-239)
    loss = triplet_loss(a, pos, n, margin)
    if demb is None:
        loss.sum().backward()  # vector backward = ones head-gradient
    else:
        emb.backward(demb)
    return loss.detach(), emb.detach(), feat.detach()


LIGHTCNN9_PLAN = [("1", 0, 96, 5, 2, True), ("2", 96, 192, 3, 1, True), ("3", 192, 384, 3, 1, True), ("4", 384, 256, 3, 1, False),
                  ("5", 256, 256, 3, 1, True)]


def lightcnn9_forward(p, x):
    """Build-defined LightCNN-9 (BASELINE configs[2]): 2-way MFM everywhere (ref branch: efm_symbol.py:62-64,76-77)."""
    cur = x
    for layer, num_r, num, k, pad, pool in LIGHTCNN9_PLAN:
        if num_r:
            cur = mfm2(F.conv2d(cur, p["conv%s_r_weight" % layer], p["conv%s_r_bias" % layer]))
        cur = mfm2(F.conv2d(cur, p["conv%s_weight" % layer], p["conv%s_bias" % layer], padding=pad))
        if pool:
            cur = F.max_pool2d(cur, 2, 2)
    return mfm2(F.linear(cur.flatten(1), p["fc1_weight"], p["fc1_bias"]))


def mining_step(forward, p, x, labels, pos, margin, backward=True):
    """Semi-hard step on any feature network: emb = rownorm(forward), cosine matrix, TF-addons semi-hard rule, indexed
    triplet loss with detached negatives; returns (loss, emb, neg); gradients land in .grad of `p`.  backward=False returns
    the attached (loss, emb) instead so that a test can push its own upstream gradient through emb."""
    from oracle import efm_oracle as O
    feat = forward(p, x)
    emb = feat / feat.norm(dim=1, keepdim=True)
    e = emb.detach().numpy()
    g = O.gram_cosine(e)
    import numpy as np
    neg = O.mine_semihard(g, labels, np.arange(len(labels)), pos)
    ok = torch.as_tensor(neg >= 0)
    n = emb[torch.as_tensor(np.where(neg >= 0, neg, 0))].detach()
    loss = torch.where(ok, triplet_loss(emb, emb[torch.as_tensor(pos.astype(np.int64))], n, margin), torch.zeros((), dtype=emb.dtype))
    if not backward:
        return loss, emb, neg
    loss.sum().backward()
    return loss.detach(), emb.detach(), neg


# ---- emulation of the bf16 tensor-core path (BASELINE configs[2]): what is rounded to bf16 and where ----------------------
class _RoundBoth(torch.autograd.Function):
    """A tensor STORED in bf16: the value is rounded going forward, its gradient is rounded coming back."""
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    """bf16 copy of an fp32 master weight: rounded going forward, full-precision gradient."""
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """fp32 accumulator consumed by the fused epilogue: untouched going forward, its gradient (dy) is stored in bf16."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def lightcnn9_layers():
    """LIGHTCNN9_PLAN as the flat (name, kernel pad, pool) list the generic MFM2-stack functions take."""
    out = []
    for layer, num_r, num, k, pad, pool in LIGHTCNN9_PLAN:
        if num_r:
            out.append(("conv%s_r" % layer, 0, False))
        out.append(("conv%s" % layer, pad, pool))
    return out


# BASELINE configs[4]'s build-defined deeper CNN (efm_symbol.DEEPCNN_LAYERS restated: name, pad, pool)
DEEPCNN_LAYERS = [("conv1", 2, True), ("conv2_r", 0, False), ("conv2a", 1, False), ("conv2b", 1, True),
                  ("conv3_r", 0, False), ("conv3a", 1, False), ("conv3b", 1, True),
                  ("conv4_r", 0, False), ("conv4a", 1, False), ("conv4b", 1, False),
                  ("conv5_r", 0, False), ("conv5a", 1, False), ("conv5b", 1, True)]


def mfm2_stack_forward(p, x, layers):
    """[conv -> MFM2 (-> 2x2 max pool)] per layer, then fc1 -> MFM2 (kernel sizes come from the weight shapes)."""
    cur = x
    for name, pad, pool in layers:
        cur = mfm2(F.conv2d(cur, p[name + "_weight"], p[name + "_bias"], padding=pad))
        if pool:
            cur = F.max_pool2d(cur, 2, 2)
    return mfm2(F.linear(cur.flatten(1), p["fc1_weight"], p["fc1_bias"]))


def deepcnn_forward(p, x):
    return mfm2_stack_forward(p, x, DEEPCNN_LAYERS)


def mfm2_stack_forward_bf16(p, x, layers, forced=None, outs=None):
    """An MFM2 stack (LightCNN-9, the deeper CNN) as the bf16 plan computes it: bf16 activations and weights as conv operands, fp32 accumulate + bias + MFM2 + pool,
    result stored in bf16 — except the last layer, which feeds the fp32 head.

    A network of rounding steps is chaotic: an accumulator that differs in the last fp32 bit lands on the other side of a bf16
    rounding boundary now and then, the next layer spreads that 2^-8 step over all its outputs, and after four layers device and
    emulation agree only to the bf16 noise itself (3e-3), which in turn flips arg-max routes in backward (5 % gradient noise).
    `forced` (the device's own stored activation of every layer, NCHW) removes the chaos without removing the check: each layer
    then starts from the device's input values (gradients still flow through the emulation's chain), its own output — appended
    to `outs` before the substitution — must equal the device's up to isolated one-ulp roundings, and the backward of the whole
    chain is compared with routes that agree."""
    k = 0

    def store(z):
        nonlocal k
        cur = _RoundBoth.apply(z)
        if outs is not None:
            outs.append(cur.detach())
        if forced is not None:
            cur = cur + (forced[k] - cur).detach()
        k += 1
        return cur

    cur = _RoundBoth.apply(x)
    for name, pad, pool in layers:
        y = F.conv2d(cur, _RoundFwd.apply(p[name + "_weight"]), p[name + "_bias"], padding=pad)
        z = mfm2(_RoundBwd.apply(y))
        if pool:
            z = F.max_pool2d(z, 2, 2)
        cur = store(z)
    y = F.linear(cur.flatten(1), _RoundFwd.apply(p["fc1_weight"]), p["fc1_bias"])
    return mfm2(_RoundBwd.apply(y))


def lightcnn9_forward_bf16(p, x, forced=None, outs=None):
    return mfm2_stack_forward_bf16(p, x, lightcnn9_layers(), forced, outs)


def deepcnn_forward_bf16(p, x, forced=None, outs=None):
    return mfm2_stack_forward_bf16(p, x, DEEPCNN_LAYERS, forced, outs)


def efm29_forward_bf16(p, x):
    """EFM-29 as the bf16 plan computes it (free-running emulation): every STORED activation is bf16 — the fused conv -> MFM3 (-> pool)
    results, the residual sums `data + conv_res_r(...)` (one rounding, after the fp32 add in the epilogue), the stand-alone MFM3 of a
    residual-block input (max / min of bf16 values: exact) — weights are bf16 copies of the fp32 masters, accumulation / bias / MFM /
    pooling decisions are fp32, and the 342-d feature that feeds the fp32 head stays fp32.  Gradients are rounded where they are
    stored (the `_Round*` functions)."""
    def conv(cur, name, pad):
        return _RoundBwd.apply(F.conv2d(cur, _RoundFwd.apply(p[name + "_weight"]), p[name + "_bias"], padding=pad))

    def block(data, lname):
        e = mfm3(data)                                        # stored bf16 (exact)
        e = _RoundBoth.apply(mfm3(conv(e, "conv%s_res" % lname, 1)))
        return _RoundBoth.apply(data + conv(e, "conv%s_res_r" % lname, 1))

    cur = _RoundBoth.apply(x)
    for num_r, num, k, pad, layer, tar in GROUPS:
        if num_r > 0:
            for i in range(tar):
                cur = block(cur, layer if i == 0 else layer + str(i))
            cur = _RoundBoth.apply(mfm3(conv(cur, "conv%s_r" % layer, 0)))
        cur = _RoundBoth.apply(F.max_pool2d(mfm3(conv(cur, "conv%s" % layer, pad)), 2, 2))
    fc1 = _RoundBwd.apply(F.linear(cur.flatten(1), _RoundFwd.apply(p["fc1_weight"]), p["fc1_bias"]))
    return mfm3(fc1)


# ---- the Gluon variant: LightCNN_29 (lightcnn.py:6-133) + the train_efm.py step (train_efm.py:229-245) -----------------------
def lightcnn29_forward(p, x):
    """(N,C,H,W) -> 684-d EFM feature.  The two convolutions of a res_block are the SAME tensors in every iteration
    (lightcnn.py:47-48 created once, :52-69 re-applied) — autograd accumulates their gradient over the uses."""
    cur = F.max_pool2d(mfm3(F.conv2d(x, p["g1_conv1_weight"], p["g1_conv1_bias"], padding=2)), 2, 2)      # lightcnn.py:82-83
    for gi, nb in enumerate([1, 2, 3, 4]):                                                                 # lightcnn.py:77
        g = gi + 2
        for _ in range(nb):                                                                                # lightcnn.py:52-69
            e = mfm3(cur)
            c = F.conv2d(e, p["g%d_res_conv0_weight" % g], p["g%d_res_conv0_bias" % g], padding=1)
            c = F.conv2d(mfm3(c), p["g%d_res_conv1_weight" % g], p["g%d_res_conv1_bias" % g], padding=1)
            cur = c + cur
        c = mfm3(F.conv2d(cur, p["g%d_conv0_weight" % g], p["g%d_conv0_bias" % g]))                        # lightcnn.py:20-28
        c = mfm3(F.conv2d(c, p["g%d_conv1_weight" % g], p["g%d_conv1_bias" % g], padding=1))               # lightcnn.py:29-37
        cur = F.max_pool2d(c, 2, 2)
    return mfm3(F.linear(cur.flatten(1), p["fc1_weight"], p["fc1_bias"]))                                  # lightcnn.py:111,121-128


def train_efm_step(p, x, labels, neg_idx, margin=0.2, alpha=0.1, dropout_mask=None, dropout_p=0.7):
    """train_efm.py:229-245 with autograd: returns (out, fc1_out, TL, id_loss, loss) detached; gradients of sum(loss) land in .grad."""
    feat = lightcnn29_forward(p, x)
    fc = F.batch_norm(feat, None, None, p["batchnorm0_gamma"], p["batchnorm0_beta"], training=True, eps=1e-5)   # lightcnn.py:130
    d = feat if dropout_mask is None else feat * dropout_mask / (1.0 - dropout_p)
    out = F.linear(d, p["dense1_weight"], p["dense1_bias"])                                                   # lightcnn.py:131
    b = x.shape[0] // 2
    anc, pos = fc[:b], fc[b:2 * b]
    neg = fc[neg_idx].detach()
    tl = triplet_loss(anc / anc.norm(), pos / pos.norm(), neg / neg.norm(), margin)                            # train_efm.py:241
    idl = F.cross_entropy(out[:b], labels[:b], reduction="none")                                               # train_efm.py:242
    loss = idl + alpha * tl                                                                                    # train_efm.py:243
    loss.sum().backward()
    return out.detach(), fc.detach(), tl.detach(), idl.detach(), loss.detach()
