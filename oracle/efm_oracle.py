"""CPU ORACLE (test infrastructure, NOT product code) — NumPy restatement of the reference hot path.

PARITY UNPINNED: the arithmetic of the reference lives in Apache MXNet (un-vendored, un-pinned, not
installable here: no network) and the reference holds no tests, golden vectors or fixtures for this
path.  This file restates the *published semantics* of the MXNet operators at the
reference's call sites; every assumption is marked [MX-assumed] and has its own known-answer test in
tests/test_oracle.py so that a wrong assumption is a one-line fix.  A second, independent restatement
(oracle/efm_oracle_torch.py, torch-CPU functional ops + autograd) must agree with this one to 1e-5.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Layout here is the reference's: NCHW activations, (Cout, Cin, KH, KW) weights, float64 by default.
"""
import math

import numpy as np

# ----------------------------------------------------------------------------------------------
# Portable counter-based RNG (splitmix64) — the same generator is implemented on the device side
# (improving_face_recognition_performance_using_triplet_loss_amd/synth.py) so inputs / weights never need to be committed as fixtures.
# ----------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def splitmix64(idx, seed):
    """Vectorised splitmix64 of (seed + (idx+1)*golden); idx: uint64 array -> uint64 array."""
    z = (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed & _M64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(n, seed, offset=0):
    """n floats in [0,1) with 24-bit mantissas (exactly representable in fp32)."""
    with np.errstate(over="ignore"):
        bits = splitmix64(np.arange(offset, offset + n, dtype=np.uint64), seed)
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def uniform_pm(shape, seed, scale):
    n = int(np.prod(shape))
    return ((uniform01(n, seed) * 2.0 - 1.0) * scale).reshape(shape)


# ----------------------------------------------------------------------------------------------
# Operators.  Each fwd returns the output; each bwd takes what it needs explicitly.
# ----------------------------------------------------------------------------------------------
def _im2col(x, kh, kw, ph, pw):
    b, c, h, w = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    win = np.lib.stride_tricks.sliding_window_view(xp, (kh, kw), axis=(2, 3))  # b,c,ho,wo,kh,kw
    ho, wo = win.shape[2], win.shape[3]
    cols = win.transpose(0, 2, 3, 1, 4, 5).reshape(b * ho * wo, c * kh * kw)
    return cols, ho, wo


def conv2d(x, w, bias, pad):
    """mx.symbol.Convolution, stride 1: cross-correlation, NCHW, weight (Cout,Cin,KH,KW), bias on.
    [MX-assumed (2)]  ref: efm_symbol.py:32,41,54,62,65,67."""
    co, ci, kh, kw = w.shape
    cols, ho, wo = _im2col(x, kh, kw, pad[0], pad[1])
    y = cols @ w.reshape(co, -1).T
    if bias is not None:
        y = y + bias
    return y.reshape(x.shape[0], ho, wo, co).transpose(0, 3, 1, 2)


def conv2d_bwd(x, w, dy, pad, need_dx=True):
    co, ci, kh, kw = w.shape
    b, _, h, wd_ = x.shape
    cols, ho, wo = _im2col(x, kh, kw, pad[0], pad[1])
    dy2 = dy.transpose(0, 2, 3, 1).reshape(-1, co)
    dw = (dy2.T @ cols).reshape(w.shape)
    db = dy2.sum(0)
    dx = None
    if need_dx:
        dcols = (dy2 @ w.reshape(co, -1)).reshape(b, ho, wo, ci, kh, kw)
        dxp = np.zeros((b, ci, h + 2 * pad[0], wd_ + 2 * pad[1]), dtype=x.dtype)
        for i in range(kh):
            for j in range(kw):
                dxp[:, :, i:i + ho, j:j + wo] += dcols[:, :, :, :, i, j].transpose(0, 3, 1, 2)
        dx = dxp[:, :, pad[0]:pad[0] + h, pad[1]:pad[1] + wd_]
    return dx, dw, db


ORDER_GROUP = 0  # maximum(maximum(s0,s1), s2)   ref: efm_symbol.py:70-73, lightcnn.py:23-26
ORDER_RES = 1    # maximum(s2, maximum(s0,s1))   ref: efm_symbol.py:26-29


def mfm3(x):
    """SliceChannel(3) + maximum/minimum + Concat on axis 1.  [MX-assumed (1)]  ref: efm_symbol.py:25-30."""
    s0, s1, s2 = np.split(x, 3, axis=1)
    return np.concatenate([np.maximum(np.maximum(s0, s1), s2), np.minimum(np.minimum(s0, s1), s2)], axis=1)


def mfm3_bwd(x, dy, order=ORDER_GROUP):
    """Gradient to arg-max / arg-min; a tie goes to the lhs of each binary op.  [MX-assumed (5)]"""
    s0, s1, s2 = np.split(x, 3, axis=1)
    gmax, gmin = np.split(dy, 2, axis=1)
    m1, n1 = np.maximum(s0, s1), np.minimum(s0, s1)
    imax = np.where(s0 >= s1, 0, 1)
    imin = np.where(s0 <= s1, 0, 1)
    if order == ORDER_GROUP:
        imax = np.where(m1 >= s2, imax, 2)
        imin = np.where(n1 <= s2, imin, 2)
    else:
        imax = np.where(s2 >= m1, 2, imax)
        imin = np.where(s2 <= n1, 2, imin)
    parts = [np.where(imax == k, gmax, 0.0) + np.where(imin == k, gmin, 0.0) for k in range(3)]
    return np.concatenate(parts, axis=1)


def mfm2(x):
    """SliceChannel(2) + maximum.  ref: efm_symbol.py:63-64,76-77."""
    s0, s1 = np.split(x, 2, axis=1)
    return np.maximum(s0, s1)


def mfm2_bwd(x, dy):
    s0, s1 = np.split(x, 2, axis=1)
    return np.concatenate([np.where(s0 >= s1, dy, 0.0), np.where(s0 >= s1, 0.0, dy)], axis=1)


def maxpool2(x):
    """Pooling(max, 2x2, stride 2), pooling_convention='valid' => floor.  [MX-assumed (3)]  ref: efm_symbol.py:78."""
    b, c, h, w = x.shape
    ho, wo = h // 2, w // 2
    v = x[:, :, :2 * ho, :2 * wo].reshape(b, c, ho, 2, wo, 2)
    return v.max(axis=(3, 5))


def maxpool2_bwd(x, dy):
    """Gradient to the first maximum of each window in scan order."""
    b, c, h, w = x.shape
    ho, wo = h // 2, w // 2
    v = x[:, :, :2 * ho, :2 * wo].reshape(b, c, ho, 2, wo, 2).transpose(0, 1, 2, 4, 3, 5).reshape(b, c, ho, wo, 4)
    idx = v.argmax(axis=-1)  # numpy argmax = first maximum
    g = np.zeros_like(v)
    np.put_along_axis(g, idx[..., None], dy[..., None], axis=-1)
    dx = np.zeros_like(x)
    dx[:, :, :2 * ho, :2 * wo] = g.reshape(b, c, ho, wo, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(b, c, 2 * ho, 2 * wo)
    return dx


def fully_connected(x, w, bias):
    """FullyConnected / Dense: flatten trailing dims in NCHW order, y = x W^T + b.  [MX-assumed (4)]"""
    x2 = x.reshape(x.shape[0], -1)
    y = x2 @ w.T
    return y + bias if bias is not None else y


def fully_connected_bwd(x, w, dy):
    x2 = x.reshape(x.shape[0], -1)
    return (dy @ w).reshape(x.shape), dy.T @ x2, dy.sum(0)


def l2norm_row(x):
    """fc[i] / mx.nd.norm(fc[i])  ref: final_efm.py:240-243."""
    n = np.sqrt((x * x).sum(axis=1, keepdims=True))
    return x / n, n[:, 0]


def l2norm_row_bwd(y, n, dy):
    return (dy - y * (y * dy).sum(axis=1, keepdims=True)) / n[:, None]


def l2norm_frob(x):
    """anc / mx.nd.norm(anc): norm over ALL elements -> scalar.  [MX-assumed (6)]  ref: train_efm.py:241."""
    n = math.sqrt(float((x * x).sum()))
    return x / n, n


def l2norm_frob_bwd(y, n, dy):
    return (dy - y * (y * dy).sum()) / n


def triplet_loss(a, p, n, margin):
    """gluon.loss.TripletLoss: relu(sum_d (p-a)^2 - (n-a)^2 + margin), one value per sample.
    [MX-assumed (7)]  ref: train_efm.py:210,241; pre-trained_efm_v3.py:183,210."""
    return np.maximum(((p - a) ** 2 - (n - a) ** 2).sum(axis=1) + margin, 0.0)


def triplet_loss_bwd(a, p, n, loss, gloss):
    g = np.where(loss > 0, 2.0 * gloss, 0.0)[:, None]
    return g * (n - p), g * (p - a), -g * (n - a)


def cosine_dist(anc, pos, neg):
    """ref: train_efm.py:26-34 — per-row dot / (|a||p|), dot / (|a||n|)."""
    na = np.sqrt((anc * anc).sum(1))
    return (anc * pos).sum(1) / (na * np.sqrt((pos * pos).sum(1))), (anc * neg).sum(1) / (na * np.sqrt((neg * neg).sum(1)))


def gram_cosine(e):
    n = np.sqrt((e * e).sum(1))
    return (e @ e.T) / (n[:, None] * n[None, :])


def pick_negatives(labels_anchor, labels_pool, draws):
    """The reference's rejection sampling (ref: train_efm.py:234-239) with the random stream made explicit:
    `draws` is an iterator of integers in [0, len(labels_pool)); for anchor i take draws until the label differs."""
    out = []
    it = iter(draws)
    for la in labels_anchor:
        j = next(it)
        while int(labels_pool[j]) == int(la):
            j = next(it)
        out.append(j)
    return np.asarray(out, dtype=np.int32)


def mine_semihard(g, labels, anchor_idx, pos_idx):
    """Build-defined (no reference): d = 1 - cos.  argmin_{label!=, d_an > d_ap} d_an, else argmax_{label!=} d_an;
    lowest index wins ties; -1 if no other identity."""
    out = np.full(len(anchor_idx), -1, dtype=np.int32)
    for t, (a, p) in enumerate(zip(anchor_idx, pos_idx)):
        d = 1.0 - g[a]
        cand = np.nonzero(labels != labels[a])[0]
        if cand.size == 0:
            continue
        sh = cand[d[cand] > d[p]]
        if sh.size:
            out[t] = sh[np.argmin(d[sh])]
        else:
            out[t] = cand[np.argmax(d[cand])]
    return out


def mining_indices(labels):
    """Batch-all layout helper: positive of row i = the next row of the same identity (cyclic) -> a permutation;
    rows whose identity appears once are their own positive (d_ap = 0, as define_pos allows, train_efm.py:42-43)."""
    labels = np.asarray(labels)
    pos = np.arange(len(labels), dtype=np.int32)
    for lab in np.unique(labels):
        idx = np.nonzero(labels == lab)[0]
        pos[idx] = np.roll(idx, -1)
    inv = np.empty_like(pos)
    inv[pos] = np.arange(len(labels), dtype=np.int32)
    return pos, inv


def triplet_indexed(e, pos, neg, margin):
    ok = neg >= 0
    n = e[np.where(ok, neg, 0)]
    return np.where(ok, triplet_loss(e, e[pos], n, margin), 0.0)


def triplet_indexed_bwd(e, pos, neg, loss, gloss):
    """d(sum_i gloss_i loss_i)/de with the negatives detached."""
    g = np.where(loss > 0, 2.0 * gloss, 0.0)[:, None]
    de = g * (e[np.where(neg >= 0, neg, 0)] - e[pos])     # as anchor
    np.add.at(de, pos, g * (e[pos] - e))                  # as positive of anchor i
    return de


def softmax_cross_entropy(logits, labels):
    """SoftmaxCrossEntropyLoss, sparse labels, per-sample.  [MX-assumed (8)]  ref: train_efm.py:211,242."""
    z = logits - logits.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1))
    return lse - z[np.arange(len(labels)), labels.astype(int)]


def softmax_cross_entropy_bwd(logits, labels, gloss):
    z = logits - logits.max(axis=1, keepdims=True)
    p = np.exp(z)
    p /= p.sum(axis=1, keepdims=True)
    p[np.arange(len(labels)), labels.astype(int)] -= 1.0
    return p * gloss[:, None]


# ----------------------------------------------------------------------------------------------
# Optimisers / schedule  [MX-assumed (11)]
# ----------------------------------------------------------------------------------------------
def sgd_step(w, g, lr, wd, rescale):
    """mx.optimizer.SGD, momentum 0: w -= lr * (rescale*g + wd*w).  ref: pre-trained_efm_v3.py:185,212."""
    return w - lr * (rescale * g + wd * w)


def adam_step(w, g, m, v, t, lr, wd, rescale, beta1=0.9, beta2=0.999, eps=1e-8):
    """mx.optimizer.Adam: g' = rescale*g + wd*w; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps).
    ref: train_efm.py:213; mutli_gpu_v3.py:159."""
    gr = rescale * g + wd * w
    m = beta1 * m + (1 - beta1) * gr
    v = beta2 * v + (1 - beta2) * gr * gr
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    return w - lr_t * m / (np.sqrt(v) + eps), m, v


def factor_scheduler(base_lr, num_update, step, factor, stop_factor_lr=5e-15):
    """mx.lr_scheduler.FactorScheduler: lr *= factor every `step` updates, floored at stop_factor_lr.
    ref: train_efm.py:212."""
    lr, count = base_lr, 0
    while num_update > count + step:
        count += step
        lr *= factor
        if lr < stop_factor_lr:
            return stop_factor_lr
    return lr


# ----------------------------------------------------------------------------------------------
# The EFM-29 network (Symbol variant, authoritative: efm_symbol.py:22-110)
# ----------------------------------------------------------------------------------------------
def efm29_layers(in_channels=3):
    """Parameter table [(name, cout, cin, kh, kw, pad)] in forward order, following
    group()/res_block() of efm_symbol.py:22-92 with the five calls at :84-92."""
    layers = []
    c = in_channels
    groups = [(0, 99, 5, 2, "1", 0), (99, 198, 3, 1, "2", 1), (198, 387, 3, 1, "3", 2), (387, 261, 3, 1, "4", 3),
              (261, 261, 3, 1, "5", 4)]
    for num_r, num, k, pad, layer, tar in groups:
        if num_r > 0:
            num_r1 = int(num_r * (2.0 / 3.0))
            for x in range(tar):
                lname = layer if x == 0 else layer + str(x)
                layers.append(("conv%s_res" % lname, num_r, 2 * c // 3, 3, 3, 1))
                layers.append(("conv%s_res_r" % lname, num_r1, 2 * num_r // 3, 3, 3, 1))
            layers.append(("conv%s_r" % layer, num_r, c, 1, 1, 0))
            c = 2 * num_r // 3
        layers.append(("conv%s" % layer, num, c, k, k, pad))
        c = 2 * num // 3
    return layers


def efm29_param_shapes(in_channels=3, image=112, fc_hidden=513):
    shapes = {}
    for name, co, ci, kh, kw, _ in efm29_layers(in_channels):
        shapes[name + "_weight"] = (co, ci, kh, kw)
        shapes[name + "_bias"] = (co,)
    s = image
    for _ in range(5):
        s //= 2
    shapes["fc1_weight"] = (fc_hidden, 174 * s * s)
    shapes["fc1_bias"] = (fc_hidden,)
    return shapes


def xavier_uniform_scale(shape):
    """Gluon init.Xavier() = uniform(+-sqrt(3 / ((fan_in + fan_out)/2))), fan_in = shape[1]*prod(shape[2:]),
    fan_out = shape[0]*prod(shape[2:]); biases are zero.  [MX-assumed (10)]  ref: train_efm.py:208."""
    hw = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[1] * hw, shape[0] * hw
    return math.sqrt(3.0 / ((fan_in + fan_out) / 2.0))


def init_params(shapes, seed=42):
    """Deterministic Gluon-Xavier-uniform weights from splitmix64; per-parameter seed = seed + index."""
    params = {}
    for i, (name, shp) in enumerate(shapes.items()):
        if name.endswith("_bias"):
            params[name] = np.zeros(shp)
        else:
            params[name] = uniform_pm(shp, seed + 1000003 * (i + 1), xavier_uniform_scale(shp))
    return params


class Tape:
    def __init__(self):
        self.ops = []

    def push(self, fn):
        self.ops.append(fn)


def efm29_forward(params, x, tape=None, in_channels=3, routing=None):
    """x: (B, C, H, W) -> 342-d post-fc1 MFM feature ('concat29_output', ref: final_efm.py:208).
    With `tape`, records closures so that efm29_backward can run.

    `routing` (optional): {node name -> tensor} whose arg-max / arg-min decisions the BACKWARD pass follows instead of
    this run's own forward values (keys: 'efm<L>_res_in', 'efm<L>_res', 'efm<L>_r', 'efm<L>', 'pool<L>', 'concat29' =
    the input of that MFM / pooling node).  max/min/pool gradients are piecewise constant in the forward values, so
    an fp32 implementation whose forward differs in the last bit can legitimately take a different route; handing its
    forward values in here makes the comparison of the backward arithmetic exact (see tests/test_e2e_gpu.py)."""
    acts = {}
    grads = {}

    def R(key, val):
        return np.asarray(routing[key]).reshape(val.shape) if routing is not None and key in routing else val

    def conv(name, inp, pad, residual=None):
        w, b = params[name + "_weight"], params[name + "_bias"]
        y = conv2d(inp, w, b, (pad, pad))
        if residual is not None:
            y = y + residual
        if tape is not None:
            def bwd(dy, inp=inp, w=w, name=name, pad=pad):
                dx, dw, db = conv2d_bwd(inp, w, dy, (pad, pad))
                grads[name + "_weight"] = grads.get(name + "_weight", 0) + dw
                grads[name + "_bias"] = grads.get(name + "_bias", 0) + db
                return dx
            tape.push(("conv", name, bwd))
        return y

    def res_block(data, num_r, lname):
        # ref: efm_symbol.py:22-44
        e = mfm3(data)
        c1 = conv("conv%s_res" % lname, e, 1)
        e2 = mfm3(c1)
        out = conv("conv%s_res_r" % lname, e2, 1, residual=data)
        if tape is not None:
            tape.ops[-2:] = [("res_block", lname, (tape.ops[-2][2], tape.ops[-1][2], R("efm%s_res_in" % lname, data),
                                                   R("efm%s_res" % lname, c1)))]
        return out

    cur = x
    groups = [(0, 99, 5, 2, "1", 0), (99, 198, 3, 1, "2", 1), (198, 387, 3, 1, "3", 2), (387, 261, 3, 1, "4", 3),
              (261, 261, 3, 1, "5", 4)]
    for num_r, num, k, pad, layer, tar in groups:
        if num_r > 0:
            for xx in range(tar):
                cur = res_block(cur, num_r, layer if xx == 0 else layer + str(xx))
            cr = conv("conv%s_r" % layer, cur, 0)
            cur = mfm3(cr)
            if tape is not None:
                tape.push(("mfm", layer + "_r", R("efm%s_r" % layer, cr)))
        cv = conv("conv%s" % layer, cur, pad)
        mf = mfm3(cv)
        cur = maxpool2(mf)
        if tape is not None:
            tape.push(("mfm_pool", layer, (R("efm%s" % layer, cv), R("pool%s" % layer, mf))))
        acts["pool" + layer] = cur
    flat = cur
    fc1 = fully_connected(flat, params["fc1_weight"], params["fc1_bias"])
    feat = mfm3(fc1)
    if tape is not None:
        tape.push(("fc1", "fc1", (flat, R("concat29", fc1))))
        tape.grads = grads
    acts["fc1"] = fc1
    acts["feat"] = feat
    return feat, acts


def efm29_backward(params, tape, dfeat):
    """Reverse sweep of the tape recorded by efm29_forward; returns (dx, grads dict)."""
    grads = tape.grads
    g = dfeat
    for kind, name, payload in reversed(tape.ops):
        if kind == "fc1":
            flat, fc1 = payload
            g = mfm3_bwd(fc1, g, ORDER_RES)  # ref: efm_symbol.py:97-100 uses maximum(slice[2], max1)
            g, dw, db = fully_connected_bwd(flat, params["fc1_weight"], g)
            grads["fc1_weight"] = dw
            grads["fc1_bias"] = db
        elif kind == "mfm_pool":
            cv, mf = payload
            g = maxpool2_bwd(mf, g)
            g = mfm3_bwd(cv, g, ORDER_GROUP)  # ref: efm_symbol.py:70-73
        elif kind == "mfm":
            g = mfm3_bwd(payload, g, ORDER_RES)  # ref: efm_symbol.py:56-59
        elif kind == "conv":
            g = payload(g)
        elif kind == "res_block":
            bwd1, bwd2, data, c1 = payload
            d_e2 = bwd2(g)
            d_c1 = mfm3_bwd(c1, d_e2, ORDER_RES)  # ref: efm_symbol.py:35-38
            d_e = bwd1(d_c1)
            g = g + mfm3_bwd(data, d_e, ORDER_RES)  # ref: efm_symbol.py:26-29
        else:
            raise AssertionError(kind)
    return g, grads


def head_forward(w_head, feat, normalize="row"):
    """Per-row L2 normalisation of the 342-d feature then Dense(128, use_bias=False)
    (ref: final_efm.py:240-243 + pre-trained_efm_v3.py:180-181)."""
    if normalize == "row":
        y, n = l2norm_row(feat)
    else:
        y, n = feat, None
    return y @ w_head.T, (y, n)


def train_step_loss(params, w_head, x, neg_idx, margin, in_channels=3, demb=None, routing=None):
    """One reference-layout step: batch = [B/2 anchors ; B/2 positives], negatives = detached rows of the anchor
    half picked by `neg_idx` (ref: train_efm.py:232-241), loss vector (B/2,), head on row-normalised features.
    Returns (loss, emb, feat, grads, g_head).  With `demb` given, that upstream gradient replaces the loss's own
    (a well-conditioned probe of the backward pass: at random init all embeddings nearly coincide, so the loss
    gradient 2(n-p) is a difference of nearly equal fp32 numbers)."""
    tape = Tape()
    feat, acts = efm29_forward(params, x, tape, in_channels, routing)
    emb, (yn, nrm) = head_forward(w_head, feat)
    h = x.shape[0] // 2
    a, p = emb[:h], emb[h:]
    n = emb[neg_idx]
    loss = triplet_loss(a, p, n, margin)
    if demb is None:
        da, dp, _ = triplet_loss_bwd(a, p, n, loss, np.ones_like(loss))  # vector backward = ones head-grad [MX-assumed (9)]
        demb = np.concatenate([da, dp], axis=0)
    g_head = demb.T @ yn
    dyn = demb @ w_head
    dfeat = l2norm_row_bwd(yn, nrm, dyn)
    _, grads = efm29_backward(params, tape, dfeat)
    return loss, emb, feat, grads, g_head


# ----------------------------------------------------------------------------------------------
# LFW verification protocol (restates feature_extraction/facenet_version/facenet.py:412-471)
# ----------------------------------------------------------------------------------------------
def lfw_distance(e1, e2, metric=0):
    if metric == 0:
        return ((e1 - e2) ** 2).sum(1)
    dot = (e1 * e2).sum(1)
    nrm = np.linalg.norm(e1, axis=1) * np.linalg.norm(e2, axis=1)
    return np.arccos(dot / nrm) / math.pi


def lfw_accuracy(threshold, dist, issame):
    pred = dist < threshold
    tp = np.sum(pred & issame)
    fp = np.sum(pred & ~issame)
    tn = np.sum(~pred & ~issame)
    fn = np.sum(~pred & issame)
    tpr = 0 if tp + fn == 0 else float(tp) / float(tp + fn)
    fpr = 0 if fp + tn == 0 else float(fp) / float(fp + tn)
    return tpr, fpr, float(tp + tn) / dist.size


def lfw_roc(thresholds, e1, e2, issame, nrof_folds=10, metric=0, subtract_mean=False):
    """k-fold (contiguous, unshuffled) best-threshold accuracy — facenet.calculate_roc."""
    n = min(len(issame), e1.shape[0])
    folds = np.array_split(np.arange(n), nrof_folds)
    tprs = np.zeros((nrof_folds, len(thresholds)))
    fprs = np.zeros((nrof_folds, len(thresholds)))
    acc = np.zeros(nrof_folds)
    for f, test in enumerate(folds):
        train = np.concatenate([folds[k] for k in range(nrof_folds) if k != f])
        mean = np.mean(np.concatenate([e1[train], e2[train]]), axis=0) if subtract_mean else 0.0
        dist = lfw_distance(e1 - mean, e2 - mean, metric)
        acc_train = np.array([lfw_accuracy(t, dist[train], issame[train])[2] for t in thresholds])
        best = int(np.argmax(acc_train))
        for ti, t in enumerate(thresholds):
            tprs[f, ti], fprs[f, ti], _ = lfw_accuracy(t, dist[test], issame[test])
        acc[f] = lfw_accuracy(thresholds[best], dist[test], issame[test])[2]
    return tprs.mean(0), fprs.mean(0), acc
