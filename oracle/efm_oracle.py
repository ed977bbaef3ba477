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


# ----------------------------------------------------------------------------------------------
# The Gluon variant: LightCNN_29 (lightcnn.py:6-133) and the train_efm.py step (train_efm.py:229-245).
# This is the network entry point 1 actually trains: the two convolutions of a res_block are created ONCE
# (lightcnn.py:47-48) and re-applied num_blocks times (:52-69), fc1 is Dense(1026) -> EFM -> 684-d (:111,123-128),
# fc1_out = BatchNorm(feature) (:113-114,130), out = Dense(classes)(Dropout(.7)(feature)) (:116-118,131).
# Every EFM here is `maximum(maximum(s0,s1), s2)` = ORDER_GROUP (:23-26,33-36,54-57,62-65,124-127).
# Parameter names are this build's (g<k>_conv0/1, g<k>_res_conv0/1, fc1, batchnorm0_*, dense1_*); GLUON_STRUCT_NAMES maps them
# to the structural names Block.save_parameters writes ("conv_net.2.conv_op_1.weight", ...) [MX-assumed (14)].
# ----------------------------------------------------------------------------------------------
LIGHTCNN29_BLOCKS = [1, 2, 3, 4]                      # lightcnn.py:77
LIGHTCNN29_GROUPS = [(99, 198), (198, 387), (387, 261), (261, 261)]  # (num_filter, num_filter1) of efm(type 1), lightcnn.py:88,94,100,106


def lightcnn29_layers(in_channels=1):
    """[(name, cout, cin, k, pad)] — each SHARED convolution listed once."""
    layers = [("g1_conv1", 99, in_channels, 5, 2)]    # efm(0, 99, 5x5, pad 2, type 0), lightcnn.py:82
    c = 66
    for gi, (nf, nf1) in enumerate(LIGHTCNN29_GROUPS):
        g = gi + 2
        layers.append(("g%d_res_conv0" % g, nf, 2 * c // 3, 3, 1))           # conv_op_1: channels=num_filter, lightcnn.py:47
        layers.append(("g%d_res_conv1" % g, int(nf * (2. / 3.)), 2 * nf // 3, 3, 1))  # conv_op_2: int(num_filter*2/3), :45,48
        layers.append(("g%d_conv0" % g, nf, c, 1, 0))                        # efm.conv_op_1 1x1, lightcnn.py:14
        layers.append(("g%d_conv1" % g, nf1, 2 * nf // 3, 3, 1))             # efm.conv_op_2 kxk, lightcnn.py:15
        c = 2 * nf1 // 3
    return layers


def lightcnn29_param_shapes(in_channels=1, image=128, classes=8398, fc_units=1026):
    shapes = {}
    for name, co, ci, k, _ in lightcnn29_layers(in_channels):
        shapes[name + "_weight"] = (co, ci, k, k)
        shapes[name + "_bias"] = (co,)
    s = image
    for _ in range(5):
        s //= 2
    shapes["fc1_weight"] = (fc_units, 174 * s * s)
    shapes["fc1_bias"] = (fc_units,)
    d = 2 * fc_units // 3
    shapes["batchnorm0_gamma"] = (d,)
    shapes["batchnorm0_beta"] = (d,)
    shapes["dense1_weight"] = (classes, d)
    shapes["dense1_bias"] = (classes,)
    return shapes


def gluon_struct_names():
    """build name -> the key `net.save_parameters` writes for LightCNN_29 (Block._collect_params_with_prefix: attribute path,
    Sequential children by index; lightcnn.py:79-118 gives the indices) [MX-assumed (14)]."""
    m = {"g1_conv1": "conv_net.0.conv_op_2"}
    for gi in range(4):
        g, base = gi + 2, 2 + 3 * gi
        m["g%d_res_conv0" % g] = "conv_net.%d.conv_op_1" % base
        m["g%d_res_conv1" % g] = "conv_net.%d.conv_op_2" % base
        m["g%d_conv0" % g] = "conv_net.%d.conv_op_1" % (base + 1)
        m["g%d_conv1" % g] = "conv_net.%d.conv_op_2" % (base + 1)
    m["fc1"] = "conv_net.15"
    out = {}
    for k, v in m.items():
        out[k + "_weight"] = v + ".weight"
        out[k + "_bias"] = v + ".bias"
    for k in ("gamma", "beta", "running_mean", "running_var"):
        out["batchnorm0_" + k] = "fc1.0." + k
    out["dense1_weight"], out["dense1_bias"] = "fc2.1.weight", "fc2.1.bias"
    return out


def init_lightcnn29_params(shapes, seed=42):
    p = init_params({k: v for k, v in shapes.items() if not k.startswith("batchnorm0_")}, seed)
    p["batchnorm0_gamma"] = np.ones(shapes["batchnorm0_gamma"])   # Gluon BatchNorm: gamma ones, beta zeros
    p["batchnorm0_beta"] = np.zeros(shapes["batchnorm0_beta"])
    return p


def batchnorm_train(x, gamma, beta, eps=1e-5):
    """Gluon nn.BatchNorm() in training mode on a (N, C) matrix: batch mean, BIASED batch variance, eps 1e-5
    [MX-assumed (13)]; returns (y, (xhat, inv_std, mean, var)).  ref: lightcnn.py:113-114,130."""
    mean = x.mean(axis=0)
    var = ((x - mean) ** 2).mean(axis=0)
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * inv
    return gamma * xhat + beta, (xhat, inv, mean, var)


def batchnorm_train_bwd(cache, gamma, dy):
    xhat, inv, _, _ = cache
    n = dy.shape[0]
    dxhat = dy * gamma
    dx = inv / n * (n * dxhat - dxhat.sum(axis=0) - xhat * (dxhat * xhat).sum(axis=0))
    return dx, (dy * xhat).sum(axis=0), dy.sum(axis=0)


def batchnorm_running_update(running_mean, running_var, mean, var, momentum=0.9):
    """MXNet: moving = moving*momentum + batch*(1-momentum), with the BIASED batch variance (torch tracks the unbiased one)."""
    return running_mean * momentum + mean * (1 - momentum), running_var * momentum + var * (1 - momentum)


def batchnorm_infer(x, gamma, beta, running_mean, running_var, eps=1e-5):
    return gamma * (x - running_mean) / np.sqrt(running_var + eps) + beta


def lightcnn29_forward(params, x, tape=None, routing=None):
    """x (N, C, H, W) -> 684-d EFM feature (the input of both heads).  With `tape`: records the backward closures
    (`tape.ops`, `tape.grads`); gradients of the shared convolutions ACCUMULATE over their uses.
    `routing` as in efm29_forward; keys: 'g1_efm', 'g1_pool', 'g<k>_res<i>_efm_in', 'g<k>_res<i>_efm', 'g<k>_efm0', 'g<k>_efm1',
    'g<k>_pool', 'efm_fc1' = the INPUT of that EFM / pooling node."""
    grads = {}

    def R(key, val):
        return np.asarray(routing[key]).reshape(val.shape) if routing is not None and key in routing else val

    def push(fn):
        if tape is not None:
            tape.push(fn)

    def conv(name, inp, pad):
        w, b = params[name + "_weight"], params[name + "_bias"]
        y = conv2d(inp, w, b, (pad, pad))

        def bwd(dy, inp=inp, w=w):
            dx, dw, db = conv2d_bwd(inp, w, dy, (pad, pad))
            grads[name + "_weight"] = grads.get(name + "_weight", 0) + dw
            grads[name + "_bias"] = grads.get(name + "_bias", 0) + db
            return dx
        return y, bwd

    def efm_op(key, inp):
        r = R(key, inp)
        return mfm3(inp), (lambda dy: mfm3_bwd(r, dy, ORDER_GROUP))

    def pool_op(key, inp):
        r = R(key, inp)
        return maxpool2(inp), (lambda dy: maxpool2_bwd(r, dy))

    def chain(fns):
        def bwd(g):
            for f in reversed(fns):
                g = f(g)
            return g
        return bwd

    # group 1: efm(type 0) + pool  (lightcnn.py:82-83)
    c, b1 = conv("g1_conv1", x, 2)
    e, b2 = efm_op("g1_efm", c)
    cur, b3 = pool_op("g1_pool", e)
    push(chain([b1, b2, b3]))
    for gi, nb in enumerate(LIGHTCNN29_BLOCKS):
        g = gi + 2
        for i in range(nb):                                   # res_block.hybrid_forward, lightcnn.py:50-71
            e, f1 = efm_op("g%d_res%d_efm_in" % (g, i), cur)
            c1, f2 = conv("g%d_res_conv0" % g, e, 1)
            e2, f3 = efm_op("g%d_res%d_efm" % (g, i), c1)
            c2, f4 = conv("g%d_res_conv1" % g, e2, 1)
            cur = c2 + cur
            inner = chain([f1, f2, f3, f4])
            push(lambda gq, inner=inner: gq + inner(gq))
        c0, f1 = conv("g%d_conv0" % g, cur, 0)                # efm(type 1), lightcnn.py:20-30
        e0, f2 = efm_op("g%d_efm0" % g, c0)
        c1, f3 = conv("g%d_conv1" % g, e0, 1)
        e1, f4 = efm_op("g%d_efm1" % g, c1)
        cur, f5 = pool_op("g%d_pool" % g, e1)
        push(chain([f1, f2, f3, f4, f5]))
    flat = cur
    fc1 = fully_connected(flat, params["fc1_weight"], params["fc1_bias"])
    feat, fe = efm_op("efm_fc1", fc1)

    def fc_bwd(gq):
        gq = fe(gq)
        dx, dw, db = fully_connected_bwd(flat, params["fc1_weight"], gq)
        grads["fc1_weight"], grads["fc1_bias"] = dw, db
        return dx
    push(fc_bwd)
    if tape is not None:
        tape.grads = grads
    return feat


def lightcnn29_heads(params, feat, dropout_mask=None, dropout_p=0.7):
    """(out, fc1_out, cache) in training mode.  `dropout_mask` (same shape as feat, 0/1) makes Dropout(.7) replayable;
    None = dropout off.  [MX-assumed (12)]: kept values are scaled by 1/(1-p)."""
    fc1_out, bn = batchnorm_train(feat, params["batchnorm0_gamma"], params["batchnorm0_beta"])
    d = feat if dropout_mask is None else feat * dropout_mask / (1.0 - dropout_p)
    out = d @ params["dense1_weight"].T + params["dense1_bias"]
    return out, fc1_out, (bn, d)


def train_efm_step(params, x, labels, neg_idx, margin=0.2, alpha=0.1, dropout_mask=None, dropout_p=0.7, routing=None):
    """One training step of train_efm.py:229-245 (with `nrom` read as `norm`): x = [B anchors ; B positives],
    `labels` (2B,), `neg_idx` (B,) rows of the anchor half (detached).  Returns a dict with out (2B, classes), fc1_out (2B, 684),
    TL / id / loss vectors (B,), `grads` (name -> d sum(loss) / d param, the ones-head-gradient backward [MX-assumed (9)]),
    and the BatchNorm batch statistics."""
    tape = Tape()
    feat = lightcnn29_forward(params, x, tape, routing)
    out, fc, (bn, dropped) = lightcnn29_heads(params, feat, dropout_mask, dropout_p)
    b = x.shape[0] // 2
    anc, pos = fc[:b], fc[b:2 * b]
    neg = fc[neg_idx]                                            # copied through NumPy in the reference: no gradient
    (ya, na), (yp, npos), (yn, _) = l2norm_frob(anc), l2norm_frob(pos), l2norm_frob(neg)
    tl = triplet_loss(ya, yp, yn, margin)
    idl = softmax_cross_entropy(out[:b], labels[:b])
    loss = idl + alpha * tl
    # backward of sum(loss)
    da, dp, _ = triplet_loss_bwd(ya, yp, yn, tl, alpha * np.ones(b))
    dfc = np.concatenate([l2norm_frob_bwd(ya, na, da), l2norm_frob_bwd(yp, npos, dp)], axis=0)
    dout = np.zeros_like(out)
    dout[:b] = softmax_cross_entropy_bwd(out[:b], labels[:b], np.ones(b))
    grads = tape.grads
    dfeat_bn, grads["batchnorm0_gamma"], grads["batchnorm0_beta"] = batchnorm_train_bwd(bn, params["batchnorm0_gamma"], dfc)
    grads["dense1_weight"] = dout.T @ dropped
    grads["dense1_bias"] = dout.sum(0)
    dd = dout @ params["dense1_weight"]
    if dropout_mask is not None:
        dd = dd * dropout_mask / (1.0 - dropout_p)
    g = dfeat_bn + dd
    for fn in reversed(tape.ops):
        g = fn(g)
    return {"out": out, "fc1_out": fc, "feat": feat, "tl": tl, "id": idl, "loss": loss, "grads": grads,
            "bn_mean": bn[2], "bn_var": bn[3]}
