"""GPU parity of every HIP kernel behind include/efm_hip.h against the CPU oracle (oracle/efm_oracle.py).

Tolerance: the north star asks for 1e-3 relative fp32 end to end; single kernels are held to 2e-4 of the
largest reference magnitude (fp32 MFMA = k-ordered fmaf chain, error ~1e-7 * sum|a*b|).
"""
import os

import numpy as np
import pytest
import torch

from oracle import efm_oracle as O
from tests.util import dev, from_nhwc, rand, rel_err, to_nhwc

pytestmark = pytest.mark.gpu

TOL = 2e-4


@pytest.fixture(scope="module")
def ops():
    from improving_face_recognition_performance_using_triplet_loss_amd import ops as _ops
    return _ops


def test_layout_roundtrip(ops):
    x = rand((3, 5, 7, 6), 0)
    y = to_nhwc(x)
    assert tuple(y.shape) == (3, 7, 6, 8)
    ref = np.zeros((3, 7, 6, 8))
    ref[..., :5] = x.transpose(0, 2, 3, 1)
    assert rel_err(y.cpu().numpy(), ref) < 1e-7
    assert rel_err(from_nhwc(y, 5), x.astype(np.float32)) < 1e-7


CONV_CASES = [
    # batch, h, w, cin, cout, k, pad
    (2, 12, 10, 3, 99, 5, 2),     # conv1-like (cin_p = 4, four taps per K step)
    (2, 9, 11, 44, 99, 3, 1),     # conv2_res
    (3, 8, 8, 66, 66, 3, 1),      # conv2_res_r (cin_p 68: K pieces straddle taps)
    (2, 7, 7, 66, 99, 1, 0),      # 1x1
    (2, 6, 5, 132, 387, 3, 1),    # two channel blocks (25 tiles)
    (1, 5, 5, 258, 258, 3, 1),    # 17 tiles -> 9+8
    (5, 3, 3, 174, 513, 3, 0),    # fc1 as a 3x3 'valid' conv, output 1x1
    (7, 1, 1, 342, 128, 1, 0),    # embedding head Dense(128)
    (2, 14, 14, 172, 387, 3, 1),
]


@pytest.mark.parametrize("mt", [1, 2])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(ops, case, mt):
    b, h, w, cin, cout, k, pad = case
    os.environ["EFM_CONV_MT"] = str(mt)
    try:
        x = rand((b, cin, h, w), 1)
        wt = rand((cout, cin, k, k), 2, 0.2)
        bias = rand((cout,), 3)
        d = ops.conv_desc(b, h, w, cin, cout, k, k, pad, pad)
        xd = to_nhwc(x)
        wp = ops.conv_pack_weights(d, dev(wt))
        assert rel_err(ops.conv_unpack_weights(d, wp).cpu().numpy(), wt.astype(np.float32)) < 1e-7
        bp = torch.zeros(d.n_pad16, device="cuda")
        bp[:cout] = dev(bias)
        ref = O.conv2d(x, wt, bias, (pad, pad))
        y = ops.conv_fwd(d, xd, wp, bp)
        assert rel_err(from_nhwc(y, cout), ref) < TOL
        # pad channels stay zero
        assert float(y[..., cout:].abs().max()) == 0.0 if d.cout_p > cout else True
        # residual epilogue
        res = rand(ref.shape, 4)
        y2 = ops.conv_fwd(d, xd, wp, bp, residual=to_nhwc(res))
        assert rel_err(from_nhwc(y2, cout), ref + res) < TOL
        # backward
        dy = rand(ref.shape, 5)
        dx_ref, dw_ref, db_ref = O.conv2d_bwd(x, wt, dy, (pad, pad))
        dyd = to_nhwc(dy)
        wd = ops.conv_make_dgrad_weights(d, wp)
        dx = ops.conv_bwd_data(d, dyd, wd)
        assert rel_err(from_nhwc(dx, cin), dx_ref) < TOL
        if d.cin_p > cin:
            assert float(dx[..., cin:].abs().max()) == 0.0
        add = rand(x.shape, 6)
        dx2 = ops.conv_bwd_data(d, dyd, wd, add=to_nhwc(add))
        assert rel_err(from_nhwc(dx2, cin), dx_ref + add) < TOL
        dw, db = ops.conv_bwd_weight(d, xd, dyd)
        assert rel_err(ops.conv_unpack_weights(d, dw).cpu().numpy(), dw_ref) < TOL
        assert rel_err(db[:cout].cpu().numpy(), db_ref) < TOL
        # packed-gradient pads are exactly zero (keeps the zero-pad invariant of the packed weights under SGD)
        dwm = dw.clone()
        ops.conv_pack_weights_into(d, ops.conv_unpack_weights(d, dw), dwm)
        assert torch.equal(dwm, dw)
        assert float(db[cout:].abs().max()) == 0.0 if d.n_pad16 > cout else True
    finally:
        os.environ.pop("EFM_CONV_MT", None)


def test_conv_wgrad_deterministic(ops):
    b, h, w, cin, cout = 4, 28, 28, 88, 198
    d = ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1)
    xd = to_nhwc(rand((b, cin, h, w), 1))
    dyd = to_nhwc(rand((b, cout, h, w), 2))
    dw1, db1 = ops.conv_bwd_weight(d, xd, dyd)
    dw1, db1 = dw1.clone(), db1.clone()
    dw2, db2 = ops.conv_bwd_weight(d, xd, dyd)
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


@pytest.mark.parametrize("c,rows", [(99, (2, 5, 4)), (66, (3, 4, 4)), (513, (6,)), (387, (1, 3, 3))])
def test_mfm3(ops, c, rows):
    shape = (rows[0], c) + tuple(rows[1:]) if len(rows) == 3 else (rows[0], c, 1, 1)
    x = rand(shape, 7)
    # force exact ties so that the tie rule is exercised
    x[0, : c // 3] = x[0, c // 3: 2 * c // 3]
    x[-1, 2 * (c // 3):] = x[-1, : c // 3]
    xd = to_nhwc(x)
    y = ops.mfm_fwd(xd, c, 3)
    co = 2 * c // 3
    assert rel_err(from_nhwc(y, co), O.mfm3(x)) < 1e-7
    if y.shape[-1] > co:
        assert float(y[..., co:].abs().max()) == 0.0
    dy = rand(O.mfm3(x).shape, 8)
    add = rand(x.shape, 9)
    for order in (O.ORDER_GROUP, O.ORDER_RES):
        dx = ops.mfm_bwd(xd, to_nhwc(dy), c, 3, order)
        assert rel_err(from_nhwc(dx, c), O.mfm3_bwd(x, dy, order)) < 1e-7
        dx2 = ops.mfm_bwd(xd, to_nhwc(dy), c, 3, order, add=to_nhwc(add))
        assert rel_err(from_nhwc(dx2, c), O.mfm3_bwd(x, dy, order) + add) < 1e-6


def test_mfm2(ops):
    x = rand((2, 96, 5, 5), 10)
    x[0, :48] = x[0, 48:]
    xd = to_nhwc(x)
    y = ops.mfm_fwd(xd, 96, 2)
    assert rel_err(from_nhwc(y, 48), O.mfm2(x)) < 1e-7
    dy = rand((2, 48, 5, 5), 11)
    dx = ops.mfm_bwd(xd, to_nhwc(dy), 96, 2)
    assert rel_err(from_nhwc(dx, 96), O.mfm2_bwd(x, dy)) < 1e-7


@pytest.mark.parametrize("h,w,c", [(8, 8, 66), (7, 7, 174), (5, 6, 44), (112, 112, 66)])
def test_maxpool2(ops, h, w, c):
    b = 2
    x = rand((b, c, h, w), 12)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = 5.0  # tie inside a window -> first wins
    xd = to_nhwc(x)
    y = ops.maxpool2_fwd(xd, c)
    ref = O.maxpool2(x)
    assert rel_err(from_nhwc(y, c), ref) < 1e-7
    dy = rand(ref.shape, 13)
    dx = torch.full_like(xd, float("nan"))  # poison: every element must be written
    ops.maxpool2_bwd(xd, to_nhwc(dy), c, out=dx)
    assert rel_err(from_nhwc(dx, c), O.maxpool2_bwd(x, dy)) < 1e-7


@pytest.mark.parametrize("rows,d", [(5, 128), (128, 342), (3, 684), (16, 100)])
def test_l2norm(ops, rows, d):
    x = rand((rows, d), 14)
    dy = rand((rows, d), 15)
    y, n = ops.l2norm_fwd(dev(x), 0)
    yr, nr = O.l2norm_row(x)
    assert rel_err(y.cpu().numpy(), yr) < 1e-6 and rel_err(n.cpu().numpy(), nr) < 1e-6
    dx = ops.l2norm_bwd(y, n, dev(dy), 0)
    assert rel_err(dx.cpu().numpy(), O.l2norm_row_bwd(yr, nr, dy)) < 1e-5
    y, n = ops.l2norm_fwd(dev(x), 1)
    yf, nf = O.l2norm_frob(x)
    assert rel_err(y.cpu().numpy(), yf) < 1e-6 and abs(float(n[0]) - nf) / nf < 1e-6
    dx = ops.l2norm_bwd(y, n, dev(dy), 1)
    assert rel_err(dx.cpu().numpy(), O.l2norm_frob_bwd(yf, nf, dy)) < 1e-5


@pytest.mark.parametrize("rows,d,margin", [(7, 128, 0.2), (128, 342, 0.5), (33, 2, 0.2)])
def test_triplet_and_cosine(ops, rows, d, margin):
    a, p, n = rand((rows, d), 16, 0.2), rand((rows, d), 17, 0.2), rand((rows, d), 18, 0.2)
    n[0] = a[0] * 50  # a row whose loss is clamped to zero -> gradient gate
    loss = ops.triplet_fwd(dev(a), dev(p), dev(n), margin)
    lr = O.triplet_loss(a, p, n, margin)
    assert (lr == 0).any() and (lr > 0).any()
    assert rel_err(loss.cpu().numpy(), lr) < 1e-5
    g = rand((rows,), 19)
    da, dp, dn = ops.triplet_bwd(dev(a), dev(p), dev(n), loss, dev(g), need_dn=True)
    ra, rp, rn = O.triplet_loss_bwd(a, p, n, lr, g)
    assert rel_err(da.cpu().numpy(), ra) < 1e-5 and rel_err(dp.cpu().numpy(), rp) < 1e-5
    assert rel_err(dn.cpu().numpy(), rn) < 1e-5
    s_ap, s_an = ops.cosine_pairs(dev(a), dev(p), dev(n))
    r_ap, r_an = O.cosine_dist(a, p, n)
    assert rel_err(s_ap.cpu().numpy(), r_ap) < 1e-5 and rel_err(s_an.cpu().numpy(), r_an) < 1e-5


def test_gather_gram_mining(ops):
    rows, d = 64, 128
    e = rand((rows, d), 20)
    labels = (np.arange(rows) // 4).astype(np.int32)
    idx = np.array([5, 0, 63, 7], dtype=np.int32)
    got = ops.gather_rows(dev(e), torch.as_tensor(idx).cuda())
    assert rel_err(got.cpu().numpy(), e[idx].astype(np.float32)) < 1e-7
    g = ops.gram_cosine(dev(e))
    gr = O.gram_cosine(e)
    assert rel_err(g.cpu().numpy(), gr) < 1e-5
    anchor = np.arange(rows, dtype=np.int32)
    pos = (anchor // 4 * 4 + (anchor + 1) % 4).astype(np.int32)
    neg = ops.mine_semihard(g, torch.as_tensor(labels).cuda(), torch.as_tensor(anchor).cuda(), torch.as_tensor(pos).cuda())
    ref = O.mine_semihard(g.cpu().numpy().astype(np.float64), labels, anchor, pos)
    assert np.array_equal(neg.cpu().numpy(), ref)
    # single identity -> -1
    one = ops.mine_semihard(g, torch.zeros(rows, dtype=torch.int32).cuda(), torch.as_tensor(anchor).cuda(), torch.as_tensor(pos).cuda())
    assert (one.cpu().numpy() == -1).all()


def test_optimisers(ops):
    n = 1000
    w, g = rand((n,), 21), rand((n,), 22)
    wd_ = dev(w)
    ops.sgd_update(wd_, dev(g), 0.01, 1e-5, 1.0 / 64)
    assert rel_err(wd_.cpu().numpy(), O.sgd_step(w, g, 0.01, 1e-5, 1.0 / 64)) < 1e-6
    wd_, m, v = dev(w), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    wr, mr, vr = w.copy(), np.zeros(n), np.zeros(n)
    for t in (1, 2, 3):
        ops.adam_update(wd_, dev(g), m, v, 2.4e-4, t, wd=1e-5, rescale=1.0 / 64)
        wr, mr, vr = O.adam_step(wr, g, mr, vr, t, 2.4e-4, 1e-5, 1.0 / 64)
    assert rel_err(wd_.cpu().numpy(), wr) < 1e-6


@pytest.mark.parametrize("metric,sub", [(0, False), (0, True), (1, False), (1, True)])
def test_lfw_evaluator_matches_reference_golden(ops, metric, sub):
    """The device LFW evaluator against vectors produced by the reference's own calculate_roc."""
    from improving_face_recognition_performance_using_triplet_loss_amd import lfw
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lfw_roc.npz"))
    key = "m%d_s%d" % (metric, int(sub))
    tpr, fpr, acc = lfw.calculate_roc(z[key + "_thresholds"], dev(z["emb1"]), dev(z["emb2"]), z["issame"], 10, metric, sub)
    # fp32 distances: a pair sitting within 1e-6 of a threshold may flip one count out of 60 per fold
    assert np.abs(acc - z[key + "_acc"]).max() <= 1.0 / 60 + 1e-9 and abs(acc.mean() - z[key + "_acc"].mean()) < 2e-3
    assert np.abs(tpr - z[key + "_tpr"]).max() < 5e-3 and np.abs(fpr - z[key + "_fpr"]).max() < 5e-3


FUSED_CASES = [
    # batch, h, w, cin, cout, k, pad, ways, pool
    (2, 12, 10, 3, 99, 5, 2, 3, True),     # conv1-like -> MFM3 -> pool
    (3, 8, 8, 66, 198, 3, 1, 3, True),     # conv2-like, NT = 13
    (2, 7, 7, 174, 261, 3, 1, 3, True),    # odd map: 7 -> 3 (floor), NT = 17
    (2, 6, 6, 132, 387, 3, 1, 3, True),    # NT = 25
    (2, 9, 11, 44, 99, 3, 1, 3, False),    # conv_res -> MFM3 (no pooling)
    (2, 5, 5, 258, 387, 1, 0, 3, False),   # conv_r 1x1
    (2, 10, 8, 3, 96, 5, 2, 2, True),      # LightCNN-style MFM2 + pool
    (1, 6, 6, 48, 80, 3, 1, 2, False),     # MFM2, NT rounds 5 -> 5
    (2, 3, 3, 64, 513, 3, 0, 3, False),    # fc1-like: 513 channels -> several channel blocks
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_conv_mfm_pool_fused(ops, case):
    """Fused conv + bias + MFM (+ pool) against the oracle chain, and BITWISE against the unfused kernels (same MFMA
    order per pixel, same tie rules), forward and backward."""
    b, h, w, cin, cout, k, pad, ways, pool = case
    x = rand((b, cin, h, w), 31)
    wt = rand((cout, cin, k, k), 32, 0.2)
    bias = rand((cout,), 33)
    d = ops.conv_desc(b, h, w, cin, cout, k, k, pad, pad)
    assert ops.conv_mfm_supported(d)
    xd = to_nhwc(x)
    wp = ops.conv_pack_weights(d, dev(wt))
    bp = torch.zeros(d.n_pad16, device="cuda")
    bp[:cout] = dev(bias)
    y_ref = O.conv2d(x, wt, bias, (pad, pad))
    z_ref = O.mfm3(y_ref) if ways == 3 else O.mfm2(y_ref)
    if pool:
        z_ref = O.maxpool2(z_ref)
    co = z_ref.shape[1]
    for order, nsplit in ((O.ORDER_GROUP, 0), (O.ORDER_RES, 0), (O.ORDER_GROUP, 2), (O.ORDER_RES, 3)):
        d.tune_fwd = nsplit << 4  # channel blocks: 0 = heuristic
        z, route = ops.conv_mfm_fwd(d, xd, wp, bp, ways, order, pool)
        d.tune_fwd = 0
        assert rel_err(from_nhwc(z, co), z_ref) < TOL
        # unfused chain on the device
        y = ops.conv_fwd(d, xd, wp, bp)
        mf = ops.mfm_fwd(y, cout, ways)
        zu = ops.maxpool2_fwd(mf, co) if pool else mf
        assert torch.equal(z, zu)
        dz = rand(z_ref.shape, 34)
        dzd = to_nhwc(dz)
        dyf = ops.mfm_pool_bwd(d, route, dzd, ways, pool)
        dmf = ops.maxpool2_bwd(mf, dzd, co) if pool else dzd
        dyu = ops.mfm_bwd(y, dmf, cout, ways, order)
        assert torch.equal(dyf, dyu)
        assert not torch.isnan(dyf).any() and tuple(dyf.shape) == (b, d.hout, d.wout, d.cout_p)


def test_gallery_scores(ops):
    g = rand((1000, 342), 41)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    q = g[[17, 900]] + 0.01 * rand((2, 342), 42)
    scores, vals, idx = ops.gallery_match(dev(q), dev(g), topk=3)
    assert rel_err(scores.cpu().numpy(), q @ g.T) < 1e-5
    assert idx[:, 0].tolist() == [17, 900]


@pytest.mark.parametrize("case", [(2, 3, 9, 11, 5, 2), (3, 1, 8, 8, 5, 2), (1, 3, 7, 5, 3, 1), (2, 2, 6, 6, 7, 3)])
@pytest.mark.parametrize("bf16", [False, True])
def test_rowpack_nchw(ops, case, bf16):
    """efm_rowpack_nchw: y[b][h][w][j*c + ch] = x[b][ch][h][w + j - pad] with zeros outside the row and in the pad channels —
    bit-exact in fp32, the bf16 rounding of the same values in bf16 (the row-packed input of the first convolution,
    ref: efm_symbol.py:84 / lightcnn.py:82: the im2col columns of one kernel row)."""
    b, c, h, w, kw, pad = case
    x = rand((b, c, h, w), 5).astype(np.float32)
    got = ops.rowpack_nchw(dev(x), kw, pad, bf16=bf16)
    cp = (kw * c + 7) // 8 * 8 if bf16 else (kw * c + 3) // 4 * 4
    assert tuple(got.shape) == (b, h, w, cp)
    ref = np.zeros((b, h, w, cp), np.float32)
    for j in range(kw):
        for ww in range(w):
            wi = ww + j - pad
            if 0 <= wi < w:
                ref[:, :, ww, j * c:(j + 1) * c] = x[:, :, :, wi].transpose(0, 2, 1)
    if bf16:
        assert torch.equal(got.cpu(), torch.from_numpy(ref).to(torch.bfloat16))
    else:
        assert np.array_equal(got.cpu().numpy(), ref)


def test_rowpacked_first_convolution_equals_the_plain_plan(ops, monkeypatch):
    """The first convolution as a kh x 1 convolution on the row-packed input (Plan, EFM_ROWPACK=1, the default) against the same
    network with the plain 5x5 convolution (EFM_ROWPACK=0): same parameters in MXNet layout in and out, embeddings / loss / every
    exported gradient equal to fp32 summation order (the products and their order in k are the same; only the zero pads between
    them move)."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    batch, image = 8, 48
    x = synth.images(batch, 3, image, 21)
    neg = synth.negative_indices(synth.parity_labels(batch, images_per_identity=2), 3).cuda()
    demb = (synth.uniform01(batch * 128, 7).view(batch, 128) * 2 - 1).contiguous()
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("EFM_ROWPACK", mode)
        tr = TripletTrainer(batch, image=image, seed=5)
        ps = tr.plan.params["conv1_weight"]
        assert ps.rowpack == (mode == "1") and tuple(ps.mx_shape) == (99, 3, 5, 5)
        assert (ps.desc.kh, ps.desc.kw, ps.desc.cin) == ((5, 1, 15) if mode == "1" else (5, 5, 3))
        loss = tr.forward_loss(x, neg).clone()
        tr.backward(demb=demb)
        res[mode] = (tr.plan.export_params(tr.flat), tr.last["emb"].clone(), loss, tr.plan.export_params(tr.grad))
    p0, e0, l0, g0 = res["0"]
    p1, e1, l1, g1 = res["1"]
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k                      # same Xavier draw, same MXNet-layout parameters
    assert rel_err(e1.cpu().numpy(), e0.cpu().numpy()) < 1e-5 and rel_err(l1.cpu().numpy(), l0.cpu().numpy()) < 1e-5
    # gradients: the conv1 outputs of the two plans differ in the last bit, which flips a few max / min / pool routes downstream
    # (see test_winograd_gpu.py) — every gradient, conv1's included, sees those flips; a wrong re-indexing would be O(1)
    worst = max(rel_err(g1[k].cpu().numpy(), g0[k].cpu().numpy()) for k in g0)
    a, b = g1["conv1_weight"].double().flatten(), g0["conv1_weight"].double().flatten()
    assert worst < 5e-2, worst
    assert float((a * b).sum() / (a.norm() * b.norm())) > 0.99999


def test_kx1_convolution_kernels(ops):
    """kh x 1 kernels (the shape of the row-packed first convolution) through the forward, fused-epilogue and weight-gradient
    kernels against the fp64 oracle run on the equivalent full-size problem: 5x1, pad (2, 0), 15 -> 30 channels."""
    b, h, w, cin, cout = 2, 9, 7, 15, 30
    x, wt, bias = rand((b, cin, h, w), 1), rand((cout, cin, 5, 1), 2, 0.3), rand((cout,), 3)
    d = ops.conv_desc(b, h, w, cin, cout, 5, 1, 2, 0)
    assert (d.hout, d.wout) == (h, w)
    wp = ops.conv_pack_weights(d, dev(wt))
    bp = torch.zeros(d.n_pad16, device="cuda")
    bp[:cout] = dev(bias)
    y = ops.conv_fwd(d, to_nhwc(x), wp, bp)
    xp = np.pad(x, ((0, 0), (0, 0), (2, 2), (0, 0)))
    ref = np.zeros((b, cout, h, w))
    for a in range(5):
        ref += np.einsum("bchw,nc->bnhw", xp[:, :, a:a + h, :], wt[:, :, a, 0])
    ref += bias[None, :, None, None]
    assert rel_err(from_nhwc(y, cout), ref) < 2e-4
    dy = rand((b, cout, h, w), 4)
    dw, db = ops.conv_bwd_weight(d, to_nhwc(x), to_nhwc(dy))
    dw_ref = np.stack([np.einsum("bchw,bnhw->nc", xp[:, :, a:a + h, :], dy) for a in range(5)], axis=2)[..., None]
    assert rel_err(ops.conv_unpack_weights(d, dw).cpu().numpy(), dw_ref) < 2e-4
    assert rel_err(db[:cout].cpu().numpy(), dy.sum(axis=(0, 2, 3))) < 2e-4
