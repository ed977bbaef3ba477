"""CPU tests of the oracle itself: hand-computed known answers for every [MX-assumed] operator semantic
(SURVEY.md §8c), agreement of the two independent restatements, and the exact work counts of BASELINE.md.
The reference holds no golden vectors for this path (parity unpinned), so these KATs are the pins."""
import math

import numpy as np
import pytest
import torch

from oracle import efm_oracle as O
from oracle import efm_oracle_torch as OT


def test_mfm3_kat_and_tie_rules():
    # 1 x 6 x 1 x 2: slices s0 = ch0-1, s1 = ch2-3, s2 = ch4-5
    x = np.array([[[[1.0, 5.0]], [[-2.0, 0.0]], [[3.0, 5.0]], [[4.0, 0.5]], [[3.0, 2.0]], [[-7.0, 0.5]]]])
    y = O.mfm3(x)
    # ch0: max(s0c0,s1c0,s2c0) = max(1,3,3)=3 ; max(5,5,2)=5 | ch1: max(-2,4,-7)=4 ; max(0,.5,.5)=.5
    # ch2: min(1,3,3)=1 ; min(5,5,2)=2 | ch3: min(-2,4,-7)=-7 ; min(0,.5,.5)=0
    assert np.array_equal(y[0, :, 0, :], np.array([[3, 5], [4, 0.5], [1, 2], [-7, 0]]))
    g = np.arange(1, 9, dtype=float).reshape(1, 4, 1, 2)  # gmax = [[1,2],[3,4]], gmin = [[5,6],[7,8]]
    d_group = O.mfm3_bwd(x, g, O.ORDER_GROUP)
    d_res = O.mfm3_bwd(x, g, O.ORDER_RES)
    # position (c0, w0): s = (1,3,3): max tie between s1 and s2. GROUP = max(max(s0,s1),s2): lhs (=s1 via max1) wins;
    # RES = max(s2, max1): s2 wins.  min = s0 either way.
    assert d_group[0, 2, 0, 0] == 1 and d_group[0, 4, 0, 0] == 0
    assert d_res[0, 2, 0, 0] == 0 and d_res[0, 4, 0, 0] == 1
    assert d_group[0, 0, 0, 0] == 5 and d_res[0, 0, 0, 0] == 5
    # position (c0, w1): s = (5,5,2): max tie s0/s1 -> s0 (lhs of maximum(s0,s1)); min = s2
    assert d_group[0, 0, 0, 1] == 2 and d_group[0, 2, 0, 1] == 0 and d_group[0, 4, 0, 1] == 6
    # position (c1, w1): s = (0,.5,.5): max tie s1/s2, min = s0
    assert d_group[0, 3, 0, 1] == 4 and d_res[0, 5, 0, 1] == 4 and d_group[0, 1, 0, 1] == 8
    # gradient mass is conserved
    assert d_group.sum() == g.sum() and d_res.sum() == g.sum()


def test_mfm2_kat():
    x = np.array([[[[1.0]], [[2.0]], [[2.0]], [[-1.0]]]])
    assert np.array_equal(O.mfm2(x)[0, :, 0, 0], [2.0, 2.0])
    d = O.mfm2_bwd(x, np.array([[[[10.0]], [[20.0]]]]))
    assert np.array_equal(d[0, :, 0, 0], [0, 20, 10, 0])


def test_conv_is_cross_correlation_with_bias():
    x = np.arange(9, dtype=float).reshape(1, 1, 3, 3)
    w = np.array([[[[1.0, 2.0], [3.0, 4.0]]]])
    y = O.conv2d(x, w, np.array([0.5]), (0, 0))
    # top-left: 0*1 + 1*2 + 3*3 + 4*4 + .5 = 27.5 (a true convolution would flip the kernel and give 13.5)
    assert y.shape == (1, 1, 2, 2) and y[0, 0, 0, 0] == 27.5 and y[0, 0, 1, 1] == 4 * 1 + 5 * 2 + 7 * 3 + 8 * 4 + 0.5
    yp = O.conv2d(x, np.ones((1, 1, 3, 3)), None, (1, 1))
    assert yp.shape == (1, 1, 3, 3) and yp[0, 0, 0, 0] == 0 + 1 + 3 + 4 and yp[0, 0, 1, 1] == 36


def test_conv_bwd_matches_finite_differences():
    rng = np.random.default_rng(0)
    x, w, b = rng.normal(size=(2, 3, 5, 4)), rng.normal(size=(4, 3, 3, 3)), rng.normal(size=4)
    dy = rng.normal(size=(2, 4, 5, 4))
    dx, dw, db = O.conv2d_bwd(x, w, dy, (1, 1))
    eps = 1e-6
    for arr, grad, idx in ((x, dx, (1, 2, 3, 1)), (w, dw, (2, 1, 0, 2))):
        old = arr[idx]
        arr[idx] = old + eps
        up = (O.conv2d(x, w, b, (1, 1)) * dy).sum()
        arr[idx] = old - eps
        dn = (O.conv2d(x, w, b, (1, 1)) * dy).sum()
        arr[idx] = old
        assert abs((up - dn) / (2 * eps) - grad[idx]) < 1e-6
    assert np.allclose(db, dy.sum(axis=(0, 2, 3)))


def test_pool_floor_7_to_3_and_first_max():
    x = np.arange(49, dtype=float).reshape(1, 1, 7, 7)
    y = O.maxpool2(x)
    assert y.shape == (1, 1, 3, 3)  # 'valid' convention: the 7th row / column is dropped
    assert y[0, 0, 0, 0] == 8 and y[0, 0, 2, 2] == 40
    x2 = np.zeros((1, 1, 2, 2))
    d = O.maxpool2_bwd(x2, np.array([[[[3.0]]]]))
    assert d[0, 0, 0, 0] == 3 and d.sum() == 3  # all equal -> first in scan order
    d7 = O.maxpool2_bwd(x, np.ones((1, 1, 3, 3)))
    assert d7[0, 0, 6].sum() == 0 and d7[0, 0, :, 6].sum() == 0 and d7.sum() == 9


def test_fully_connected_flattens_nchw():
    x = np.arange(8, dtype=float).reshape(1, 2, 2, 2)
    w = np.zeros((1, 8))
    w[0, 5] = 1.0  # NCHW flatten: index 5 = channel 1, h 0, w 1
    assert O.fully_connected(x, w, np.array([1.0]))[0, 0] == x[0, 1, 0, 1] + 1.0


def test_norms():
    x = np.array([[3.0, 4.0], [0.0, 2.0]])
    y, n = O.l2norm_row(x)
    assert np.allclose(n, [5, 2]) and np.allclose(y, [[0.6, 0.8], [0, 1]])
    yf, nf = O.l2norm_frob(x)
    assert math.isclose(nf, math.sqrt(29)) and np.allclose(yf, x / math.sqrt(29))
    # backward of a unit-norm map is orthogonal to y
    dy = np.array([[1.0, -2.0], [0.5, 0.25]])
    dx = O.l2norm_row_bwd(y, n, dy)
    assert np.allclose((dx * y).sum(1), 0)


def test_triplet_loss_kat():
    a, p, n = np.array([[0.0, 0.0], [0.0, 0.0]]), np.array([[1.0, 0.0], [0.1, 0.0]]), np.array([[0.0, 1.0], [2.0, 0.0]])
    assert np.allclose(O.triplet_loss(a, p, n, 0.2), [0.2, 0.0])    # 1 - 1 + .2 ; .01 - 4 + .2 < 0 -> relu off
    assert np.allclose(O.triplet_loss(a, p, n, 0.5), [0.5, 0.0])
    da, dp, dn = O.triplet_loss_bwd(a, p, n, O.triplet_loss(a, p, n, 0.2), np.ones(2))
    assert np.allclose(da[0], 2 * (n[0] - p[0])) and np.allclose(dp[0], 2 * p[0]) and np.allclose(dn[0], -2 * n[0])
    assert not da[1].any() and not dp[1].any() and not dn[1].any()
    # unit-norm rows: L = max(0, 2(d_ap - d_an) + m) with cosine distance d = 1 - cos  
    rng = np.random.default_rng(1)
    u = [v / np.linalg.norm(v, axis=1, keepdims=True) for v in rng.normal(size=(3, 4, 8))]
    s_ap, s_an = O.cosine_dist(u[0], u[1], u[2])
    assert np.allclose(O.triplet_loss(u[0], u[1], u[2], 0.2), np.maximum(2 * ((1 - s_ap) - (1 - s_an)) + 0.2, 0))


def test_negative_pick_and_mining():
    labels = np.array([0, 0, 1, 1])
    assert np.array_equal(O.pick_negatives(labels, labels, [0, 1, 3, 2, 2, 0, 3, 1]), [3, 2, 0, 1])
    e = np.array([[1.0, 0.0], [0.9, 0.1], [0.0, 1.0], [0.6, 0.8], [-1.0, 0.0]])
    g = O.gram_cosine(e)
    lab = np.array([0, 0, 1, 1, 2])
    # anchor 0 / positive 1: d_ap ~ 0.006; negatives d: idx2 1.0, idx3 0.4, idx4 2.0 -> semi-hard (smallest above d_ap) = 3
    assert O.mine_semihard(g, lab, [0], [1])[0] == 3
    # anchor 2 / positive 3: d_ap = 0.2; all other-label d: idx0 1.0, idx1 ~.89, idx4 1.0 -> 1
    assert O.mine_semihard(g, lab, [2], [3])[0] == 1
    assert O.mine_semihard(g[:2, :2], np.array([0, 0]), [0], [1])[0] == -1


def test_optimisers_and_schedule():
    w, g = np.array([1.0, -2.0]), np.array([0.5, 0.25])
    assert np.allclose(O.sgd_step(w, g, 0.1, 0.01, 0.5), w - 0.1 * (0.5 * g + 0.01 * w))
    w1, m1, v1 = O.adam_step(w, g, np.zeros(2), np.zeros(2), 1, 0.001, 0.0, 1.0)
    # first Adam step moves every weight by ~lr against the gradient sign
    assert np.allclose(w1, w - 0.001 * np.sign(g), atol=1e-6)
    assert O.factor_scheduler(1.0, 5, 6, 0.88) == 1.0 and math.isclose(O.factor_scheduler(1.0, 7, 6, 0.88), 0.88)
    assert math.isclose(O.factor_scheduler(1.0, 13, 6, 0.88), 0.88 ** 2)


def test_softmax_ce():
    z = np.array([[1.0, 2.0, 3.0]])
    l = O.softmax_cross_entropy(z, np.array([2]))
    assert math.isclose(l[0], -math.log(math.exp(3) / (math.exp(1) + math.exp(2) + math.exp(3))))
    d = O.softmax_cross_entropy_bwd(z, np.array([2]), np.ones(1))
    assert math.isclose(d.sum(), 0.0, abs_tol=1e-12) and d[0, 2] < 0


def test_efm29_table_matches_survey_counts():
    layers = O.efm29_layers(3)
    assert len(layers) == 29
    names = [l[0] for l in layers]
    assert names[:5] == ["conv1", "conv2_res", "conv2_res_r", "conv2_r", "conv2"] and "conv31_res_r" in names and names[-1] == "conv5"
    shapes = O.efm29_param_shapes(3, 112)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 9068013
    sizes = {"1": 112, "2": 56, "3": 28, "4": 14, "5": 7}
    flops = 0
    for name, co, ci, kh, kw, _ in layers:
        hw = sizes[name[4]] ** 2
        flops += 2 * hw * co * ci * kh * kw
    assert flops == 5145993720
    assert flops + 2 * 513 * 1566 == 5147600436
    assert 3 * (flops + 2 * 513 * 1566) - 2 * 112 * 112 * 99 * 3 * 25 == 15256522908


def test_numpy_and_torch_restatements_agree():
    shapes = O.efm29_param_shapes(3, 32)
    params = O.init_params(shapes, 42)
    x = O.uniform01(4 * 3 * 32 * 32, 1234).reshape(4, 3, 32, 32)
    wh = O.uniform_pm((128, 342), 777, O.xavier_uniform_scale((128, 342)))
    neg = np.array([1, 0], dtype=np.int32)
    demb = np.random.default_rng(3).uniform(-1, 1, (4, 128))
    loss, emb, feat, grads, gh = O.train_step_loss(params, wh, x, neg, 0.2, demb=demb)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    twh = torch.tensor(wh, dtype=torch.float64, requires_grad=True)
    tl, te, tf = OT.train_step(tp, twh, torch.tensor(x), torch.tensor(neg.astype(np.int64)), 0.2, demb=torch.tensor(demb))

    def rel(a, b):
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)

    assert rel(feat, tf.numpy()) < 1e-10 and rel(emb, te.numpy()) < 1e-10 and rel(loss, tl.numpy()) < 1e-10
    assert max(rel(grads[k], tp[k].grad.numpy()) for k in params) < 1e-5     # <= 1e-5 before freezing
    assert rel(gh, twh.grad.numpy()) < 1e-5
    # routing hand-over with the run's own values is the identity
    _, _, _, grads2, _ = O.train_step_loss(params, wh, x, neg, 0.2, demb=demb, routing={})
    assert all(np.array_equal(grads[k], grads2[k]) for k in grads)


def test_xavier_and_generator_are_portable():
    assert math.isclose(O.xavier_uniform_scale((128, 342)), math.sqrt(3.0 / ((342 + 128) / 2.0)))
    assert math.isclose(O.xavier_uniform_scale((99, 3, 5, 5)), math.sqrt(3.0 / ((75 + 2475) / 2.0)))
    u = O.uniform01(5, 1234)
    # frozen values of the splitmix64 stream (seed 1234): any platform / numpy version must reproduce them
    assert np.array_equal(u, O.uniform01(7, 1234)[:5]) and (0 <= u).all() and (u < 1).all()
    assert np.array_equal((u * (1 << 24)).astype(np.int64), (u * (1 << 24)).round().astype(np.int64))


@pytest.mark.parametrize("metric", [0, 1])
def test_lfw_protocol_perfectly_separable(metric):
    rng = np.random.default_rng(0)
    e1 = rng.normal(size=(60, 16))
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    same = np.arange(60) % 2 == 0
    near = e1 + 0.05 * rng.normal(size=e1.shape)  # not exactly equal: arccos(1 + 1e-16) is NaN in the reference too
    near /= np.linalg.norm(near, axis=1, keepdims=True)
    e2 = np.where(same[:, None], near, -near)
    tpr, fpr, acc = O.lfw_roc(np.arange(0, 4, 0.01), e1, e2, same, nrof_folds=10, metric=metric)
    assert acc.mean() > 0.95 and tpr[-1] == 1.0 and fpr[0] == 0.0  # first-best-threshold rule can miss a test pair


@pytest.mark.parametrize("metric,sub", [(0, False), (0, True), (1, False), (1, True)])
def test_lfw_restatement_matches_reference_golden(metric, sub):
    """Golden vectors produced by the reference's own calculate_roc (tests/golden/make_lfw_golden.py)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lfw_roc.npz"))
    key = "m%d_s%d" % (metric, int(sub))
    tpr, fpr, acc = O.lfw_roc(z[key + "_thresholds"], z["emb1"], z["emb2"], z["issame"], 10, metric, sub)
    assert np.allclose(tpr, z[key + "_tpr"], atol=1e-12) and np.allclose(fpr, z[key + "_fpr"], atol=1e-12)
    assert np.array_equal(acc, z[key + "_acc"])
    assert 0.7 < acc.mean() < 0.99  # the fixture is neither trivial nor random


def test_oracle_reproduces_mini_efm_golden():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_mini_efm_golden as M
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mini_efm.npz"))
    params, w_head, x, neg, demb = M.inputs()
    loss, emb, feat, grads, g_head = O.train_step_loss(params, w_head, x, neg, 0.2, demb=demb)
    assert np.allclose(loss, z["loss"], rtol=1e-12) and np.allclose(emb, z["emb"], rtol=1e-10, atol=1e-14)
    assert np.allclose(feat, z["feat"], rtol=1e-10, atol=1e-14)
    assert np.allclose(np.array([np.abs(grads[k]).sum() for k in sorted(grads)]), z["grad_abs_sums"], rtol=1e-9)


def test_factor_scheduler_module_matches_oracle():
    from improving_face_recognition_performance_using_triplet_loss_amd.nn import FactorScheduler
    s = FactorScheduler(step=6, factor=0.88, stop_factor_lr=5e-15)
    s.base_lr = 2.4e-4
    for n in (1, 6, 7, 12, 13, 100, 1000):
        assert math.isclose(s(n), O.factor_scheduler(2.4e-4, n, 6, 0.88, 5e-15), rel_tol=1e-12)


def test_oracle_reproduces_config1_golden_rows():
    """BASELINE configs[0] fixture (tests/golden/config1_efm112.npz): the torch restatement regenerates rows 0, 31 and 63 (every
    embedding row depends on its own image only) — pins the oracle against drift at the BASELINE geometry; the fp32 run of the
    same restatement shows the headroom under the 1e-3 tolerance."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_config1_golden as M
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "config1_efm112.npz"))
    rows = [0, 31, 63]
    params, w_head, x, neg = M.inputs(rows)
    feat, emb = M.forward(params, w_head, x)
    assert np.abs(emb - z["emb"][rows]).max() < 1e-6 * np.abs(z["emb"]).max()
    assert np.abs(feat.sum(1) / z["feat_sum"][rows] - 1).max() < 1e-9
    loss, cosines = M.loss_and_cosines(z["emb"].astype(np.float64), z["neg"])
    assert np.abs(loss - z["loss"]).max() < 1e-6 and np.abs(cosines - z["cosines"]).max() < 1e-6
    p32 = {k: v.astype(np.float32) for k, v in params.items()}
    _, emb32 = M.forward(p32, w_head.astype(np.float32), x.astype(np.float32))
    assert np.abs(emb32 - z["emb"][rows]).max() < 1e-4 * np.abs(z["emb"]).max()


@pytest.mark.parametrize("nshards", [2, 4])
def test_shard_gradients_sum_to_the_full_batch_gradient(nshards):
    """The data-parallel contract (SURVEY §8c iv / §8e): with negatives local to a shard, as the reference draws them
    (train_efm.py:234-239), the per-shard parameter gradients SUM to the single-process gradient of the same triplets — fp64."""
    import torch
    from oracle import efm_oracle_torch as OT
    image, anchors = 32, 8
    params = O.init_params(O.efm29_param_shapes(3, image), 42)
    wh = O.uniform_pm((128, 342), 777, O.xavier_uniform_scale((128, 342)))
    x = O.uniform01(2 * anchors * 3 * image * image, 9).reshape(2 * anchors, 3, image, image)
    per = anchors // nshards
    neg_local = (np.arange(per) + 1) % per                      # inside a shard: the next anchor of that shard
    neg_full = np.concatenate([s * per + neg_local for s in range(nshards)])

    def grads(xs, neg):
        tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
        twh = torch.tensor(wh, requires_grad=True)
        OT.train_step(tp, twh, torch.tensor(xs), torch.tensor(neg.astype(np.int64)), 0.2)
        return {**{k: t.grad.numpy() for k, t in tp.items()}, "head": twh.grad.numpy()}
    full = grads(x, neg_full)
    total = None
    for s in range(nshards):
        rows = np.concatenate([np.arange(s * per, (s + 1) * per), anchors + np.arange(s * per, (s + 1) * per)])  # its anchors ; its positives
        g = grads(x[rows], neg_local)
        total = g if total is None else {k: total[k] + g[k] for k in g}
    for k in full:
        assert np.abs(total[k] - full[k]).max() <= 1e-10 * (np.abs(full[k]).max() + 1e-30), k


# ---- the Gluon variant: LightCNN_29 + the train_efm.py step (ref: lightcnn.py:6-133, train_efm.py:229-245) ------------------
def test_lightcnn29_table_weight_sharing_and_gluon_keys():
    layers = O.lightcnn29_layers(1)
    assert len(layers) == 17 and layers[0] == ("g1_conv1", 99, 1, 5, 2)          # 29 applied convolutions, 17 distinct + Dense(1026)
    assert dict((n, (co, ci)) for n, co, ci, _, _ in layers)["g4_res_conv0"] == (387, 172)
    assert dict((n, (co, ci)) for n, co, ci, _, _ in layers)["g4_res_conv1"] == (258, 258)
    sh = O.lightcnn29_param_shapes(1, 128, 8398)                                  # train_efm.py:154-159: 1x128x128, 8398 classes
    assert sh["fc1_weight"] == (1026, 174 * 4 * 4) and sh["dense1_weight"] == (8398, 684) and sh["batchnorm0_gamma"] == (684,)
    keys = O.gluon_struct_names()
    assert keys["g1_conv1_weight"] == "conv_net.0.conv_op_2.weight" and keys["g5_res_conv1_bias"] == "conv_net.11.conv_op_2.bias"
    assert keys["g5_conv0_weight"] == "conv_net.12.conv_op_1.weight" and keys["fc1_bias"] == "conv_net.15.bias"
    assert keys["batchnorm0_running_var"] == "fc1.0.running_var" and keys["dense1_weight"] == "fc2.1.weight"
    import lightcnn
    assert lightcnn.gluon_param_names() == keys                                   # product and oracle derive the same table independently


def test_batchnorm_kat():
    x = np.array([[1.0, 10.0], [3.0, 10.0], [5.0, 16.0]])
    y, (xhat, inv, mean, var) = O.batchnorm_train(x, np.array([2.0, 1.0]), np.array([0.5, 0.0]), eps=0.0)
    assert np.allclose(mean, [3, 12]) and np.allclose(var, [8 / 3, 8])           # BIASED variance
    assert np.allclose(y[:, 0], 2 * np.array([-2, 0, 2]) / math.sqrt(8 / 3) + 0.5)
    rm, rv = O.batchnorm_running_update(np.zeros(2), np.ones(2), mean, var)
    assert np.allclose(rm, [0.3, 1.2]) and np.allclose(rv, [0.9 + 0.8 / 3, 0.9 + 0.8])
    # backward against finite differences
    rng = np.random.default_rng(0)
    x, g, b, dy = rng.normal(size=(6, 5)), rng.normal(size=5), rng.normal(size=5), rng.normal(size=(6, 5))
    _, cache = O.batchnorm_train(x, g, b)
    dx, dg, db = O.batchnorm_train_bwd(cache, g, dy)
    f = lambda xx: (O.batchnorm_train(xx, g, b)[0] * dy).sum()  # noqa: E731
    e = np.zeros_like(x)
    e[2, 3] = 1e-6
    assert abs((f(x + e) - f(x - e)) / 2e-6 - dx[2, 3]) < 1e-6
    assert np.allclose(db, dy.sum(0))


def test_lightcnn29_restatements_agree_and_reproduce_fixture():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_lightcnn29_golden as M
    params, x, labels, neg = M.inputs()
    rng = np.random.default_rng(3)
    mask = (rng.uniform(size=(2 * M.BATCH, 684)) > 0.7).astype(np.float64)       # Dropout(.7) replayed through an explicit mask
    for m in (None, mask):
        r = O.train_efm_step(params, x, labels, neg, M.MARGIN, M.ALPHA, dropout_mask=m)
        tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
        out, fc, tl, idl, loss = OT.train_efm_step(tp, torch.tensor(x), torch.tensor(labels), torch.tensor(neg.astype(np.int64)), M.MARGIN,
                                                   M.ALPHA, dropout_mask=None if m is None else torch.tensor(m))
        assert np.abs(out.numpy() - r["out"]).max() < 1e-10 and np.abs(loss.numpy() - r["loss"]).max() < 1e-10
        for k, g in r["grads"].items():
            assert np.abs(tp[k].grad.numpy() - g).max() <= 1e-9 * np.abs(g).max(), k
        if m is None:
            z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lightcnn29_step.npz"))
            assert np.allclose(r["loss"], z["loss"], rtol=1e-12) and np.allclose(r["fc1_out"], z["fc1_out"], rtol=1e-10, atol=1e-12)
            assert np.allclose([np.abs(r["grads"][str(k)]).sum() for k in z["names"]], z["grad_abs_sums"], rtol=1e-10)
    # a shared convolution's gradient is the SUM over its uses: perturbing the shared weight moves the loss by <grad, dw>
    # (alpha = 0: the detached negatives make the triplet term's true derivative differ from its gradient BY DESIGN)
    r = O.train_efm_step(params, x, labels, neg, M.MARGIN, 0.0)
    dw = O.uniform_pm(params["g4_res_conv1_weight"].shape, 5, 1e-7)
    p2 = dict(params)
    p2["g4_res_conv1_weight"] = params["g4_res_conv1_weight"] + dw
    r2 = O.train_efm_step(p2, x, labels, neg, M.MARGIN, 0.0)
    pred = (r["grads"]["g4_res_conv1_weight"] * dw).sum()
    assert abs((r2["loss"].sum() - r["loss"].sum()) - pred) < 1e-3 * abs(pred) + 1e-12


def test_host_loops_restatement_agrees_with_the_vectorised_forms():
    """oracle/host_loops.py (the reference's literal per-sample loops, train_efm.py:234-239 and :26-34, the thing bench.py's
    `host_loops` leg times) against the vectorised statements of the same arithmetic: every picked negative has another label,
    the copied rows are the indexed rows, the cosine lists equal the row-wise formula."""
    import random

    import torch

    from oracle import host_loops as H
    b, d = 24, 16
    lab = (torch.arange(b) % 6).to(torch.float32)
    lab2 = torch.cat([lab, lab])
    fc = torch.as_tensor(np.random.default_rng(3).uniform(-1, 1, size=(2 * b, d)), dtype=torch.float32)
    neg, idx = H.pick_negatives_loop(lab2, fc, b, random.Random(7))
    assert all(int(lab2[j]) != int(lab2[i]) and 0 <= j < b for i, j in enumerate(idx))
    assert torch.equal(neg, fc[torch.tensor(idx)])
    pd, nd = H.cosine_dist_loop(fc[:b], fc[b:], neg, b)
    rows = H.csv_rows(pd, nd, b)
    cos = torch.nn.functional.cosine_similarity
    assert np.allclose([r[0] for r in rows], cos(fc[:b], fc[b:]).numpy(), atol=1e-6)
    assert np.allclose([r[1] for r in rows], cos(fc[:b], neg).numpy(), atol=1e-6)
