"""End-to-end GPU parity of the EFM-29 embedding path (plan + trainer) against the CPU oracles.

Tolerance is the north star's: 1e-3 relative fp32 (measured here as max|diff| / max|ref|) for embeddings, loss
and every parameter gradient.  Inputs / weights come from the portable splitmix64 generator, nothing is read
from /root/reference.
"""
import numpy as np
import pytest
import torch

from oracle import efm_oracle as O
from oracle import efm_oracle_torch as OT
from tests.util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _make(batch, image, seed=1234, fuse=None):
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    shapes = O.efm29_param_shapes(3, image)
    params = O.init_params(shapes, 42)
    w_head = O.uniform_pm((128, 342), 777, O.xavier_uniform_scale((128, 342)))
    x = O.uniform01(batch * 3 * image * image, seed).reshape(batch, 3, image, image)
    tr = TripletTrainer(batch, image=image, optimizer="sgd", lr=0.05, wd=1e-5, fuse=fuse)
    allp = dict(params)
    allp["head_weight"] = w_head
    tr.plan.load_params(tr.flat, allp)
    return tr, params, w_head, x


def test_param_table_matches_reference_names():
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    tr = TripletTrainer(2, image=112)
    shapes = O.efm29_param_shapes(3, 112)
    mine = {n: tuple(ps.mx_shape) for n, ps in tr.plan.params.items() if n != "head_weight"}
    ref = {n: (s if not n == "fc1_weight" else (513, 174, 3, 3)) for n, s in shapes.items()}
    assert mine == ref
    assert sum(int(np.prod(s)) for s in shapes.values()) == 9068013
    # pack/export round trip is exact
    exp = tr.plan.export_params(tr.flat)
    flat2 = tr.plan.new_flat()
    tr.plan.load_params(flat2, {k: v.cpu().numpy() for k, v in exp.items()})
    assert torch.equal(flat2, tr.flat)


def test_mini_efm_step_vs_numpy_oracle():
    """B=4 (2 anchors + 2 positives), 3x32x32, all 29 convs real width (SURVEY.md §8c fixture (ii))."""
    tr, params, w_head, x = _make(4, 32)
    neg = np.array([1, 0], dtype=np.int32)
    demb = np.random.default_rng(5).uniform(-1, 1, size=(4, 128))
    loss_r, emb_r, feat_r, grads_r, ghead_r = O.train_step_loss(params, w_head, x, neg, 0.2, demb=demb)
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), torch.as_tensor(neg).cuda())
    assert rel_err(tr.last["feat"][:, :342].cpu().numpy(), feat_r) < TOL
    assert rel_err(tr.last["emb"].cpu().numpy(), emb_r) < TOL
    assert rel_err(loss.cpu().numpy(), loss_r) < TOL
    tr.backward(demb=torch.as_tensor(demb, dtype=torch.float32).cuda())
    g = tr.plan.export_params(tr.grad)
    worst = 0.0
    for name, ref in grads_r.items():
        got = g[name].cpu().numpy().reshape(ref.shape)
        worst = max(worst, rel_err(got, ref))
    assert worst < TOL, worst
    assert rel_err(g["head_weight"].cpu().numpy().reshape(128, 342), ghead_r) < TOL
    # the loss's own gradient (difference of nearly equal embeddings: conditioning-limited, see oracle docstring)
    _, _, _, grads_l, _ = O.train_step_loss(params, w_head, x, neg, 0.2)
    tr.backward()
    gl = tr.plan.export_params(tr.grad)
    assert rel_err(gl["conv3_res_weight"].cpu().numpy(), grads_l["conv3_res_weight"]) < 5e-2
    tr.backward(demb=torch.as_tensor(demb, dtype=torch.float32).cuda())
    # SGD update against the oracle formula, rescale = 1/(B/2)
    before = tr.plan.export_params(tr.flat)["conv3_res_weight"].cpu().numpy().astype(np.float64)
    tr.update()
    after = tr.plan.export_params(tr.flat)["conv3_res_weight"].cpu().numpy()
    assert rel_err(after, O.sgd_step(before, grads_r["conv3_res_weight"], 0.05, 1e-5, 0.5)) < 1e-5
    # cosine log rows
    s_ap, s_an = tr.cosine_log()
    r_ap, r_an = O.cosine_dist(emb_r[:2], emb_r[2:], emb_r[neg])
    assert rel_err(s_ap.cpu().numpy(), r_ap) < TOL and rel_err(s_an.cpu().numpy(), r_an) < TOL


def test_112_step_vs_oracles():
    """3x112x112 (the BASELINE geometry, 7->3 floor pooling included), B=8.

    Forward (feature, embedding, loss) against the torch-CPU fp64 oracle: 1e-3.
    Backward against the NumPy fp64 oracle at 1e-3 with the arg-max routes of max/min/pool taken from the HIP
    forward (`routing=`): those gradients are piecewise constant in the forward values, and a last-bit difference
    between two correct forwards flips a few routes — the SAME oracle run in fp32 on the CPU differs from its own
    fp64 run by ~6e-3 on these inputs (asserted below as the noise floor that makes the routing hand-over necessary)."""
    tr, params, w_head, x = _make(8, 112, fuse=False)  # one kernel per graph node: the MFM inputs exist for the route hand-over
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    labels = synth.parity_labels(8, images_per_identity=2)
    neg = synth.negative_indices(labels, 99)
    demb = torch.as_tensor(np.random.default_rng(6).uniform(-1, 1, size=(8, 128)))
    ref = {}
    for dt in (torch.float64, torch.float32):
        tp = {k: torch.tensor(v, dtype=dt, requires_grad=True) for k, v in params.items()}
        twh = torch.tensor(w_head, dtype=dt, requires_grad=True)
        loss_r, emb_r, feat_r = OT.train_step(tp, twh, torch.tensor(x, dtype=dt), neg.long(), 0.2, demb=demb.to(dt))
        grads = {k: t.grad.double().numpy() for k, t in tp.items()}
        ref[dt] = (loss_r.double().numpy(), emb_r.double().numpy(), feat_r.double().numpy(), grads)
    loss_r, emb_r, feat_r, grads_t = ref[torch.float64]
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), neg.cuda())
    assert rel_err(tr.last["feat"][:, :342].cpu().numpy(), feat_r) < TOL
    assert rel_err(tr.last["emb"].cpu().numpy(), emb_r) < TOL
    assert rel_err(loss.cpu().numpy(), loss_r) < TOL
    routing = {k: v.cpu().numpy().astype(np.float64) for k, v in tr.plan.routing_inputs().items()}
    tr.backward(demb=demb.float().cuda())
    g = tr.plan.export_params(tr.grad)
    _, _, _, grads_r, ghead_r = O.train_step_loss(params, w_head, x, neg.numpy(), 0.2, demb=demb.numpy(), routing=routing)
    worst, worst_name = 0.0, None
    for name, r in grads_r.items():
        e = rel_err(g[name].cpu().numpy().reshape(r.shape), r)
        if e > worst:
            worst, worst_name = e, name
    floor = max(rel_err(ref[torch.float32][3][k], grads_t[k]) for k in grads_t)
    print("112 gradient parity (same routes): worst %.3e (%s); fp32-vs-fp64 CPU oracle without route hand-over: %.3e"
          % (worst, worst_name, floor))
    assert worst < TOL, (worst, worst_name)
    assert rel_err(g["head_weight"].cpu().numpy().reshape(128, 342), ghead_r) < TOL


def test_step_is_bitwise_reproducible():
    """No atomics anywhere: two runs of the same step give identical bits (SURVEY.md §7 'hard parts')."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    outs = []
    for _ in range(2):
        tr = TripletTrainer(16, image=64, seed=3)
        x = synth.images(16, 3, 64, 11)
        labels = synth.parity_labels(16, images_per_identity=2)
        neg = synth.negative_indices(labels, 5).cuda()
        tr.step(x, neg)
        outs.append((tr.grad.clone(), tr.flat.clone(), tr.last["loss"].clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_shard_sum_identity():
    """Data parallelism by construction: two ranks with 8 images each (own anchors, positives and LOCAL negatives,
    ref: train_efm.py:234-239) produce gradients whose SUM equals the gradient of one process that runs the same 16
    images with the same triplets — the all-reduce is a plain sum and the 1/global_batch scale lives in the optimiser
    (ref: mutli_gpu_v3.py:159).  Forward values are per-sample and bitwise independent of the batch they sit in, so
    the arg-max routes agree and only the split-K summation order differs: tolerance 1e-4."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    xs, negs, shard_grads, shard_loss = [], [], [], []
    for rank in range(2):
        tr = TripletTrainer(8, image=32, seed=3)
        x = synth.images(8, 3, 32, 100 + rank)
        neg = synth.negative_indices(synth.parity_labels(8, images_per_identity=2), 5 + rank).cuda()
        shard_loss.append(tr.forward_loss(x, neg).clone())
        tr.backward()
        shard_grads.append(tr.grad.clone())
        xs.append(x)
        negs.append(neg)
    big = TripletTrainer(16, image=32, seed=3)
    x = torch.cat([xs[0][:4], xs[1][:4], xs[0][4:], xs[1][4:]])          # [anchors r0, anchors r1 ; positives r0, r1]
    neg = torch.cat([negs[0], negs[1] + 4]).to(torch.int32)
    loss = big.forward_loss(x, neg)
    big.backward()
    assert torch.equal(loss, torch.cat(shard_loss))                       # per-sample forward is batch independent
    total = shard_grads[0] + shard_grads[1]
    err = float((total - big.grad).abs().max() / big.grad.abs().max())
    assert err < 1e-4, err


def test_mini_efm_vs_committed_golden():
    """HIP path against the committed fixture tests/golden/mini_efm.npz (no oracle call at test time)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_mini_efm_golden as M
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "mini_efm.npz"))
    params, w_head, x, neg, demb = M.inputs()
    tr = TripletTrainer(4, image=32)
    allp = dict(params)
    allp["head_weight"] = w_head
    tr.plan.load_params(tr.flat, allp)
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), torch.as_tensor(neg).cuda())
    assert rel_err(loss.cpu().numpy(), z["loss"]) < TOL
    assert rel_err(tr.last["emb"].cpu().numpy(), z["emb"]) < TOL
    assert rel_err(tr.last["feat"][:, :342].cpu().numpy(), z["feat"]) < TOL
    tr.backward(demb=torch.as_tensor(demb, dtype=torch.float32).cuda())
    g = tr.plan.export_params(tr.grad)
    assert rel_err(g["conv1_weight"].cpu().numpy(), z["grad_conv1_weight"]) < TOL
    sums = np.array([float(g[k].abs().sum()) for k in sorted(k for k in g if k != "head_weight")])
    assert np.abs(sums / z["grad_abs_sums"] - 1).max() < 5e-3


def test_fused_plan_is_bitwise_equal_to_unfused():
    """The fused conv+MFM(+pool) epilogues change where bytes travel, not a single bit of the result: loss, embeddings
    and the whole flat gradient of a step are identical with fuse=True / fuse=False (3x112x112: includes 7 -> 3 pooling)."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    res = []
    for fuse in (True, False):
        tr = TripletTrainer(8, image=112, seed=5, fuse=fuse)
        assert (tr.plan.fused == 20) if fuse else (tr.plan.fused == 0)
        x = synth.images(8, 3, 112, 21)
        neg = synth.negative_indices(synth.parity_labels(8, images_per_identity=2), 9).cuda()
        loss = tr.forward_loss(x, neg).clone()
        tr.backward()
        res.append((loss, tr.last["emb"].clone(), tr.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*res))


def test_config1_64_faces_vs_committed_golden():
    """BASELINE configs[0]: EFM-29 on 64 synthetic 112x112x3 faces, embeddings + triplet loss + the cosine rows train_efm.py logs,
    against tests/golden/config1_efm112.npz (torch-CPU fp64 restatement; no oracle call at test time)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_config1_golden as M
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "config1_efm112.npz"))
    params, w_head, x, neg = M.inputs()
    tr = TripletTrainer(M.BATCH, image=M.IMAGE)
    allp = dict(params)
    allp["head_weight"] = w_head
    tr.plan.load_params(tr.flat, allp)
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), torch.as_tensor(neg).cuda())
    emb = tr.last["emb"]
    e_emb = rel_err(emb.cpu().numpy(), z["emb"])
    assert e_emb < TOL, e_emb
    assert rel_err(loss.cpu().numpy(), z["loss"]) < TOL
    feat = tr.last["feat"][:, :342].double().cpu().numpy()
    assert np.abs(feat.sum(1) / z["feat_sum"] - 1).max() < TOL
    assert np.abs(np.abs(feat).sum(1) / z["feat_abs"] - 1).max() < TOL
    h = M.BATCH // 2
    s_ap, s_an = ops.cosine_pairs(emb[:h], emb[h:], emb[:h][torch.as_tensor(neg).long().cuda()].contiguous())
    assert rel_err(torch.stack([s_ap, s_an], 1).cpu().numpy(), z["cosines"]) < TOL
    # the part of the loss that is not the margin (d_ap - d_an, ~1e-3 of it on an untrained net) still agrees to 1e-3 of its own scale
    dl, dr = loss.cpu().numpy() - 0.2, z["loss"] - 0.2
    print("config-1 golden: emb %.2e, (loss - margin) %.2e of %.2e" % (e_emb, np.abs(dl - dr).max(), np.abs(dr).max()))
    assert np.abs(dl - dr).max() < 1e-3 * np.abs(dr).max() + 1e-7


def test_full_size_properties():
    """BASELINE configs[1] at its full size (256 images of 3x112x112): properties that need no oracle run.
    (a) a batch permutation permutes the embeddings bit for bit (no result depends on the position of an image in the batch);
    (b) the parameter gradient of a step is bitwise reproducible and equals the sum of the gradients of its two half-batch shards
        with the same negatives (the data-parallel contract), to fp32 summation noise;
    (c) the weight gradient is linear in the upstream gradient."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    batch, image = 256, 112
    tr = TripletTrainer(batch, image=image, seed=42)
    x = synth.images(batch, 3, image, 1234)
    perm = torch.randperm(batch, generator=torch.Generator().manual_seed(0)).cuda()
    emb, _ = tr.plan.forward(x, tr.flat, train=False)
    emb = emb.clone()
    emb_p, _ = tr.plan.forward(x[perm].contiguous(), tr.flat, train=False)
    assert torch.equal(emb_p, emb[perm])
    assert torch.isfinite(emb).all() and float((emb.norm(dim=1) - 0).abs().min()) > 0
    demb = (synth.uniform01(batch * emb.shape[1], 77).view(batch, -1) * 2 - 1)[:, :emb.shape[1]].contiguous()
    neg = synth.negative_indices(synth.parity_labels(batch), 5).cuda()

    def grad_of(scale):
        tr.forward_loss(x, neg)
        tr.backward(demb=demb * scale)
        return tr.grad.clone()
    g1 = grad_of(1.0)
    assert torch.equal(g1, grad_of(1.0))                                   # (b) reproducible
    g2 = grad_of(2.0)
    assert rel_err(g2.cpu().numpy(), 2 * g1.cpu().numpy()) < 1e-5          # (c) linear in the upstream gradient
    # (b) shards: two 128-image trainers on the halves, upstream gradients of their rows
    half = batch // 2
    tr_h = TripletTrainer(half, image=image, seed=42)
    tr_h.flat.copy_(tr.flat)
    gsum = torch.zeros_like(g1)
    neg_h = synth.negative_indices(synth.parity_labels(half), 5).cuda()
    for s in range(2):
        rows = slice(s * half, (s + 1) * half)
        tr_h.forward_loss(x[rows].contiguous(), neg_h)
        tr_h.backward(demb=demb[rows].contiguous())
        gsum += tr_h.grad
    assert rel_err(gsum.cpu().numpy(), g1.cpu().numpy()) < 1e-4


def test_training_separates_identities():
    """The whole step actually learns: EFM-29 at 112x112 on identity-structured synthetic faces (reference batch layout, random
    negatives of another identity), autotuned kernels (Winograd where faster): after 100 SGD steps the anchor-positive cosine stays
    ~1 while the anchor-negative cosine has dropped, the loss has fallen under a quarter of the margin, all weights are finite."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    batch, image = 64, 112
    tr = TripletTrainer(batch, image=image, optimizer="sgd", lr=0.05, wd=1e-5, margin=0.2, autotune=True)
    rng = np.random.default_rng(0)
    h = batch // 2
    losses = []
    for step in range(1, 101):
        ids = rng.choice(256, size=h, replace=False)
        x = torch.cat([synth.identity_faces(ids, 3, image, 2 * step, 0.25), synth.identity_faces(ids, 3, image, 2 * step + 1, 0.25)]).contiguous()
        neg = torch.as_tensor(((np.arange(h) + rng.integers(1, h, size=h)) % h).astype(np.int32)).cuda()
        losses.append(float(tr.step(x, neg).mean()))
    s_ap, s_an = tr.cosine_log()
    print("training sanity: loss %.4f -> %.4f, s_ap %.4f, s_an %.4f" % (losses[0], np.mean(losses[-10:]), float(s_ap.mean()), float(s_an.mean())))
    assert np.mean(losses[-10:]) < 0.05 < losses[0]
    assert float(s_ap.mean()) - float(s_an.mean()) > 0.1
    assert bool(torch.isfinite(tr.flat).all())
