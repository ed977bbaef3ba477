"""BASELINE configs[2] as a parity case: LightCNN-9 (2-way MFM), 256-d embedding, in-batch semi-hard mining — fp32 here
(the bf16 MFMA path is a later round).  The network is build-defined: the oracle is the torch-CPU
restatement in oracle/efm_oracle_torch.py, parity unpinned."""
import numpy as np
import pytest
import torch

from oracle import efm_oracle as O
from oracle import efm_oracle_torch as OT
from tests.util import dev, rand, rel_err

pytestmark = pytest.mark.gpu


def test_indexed_triplet_kernels():
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    rows, d = 48, 256
    e = rand((rows, d), 1, 0.3)
    labels = (np.arange(rows) // 4).astype(np.int32)
    pos, inv = O.mining_indices(labels)
    g = O.gram_cosine(e)
    neg = O.mine_semihard(g, labels, np.arange(rows), pos)
    neg[5] = -1  # a row without a negative contributes nothing
    ed = dev(e)
    t = lambda a: torch.as_tensor(a.astype(np.int32)).cuda()  # noqa: E731
    loss = ops.triplet_indexed_fwd(ed, t(pos), t(neg), 0.2)
    lr = O.triplet_indexed(e, pos, neg, 0.2)
    assert (lr > 0).any() and rel_err(loss.cpu().numpy(), lr) < 1e-5
    gl = rand((rows,), 2)
    de = ops.triplet_indexed_bwd(ed, t(pos), t(neg), t(inv), loss, dev(gl))
    assert rel_err(de.cpu().numpy(), O.triplet_indexed_bwd(e, pos, neg, lr, gl)) < 1e-5


def test_lightcnn9_structure_and_flops():
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    plan = Plan(efm_symbol.lightcnn9_embedding_net(), (2, 3, 112, 112))
    convs = [s for s in plan.steps if s.op == "conv"]
    assert len(convs) == 10 and plan.fused == 10  # 9 convolutions + fc1, every conv -> MFM2 [-> pool] chain fused
    assert plan.outputs[0].shape == (256, 1, 1)
    assert plan.flops_fwd // 2 == 1616068608  # 1.616 GFLOP / image forward


def test_lightcnn9_semihard_step_vs_oracle():
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import MiningTripletTrainer
    batch, image = 16, 32
    tr = MiningTripletTrainer(batch, image=image, outputs=efm_symbol.lightcnn9_embedding_net(), seed=7)
    labels = (np.arange(batch) // 4).astype(np.int32)
    tr.set_labels(labels)
    params = {k: v.cpu().numpy().astype(np.float64) for k, v in tr.plan.export_params(tr.flat).items()}
    x = O.uniform01(batch * 3 * image * image, 5).reshape(batch, 3, image, image)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    pos, _ = O.mining_indices(labels)
    loss_r, emb_r, neg_r = OT.mining_step(OT.lightcnn9_forward, tp, torch.tensor(x), labels, pos, 0.2)
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda())
    assert rel_err(tr.last["emb"].cpu().numpy(), emb_r.numpy()) < 1e-3
    assert rel_err(tr.last["gram"].cpu().numpy(), O.gram_cosine(emb_r.numpy())) < 1e-3
    # the mined negatives agree wherever the oracle's choice is not a near-tie in distance
    neg = tr.last["neg"].cpu().numpy()
    d = 1.0 - O.gram_cosine(emb_r.numpy())
    for i in np.nonzero(neg != neg_r)[0]:
        assert abs(d[i, neg[i]] - d[i, neg_r[i]]) < 1e-4
    # loss / gradients with the oracle's negatives forced (removes the discrete choice from the comparison)
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), neg_idx=torch.as_tensor(neg_r.astype(np.int32)).cuda())
    assert rel_err(loss.cpu().numpy(), loss_r.numpy()) < 1e-3
    tr.backward()
    g = tr.plan.export_params(tr.grad)
    worst = max(rel_err(g[k].cpu().numpy().reshape(tp[k].shape), tp[k].grad.numpy()) for k in tp)
    assert worst < 2e-2, worst  # fp32 arg-max route flips through 10 MFM/pool stages (see test_e2e_gpu.py); forward is 1e-3
    tr.update()


def test_deepcnn_structure_and_step_vs_oracle():
    """BASELINE configs[4]'s build-defined deeper CNN (512-d) in fp32: structure, then one semi-hard step against the torch oracle."""
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import MiningTripletTrainer
    plan = Plan(efm_symbol.deepcnn_embedding_net(), (2, 3, 112, 112))
    assert len([s for s in plan.steps if s.op == "conv"]) == 14 and plan.fused == 14
    assert plan.outputs[0].shape == (512, 1, 1)
    assert plan.flops_fwd // 2 == 5199839232
    batch, image = 8, 32
    tr = MiningTripletTrainer(batch, image=image, outputs=efm_symbol.deepcnn_embedding_net(), seed=11)
    labels = (np.arange(batch) // 2).astype(np.int32)
    tr.set_labels(labels)
    params = {k: v.cpu().numpy().astype(np.float64) for k, v in tr.plan.export_params(tr.flat).items()}
    x = O.uniform01(batch * 3 * image * image, 6).reshape(batch, 3, image, image)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    pos, _ = O.mining_indices(labels)
    loss_r, emb_r, neg_r = OT.mining_step(OT.deepcnn_forward, tp, torch.tensor(x), labels, pos, 0.2)
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), neg_idx=torch.as_tensor(neg_r.astype(np.int32)).cuda())
    assert rel_err(tr.last["emb"].cpu().numpy(), emb_r.numpy()) < 1e-3
    assert rel_err(loss.cpu().numpy(), loss_r.numpy()) < 1e-3
    tr.backward()
    g = tr.plan.export_params(tr.grad)
    worst = max(rel_err(g[k].cpu().numpy().reshape(tp[k].shape), tp[k].grad.numpy()) for k in tp)
    assert worst < 3e-2, worst  # fp32 arg-max route flips through 14 MFM/pool stages; forward is 1e-3
    tr.update()
