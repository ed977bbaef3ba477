"""Winograd F(2x2, 3x3) kernels (efm_wino_*) against the NumPy fp64 oracle and the direct implicit-GEMM kernels.
Tolerance: 2e-4 of the largest reference magnitude like every conv kernel test (measured ~1e-6: the transforms only add / halve)."""
import numpy as np
import pytest
import torch

from oracle import efm_oracle as O
from tests.util import dev, from_nhwc, rand, rel_err, to_nhwc

pytestmark = pytest.mark.gpu
TOL = 2e-4

# (batch, h, w, cin, cout): even / odd maps (7 -> 4 tiles with a half-empty last one), channel counts off the 4 / 8 / 16 grids,
# 1..3 channel blocks, tile counts that are not a multiple of 32
CASES = [(2, 8, 8, 8, 16), (3, 14, 14, 44, 99), (2, 7, 7, 174, 261), (1, 28, 28, 66, 66), (2, 6, 10, 5, 35), (1, 56, 56, 12, 198),
         (5, 7, 5, 258, 387)]


@pytest.mark.parametrize("case", CASES)
def test_wino_fwd_and_dgrad(case):
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w, cin, cout = case
    x = rand((b, cin, h, w), 1)
    wt = rand((cout, cin, 3, 3), 2, 0.2)
    bias = rand((cout,), 3)
    d = ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1)
    assert ops.wino_supported(d)
    xd = to_nhwc(x)
    wp = ops.conv_pack_weights(d, dev(wt))
    bp = torch.zeros(d.n_pad16, device="cuda")
    bp[:cout] = dev(bias)
    ref = O.conv2d(x, wt, bias, (1, 1))
    u = ops.wino_make_u(d, wp)
    y = ops.wino_fwd(d, xd, u, bp)
    assert rel_err(from_nhwc(y, cout), ref) < TOL
    if d.cout_p > cout:
        assert float(y[..., cout:].abs().max()) == 0.0      # pad channels stay zero
    assert rel_err(y.cpu().numpy(), ops.conv_fwd(d, xd, wp, bp).cpu().numpy()) < 1e-5   # vs the direct kernel
    res = rand(ref.shape, 4)
    y2 = ops.wino_fwd(d, xd, u, bp, residual=to_nhwc(res))
    assert rel_err(from_nhwc(y2, cout), ref + res) < TOL
    # data gradient = the same kernel on dy with U made from the tap-flipped transpose of the same packed weights
    dy = rand(ref.shape, 5)
    dx_ref, _, _ = O.conv2d_bwd(x, wt, dy, (1, 1))
    dyd = to_nhwc(dy)
    ud = ops.wino_make_u(d, wp, dgrad=True)
    dx = ops.wino_bwd_data(d, dyd, ud)
    assert rel_err(from_nhwc(dx, cin), dx_ref) < TOL
    if d.cin_p > cin:
        assert float(dx[..., cin:].abs().max()) == 0.0
    add = rand(x.shape, 6)
    dx2 = ops.wino_bwd_data(d, dyd, ud, add=to_nhwc(add))
    assert rel_err(from_nhwc(dx2, cin), dx_ref + add) < TOL


def test_wino_rejects_other_geometries():
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    assert not ops.wino_supported(ops.conv_desc(1, 8, 8, 4, 4, 1, 1, 0, 0))
    assert not ops.wino_supported(ops.conv_desc(1, 8, 8, 4, 4, 5, 5, 2, 2))
    assert not ops.wino_supported(ops.conv_desc(1, 8, 8, 4, 4, 3, 3, 0, 0))
    d = ops.conv_desc(1, 8, 8, 4, 4, 5, 5, 2, 2)
    with pytest.raises(Exception):
        ops.wino_make_u(d, torch.zeros(d.n_pad16 * d.k_pad, device="cuda"))


def test_plan_with_winograd_layers_matches_direct_plan(monkeypatch):
    """EFM-29 step with every eligible plain forward / data gradient on the Winograd kernels (autotune, forced) against the same
    step on the direct kernels: embeddings, loss and all gradients agree to fp32 rounding."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    batch, image = 8, 64
    x = synth.images(batch, 3, image, 11)
    labels = synth.parity_labels(batch, images_per_identity=2)
    neg = synth.negative_indices(labels, 3).cuda()
    out = {}
    for mode in ("direct", "winograd"):
        monkeypatch.setenv("EFM_WINO", "force" if mode == "winograd" else "0")
        tr = TripletTrainer(batch, image=image, seed=5, autotune=True)
        nw = sum(bool(getattr(s, "wino_fwd", False)) + bool(getattr(s, "wino_dgrad", False)) for s in tr.plan.steps)
        assert (nw > 20) if mode == "winograd" else (nw == 0)
        loss = tr.forward_loss(x, neg)
        tr.backward()
        out[mode] = (tr.last["emb"].clone(), loss.clone(), tr.grad.clone())
    assert rel_err(out["winograd"][0].cpu().numpy(), out["direct"][0].cpu().numpy()) < 1e-5
    assert rel_err(out["winograd"][1].cpu().numpy(), out["direct"][1].cpu().numpy()) < 1e-5
    # gradients go through max / min / pool routes, and a last-bit difference in a forward value flips some: any two correct fp32
    # implementations differ by ~6e-3 there (measured in test_e2e_gpu.py::test_112_step_vs_oracles); a wiring error would be O(1).
    # The kernels themselves are compared tightly above.
    gw, gd = out["winograd"][2].double(), out["direct"][2].double()
    assert rel_err(gw.cpu().numpy(), gd.cpu().numpy()) < 6e-2
    assert float((gw * gd).sum() / (gw.norm() * gd.norm())) > 0.9995


def test_inference_forward_between_training_forward_and_backward(monkeypatch):
    """An evaluation forward on the same plan between a training forward and its backward (same-batch validation inside a step)
    must not disturb the backward: the saved activations, route bytes and the data-gradient U of the TRAINING forward stay in
    place (the weights have not changed).  Gradients bit-identical to forward -> backward."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    monkeypatch.setenv("EFM_WINO", "force")
    batch, image = 8, 32
    x, x2 = synth.images(batch, 3, image, 11), synth.images(batch, 3, image, 12)
    neg = synth.negative_indices(synth.parity_labels(batch, images_per_identity=2), 3).cuda()
    tr = TripletTrainer(batch, image=image, seed=5, autotune=True)
    assert any(getattr(s, "wino_dgrad", False) for s in tr.plan.steps)
    tr.forward_loss(x, neg)
    tr.backward()
    g_ref = tr.grad.clone()
    tr.forward_loss(x, neg)
    emb_eval = tr.plan.forward(x2, tr.flat, train=False)[0].clone()     # evaluation of other images in the middle of the step
    tr.backward()
    assert torch.equal(tr.grad, g_ref)
    assert torch.equal(tr.plan.forward(x2, tr.flat, train=False)[0], emb_eval)


# (batch, h, w, cin, cout, ways, pool): MFM3 / MFM2, pooled (a 2x2 tile is the pooling window; 7 -> 3 floor pooling drops the half tile)
# and unpooled, one and several channel blocks, tile counts off the 64-tile grid
FUSED = [(2, 8, 8, 8, 18, 3, True), (3, 14, 14, 44, 99, 3, False), (2, 7, 7, 58, 261, 3, True), (1, 28, 28, 24, 198, 3, True),
         (2, 6, 10, 5, 34, 2, True), (3, 7, 5, 20, 96, 2, False), (1, 14, 14, 40, 387, 3, False), (2, 16, 16, 16, 256, 2, True)]


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("case", FUSED)
def test_wino_fused_epilogue(case, variant):
    """efm_wino_mfm_fwd = efm_wino_fwd followed by the stand-alone MFM / pooling kernels, BITWISE (same arithmetic, same tie rules),
    and its route bytes drive efm_mfm_pool_bwd to the same gradient as the unfused backward chain."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w, cin, cout, ways, pool = case
    x = rand((b, cin, h, w), 31)
    wt = rand((cout, cin, 3, 3), 32, 0.2)
    bias = rand((cout,), 33)
    d = ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1)
    d.tune_fwd = variant << 8   # 1 = 8-wave kernel, 2 = 4-wave kernel
    xd = to_nhwc(x)
    wp = ops.conv_pack_weights(d, dev(wt))
    bp = torch.zeros(d.n_pad16, device="cuda")
    bp[:cout] = dev(bias)
    y_ref = O.conv2d(x, wt, bias, (1, 1))
    z_ref = O.mfm3(y_ref) if ways == 3 else O.mfm2(y_ref)
    if pool:
        z_ref = O.maxpool2(z_ref)
    co = z_ref.shape[1]
    y = ops.wino_fwd(d, xd, ops.wino_make_u(d, wp), bp)
    for order in (O.ORDER_GROUP, O.ORDER_RES):
        z, route = ops.wino_mfm_fwd(d, xd, ops.wino_mfm_make_u(d, wp, ways), bp, ways, order, pool)
        assert rel_err(from_nhwc(z, co), z_ref) < TOL
        mf = ops.mfm_fwd(y, cout, ways)
        zu = ops.maxpool2_fwd(mf, co) if pool else mf
        assert torch.equal(z, zu)
        dzd = to_nhwc(rand(z_ref.shape, 34))
        dyf = ops.mfm_pool_bwd(d, route, dzd, ways, pool)
        dmf = ops.maxpool2_bwd(mf, dzd, co) if pool else dzd
        assert torch.equal(dyf, ops.mfm_bwd(y, dmf, cout, ways, order))


def test_wino_fused_pool_ties_pick_first_maximum():
    """Constant input and weights: all four pixels of interior windows tie, and the slices tie too — the route must be the FIRST
    window pixel and the lhs slice, as the direct kernel and MXNet's pooling / maximum backward choose."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w, cin, cout = 1, 8, 8, 4, 48
    d = ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1)
    xd = torch.ones((b, h, w, d.cin_p), device="cuda")
    wt = np.zeros((cout, cin, 3, 3))
    wt[:, :, 1, 1] = 0.25          # centre tap only: every output pixel = 1.0 exactly, whatever the border
    wp = ops.conv_pack_weights(d, dev(wt))
    bp = torch.zeros(d.n_pad16, device="cuda")
    for variant in (1, 2):
        d.tune_fwd = variant << 8
        u = ops.wino_mfm_make_u(d, wp, 3)
        for order in (O.ORDER_GROUP, O.ORDER_RES):
            z, route = ops.wino_mfm_fwd(d, xd, u, bp, 3, order, True)
            zd, rd = ops.conv_mfm_fwd(ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1), xd, wp, bp, 3, order, True)
            assert torch.equal(z, zd) and torch.equal(route, rd)


@pytest.mark.parametrize("case", CASES)
def test_wino_wgrad(case):
    """Winograd weight gradient (+ bias gradient) against the fp64 oracle and the direct kernel; accumulate mode; the packed
    gradient's pads are exact zeros; bitwise reproducible."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w, cin, cout = case
    x = rand((b, cin, h, w), 1)
    wt = rand((cout, cin, 3, 3), 2, 0.2)
    dy = rand((b, cout, h, w), 5)
    d = ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1)
    xd, dyd = to_nhwc(x), to_nhwc(dy)
    _, dw_ref, db_ref = O.conv2d_bwd(x, wt, dy, (1, 1))
    dw, db = ops.wino_bwd_weight(d, xd, dyd)
    assert rel_err(ops.conv_unpack_weights(d, dw).cpu().numpy(), dw_ref) < TOL
    assert rel_err(db[:cout].cpu().numpy(), db_ref) < TOL
    dwd, dbd = ops.conv_bwd_weight(d, xd, dyd)
    assert rel_err(dw.cpu().numpy(), dwd.cpu().numpy()) < 1e-5 and rel_err(db.cpu().numpy(), dbd.cpu().numpy()) < 1e-5
    dwm = dw.clone()
    ops.conv_pack_weights_into(d, ops.conv_unpack_weights(d, dw), dwm)
    assert torch.equal(dwm, dw)                       # pad rows / columns are exactly zero
    dw2, db2 = ops.wino_bwd_weight(d, xd, dyd)
    assert torch.equal(dw2, dw) and torch.equal(db2, db)
    acc_w, acc_b = dw.clone(), db.clone()
    ops.wino_bwd_weight(d, xd, dyd, dw=acc_w, dbias=acc_b, accumulate=True)
    assert rel_err(acc_w.cpu().numpy(), 2 * dw.cpu().numpy()) < 1e-6 and rel_err(acc_b.cpu().numpy(), 2 * db.cpu().numpy()) < 1e-6


WGRAD_SHAPES = [(6, 4), (4, 6), (7, 3), (5, 4), (4, 5), (6, 3), (5, 3), (3, 5), (4, 4)]   # kShapes of efm_wino_wgrad.hip, in its order


@pytest.mark.parametrize("shape", range(len(WGRAD_SHAPES) + 1))
def test_wino_wgrad_behind_conv_bwd_weight(shape):
    """tune_wgrad bit 12 routes efm_conv_bwd_weight{,_slabs,_finish} / efm_conv_wgrad_workspace_bytes to the Winograd form (bits 3:0 =
    1 + block shape, 0 = least padding; bits 9:4 = blocks / 64): every block shape against the fp64 oracle on maps whose chunks have
    4, 3, 2 and 1 k steps and partial channel blocks; the two-launch protocol equals the single call bit for bit; efm_conv_kernel_info
    names the instance the launch resolves to."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    for (b, h, w, cin, cout) in [(3, 14, 14, 44, 99), (2, 7, 5, 70, 35), (1, 18, 22, 12, 120)]:
        x = rand((b, cin, h, w), 11)
        wt = rand((cout, cin, 3, 3), 12, 0.2)
        dy = rand((b, cout, h, w), 15)
        _, dw_ref, db_ref = O.conv2d_bwd(x, wt, dy, (1, 1))
        xd, dyd = to_nhwc(x), to_nhwc(dy)
        d = ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1)
        d.tune_wgrad = 0x1000 | (4 << 4) | shape
        name, flops = ops.conv_kernel_info(d, ops.PASS_WGRAD)
        assert name.startswith("wino_wgrad_k<") and flops > 0
        if shape:
            assert name == "wino_wgrad_k<%d, %d, 4>" % WGRAD_SHAPES[shape - 1]
        dw, db = ops.conv_bwd_weight(d, xd, dyd)
        assert rel_err(ops.conv_unpack_weights(d, dw).cpu().numpy(), dw_ref) < TOL
        assert rel_err(db[:cout].cpu().numpy(), db_ref) < TOL
        ws = torch.empty(ops.conv_wgrad_workspace_bytes(d) // 4 + 16, device="cuda")
        dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
        ops.conv_bwd_weight_slabs(d, xd, dyd, ws)
        ops.conv_bwd_weight_finish(d, ws, dw2, db2)
        assert torch.equal(dw2, dw) and torch.equal(db2, db)
        ops.conv_bwd_weight_finish(d, ws, dw2, db2, accumulate=True)
        assert rel_err(dw2.cpu().numpy(), 2 * dw.cpu().numpy()) < 1e-6
        dwm = dw.clone()
        ops.conv_pack_weights_into(d, ops.conv_unpack_weights(d, dw), dwm)
        assert torch.equal(dwm, dw)                       # pad rows / columns are exactly zero
        d.tune_wgrad = 0
        assert ops.conv_kernel_info(d, ops.PASS_WGRAD)[0].startswith("conv_wgrad_k<")


def test_wino_wgrad_random_geometries():
    """Seeded sweep of odd geometries (1..3 images, 1..23 pixels per side, 1..150 channels) through every block shape: the Winograd
    weight gradient (+ bias gradient) against the direct kernel on the same device tensors, pads exactly zero."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    rng = np.random.default_rng(20261005)
    for trial in range(24):
        b, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 24)), int(rng.integers(1, 24))
        cin, cout = int(rng.integers(1, 151)), int(rng.integers(1, 151))
        d = ops.conv_desc(b, h, w, cin, cout, 3, 3, 1, 1)
        xd = torch.zeros((b, h, w, d.cin_p), device="cuda")
        xd[..., :cin] = torch.rand((b, h, w, cin), device="cuda") - 0.5
        dyd = torch.zeros((b, h, w, d.cout_p), device="cuda")
        dyd[..., :cout] = torch.rand((b, h, w, cout), device="cuda") - 0.5
        dwd, dbd = ops.conv_bwd_weight(d, xd, dyd)
        scale = float(dwd.abs().max()) + 1e-30
        d.tune_wgrad = 0x1000 | (4 << 4) | (trial % (len(WGRAD_SHAPES) + 1))
        dw, db = ops.conv_bwd_weight(d, xd, dyd)
        assert float((dw - dwd).abs().max()) / scale < 1e-5, (trial, b, h, w, cin, cout)
        assert float((db - dbd).abs().max()) / (float(dbd.abs().max()) + 1e-30) < 1e-5, (trial, b, h, w, cin, cout)
        dwm = dw.clone()
        ops.conv_pack_weights_into(d, ops.conv_unpack_weights(d, dw), dwm)
        assert torch.equal(dwm, dw), (trial, b, h, w, cin, cout)
