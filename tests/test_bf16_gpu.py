"""bf16 tensor-core path (BASELINE configs[2]) against a CPU emulation of the same arithmetic: operands rounded to bf16
(round-to-nearest-even), products and sums in fp32.  bf16 x bf16 products are exact in fp32, so the kernels must match the
emulation to accumulation-order noise (1e-5), not to bf16 noise."""
import numpy as np
import pytest
import torch

from oracle import efm_oracle as O
from tests.util import dev, rand, rel_err

pytestmark = pytest.mark.gpu


def bf(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).bfloat16().float().numpy().astype(np.float64)


def to_nhwc_bf16(ops, x):
    return ops.nchw_to_nhwc_bf16(dev(x))


def from_nhwc(t, c):
    return t.float().cpu().numpy()[..., :c].transpose(0, 3, 1, 2).astype(np.float64)


CASES = [
    # batch, h, w, cin, cout, k, pad
    (2, 12, 10, 3, 96, 5, 2),
    (2, 9, 11, 48, 96, 1, 0),
    (3, 8, 8, 48, 192, 3, 1),
    (2, 6, 6, 96, 384, 3, 1),
    (2, 7, 7, 128, 256, 3, 1),
    (2, 5, 5, 66, 99, 3, 1),     # channel counts that are no multiple of 8
    # the halo-tile weight gradient (efm_convb_wgrad.hip): maps that cross its 4 x 16 stage tiles raggedly, one / three tap groups,
    # one / two n parts, both wave-tile shapes
    (2, 21, 37, 48, 192, 3, 1),
    (1, 14, 14, 192, 256, 3, 1),
    (3, 5, 19, 96, 384, 3, 1),
    (2, 13, 9, 128, 256, 1, 0),
    (1, 30, 18, 16, 96, 3, 1),
]


@pytest.mark.parametrize("case", CASES)
def test_convb_fwd_dgrad_wgrad(case):
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w, cin, cout, k, pad = case
    x, wt, bias = bf(rand((b, cin, h, w), 1)), rand((cout, cin, k, k), 2, 0.2), rand((cout,), 3)
    d = ops.conv_desc(b, h, w, cin, cout, k, k, pad, pad)
    xd = to_nhwc_bf16(ops, x)
    assert rel_err(from_nhwc(xd, cin), x) == 0.0 and float(xd[..., cin:].float().abs().max() if xd.shape[-1] > cin else 0) == 0.0
    wp = ops.conv_pack_weights(d, dev(wt))
    wb, wdb = ops.convb_cast_weights(d, wp)
    bp = torch.zeros(d.n_pad16, device="cuda")
    bp[:cout] = dev(bias)
    ref = O.conv2d(x, bf(wt), bias, (pad, pad))
    y = ops.convb_fwd(d, xd, wb, bp)
    assert rel_err(from_nhwc(y, cout), bf(ref)) < 4e-3          # output storage rounding to bf16 (2^-8 relative)
    # backward with bf16 dy
    dy = bf(rand(ref.shape, 5))
    dyd = to_nhwc_bf16(ops, dy)
    dx_ref, dw_ref, db_ref = O.conv2d_bwd(x, bf(wt), dy, (pad, pad))
    dx = ops.convb_bwd_data(d, dyd, wdb)
    assert rel_err(from_nhwc(dx, cin), bf(dx_ref)) < 4e-3
    dw, db = ops.convb_bwd_weight(d, xd, dyd)
    assert rel_err(ops.conv_unpack_weights(d, dw).cpu().numpy(), dw_ref) < 2e-5   # fp32 output: only summation order differs
    assert rel_err(db[:cout].cpu().numpy(), db_ref) < 2e-5
    dw2, db2 = ops.convb_bwd_weight(d, xd, dyd)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)          # fixed-order reductions: bitwise reproducible
    dwm = dw.clone()
    ops.conv_pack_weights_into(d, ops.conv_unpack_weights(d, dw), dwm)
    assert torch.equal(dwm, dw)  # pad rows / columns of the packed gradient are exactly zero


@pytest.mark.parametrize("ways,pool,out_f32", [(2, True, False), (2, False, False), (3, True, True), (2, True, True)])
def test_convb_fused_epilogue(ways, pool, out_f32):
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w, cin, cout, k, pad = 2, 8, 8, 48, 96, 3, 1
    x, wt, bias = bf(rand((b, cin, h, w), 11)), rand((cout, cin, k, k), 12, 0.2), rand((cout,), 13)
    d = ops.conv_desc(b, h, w, cin, cout, k, k, pad, pad)
    xd = to_nhwc_bf16(ops, x)
    wb, _ = ops.convb_cast_weights(d, ops.conv_pack_weights(d, dev(wt)))
    bp = torch.zeros(d.n_pad16, device="cuda")
    bp[:cout] = dev(bias)
    y_ref = O.conv2d(x, bf(wt), bias, (pad, pad))
    z_ref = O.mfm3(y_ref) if ways == 3 else O.mfm2(y_ref)
    mf_ref = z_ref
    if pool:
        z_ref = O.maxpool2(z_ref)
    co = z_ref.shape[1]
    z, route = ops.convb_mfm_fwd(d, xd, wb, bp, ways, O.ORDER_GROUP, pool, out_f32)
    assert z.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    assert rel_err(from_nhwc(z, co), z_ref if out_f32 else bf(z_ref)) < (2e-5 if out_f32 else 4e-3)
    # backward of the epilogue: bf16 dy, exact scatter of the (bf16-rounded) upstream gradient
    dz = rand(z_ref.shape, 14) if out_f32 else bf(rand(z_ref.shape, 14))
    dzd = torch.zeros(z.shape, dtype=z.dtype, device="cuda")
    dzd[..., :co] = torch.as_tensor(dz.transpose(0, 2, 3, 1)).to(z.dtype).cuda()
    dy = ops.convb_mfm_pool_bwd(d, route, dzd, ways, pool)
    dmf = O.maxpool2_bwd(mf_ref, dz) if pool else dz
    dy_ref = O.mfm3_bwd(y_ref, dmf, O.ORDER_GROUP) if ways == 3 else O.mfm2_bwd(y_ref, dmf)
    assert rel_err(from_nhwc(dy, cout), bf(dy_ref)) < 1e-6


@pytest.mark.parametrize("net,batch,image", [("lightcnn9", 16, 32), ("deepcnn", 16, 32), ("lightcnn9", 8, 112), ("deepcnn", 8, 112)])
def test_mfm2_stack_bf16_step_vs_emulation(net, batch, image):
    """Whole mining step in bf16 — LightCNN-9 (BASELINE configs[2]) and the deeper 512-d CNN (configs[4]) — against the torch-CPU
    emulation of the same rounding points (fp64 in between).  Free-running, device and emulation decorrelate to the bf16 noise
    level after a few layers (see the oracle's docstring), so the tight comparison is teacher-forced: every emulated layer
    starts from the device's stored activation.  Run at 32x32 and at the BASELINE geometry 3x112x112 (every real grid shape of the
    layers, the 7x7x128 -> 512 fc, floor pooling 7 -> 3 does not occur: 112 -> 56 -> 28 -> 14 -> 7)."""
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import MiningTripletTrainer
    from oracle import efm_oracle_torch as OT
    outputs, fwd = ((efm_symbol.lightcnn9_embedding_net(), OT.lightcnn9_forward_bf16) if net == "lightcnn9"
                    else (efm_symbol.deepcnn_embedding_net(), OT.deepcnn_forward_bf16))
    tr = MiningTripletTrainer(batch, image=image, outputs=outputs, seed=7, dtype="bf16")
    labels = (np.arange(batch) // 4).astype(np.int32)
    tr.set_labels(labels)
    params = {k: v.cpu().numpy().astype(np.float64) for k, v in tr.plan.export_params(tr.flat).items()}
    x = O.uniform01(batch * 3 * image * image, 5).reshape(batch, 3, image, image)
    pos, _ = O.mining_indices(labels)

    # (1) free-running emulation: mined negatives, loss and embeddings at the bf16 noise level
    tp = {k: torch.tensor(v) for k, v in params.items()}
    with torch.no_grad():
        loss_r, emb_r, neg_r = OT.mining_step(fwd, tp, torch.tensor(x), labels, pos, 0.2, backward=False)
    neg = torch.as_tensor(neg_r.astype(np.int32)).cuda()
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), neg_idx=neg)
    emb_d = tr.last["emb"]
    assert emb_d.dtype == torch.float32
    e_free = rel_err(emb_d.cpu().numpy(), emb_r.numpy())
    assert e_free < 2e-2, e_free
    assert rel_err(loss.cpu().numpy(), loss_r.numpy()) < 2e-2

    # (2) teacher-forced: per-layer outputs equal up to isolated one-ulp roundings; feature vector to fp32 accumulation noise
    convs = [st for st in tr.plan.steps if st.op == "conv"]
    dev, chans = [], []
    for st in convs:
        t = tr.plan._acts[st.index]
        d = st.desc
        dev.append(t.reshape(d.batch, t.shape[1] if t.dim() == 4 else 1, t.shape[2] if t.dim() == 4 else 1, -1))
        chans.append(d.cout // 2)
    nl = len(convs) - 1  # all but fc1 hand their stored activation to the next layer
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    forced = [dev[i][..., :chans[i]].permute(0, 3, 1, 2).double().cpu() for i in range(nl)]
    outs = []
    feat = fwd(tp, torch.tensor(x), forced=forced, outs=outs)
    for i in range(nl):
        a, b = forced[i].numpy(), outs[i].numpy()
        assert a.shape == b.shape
        diff = np.abs(a - b)
        # one bf16 ulp of the value, or — where a sum cancels to ~0 — the fp32 accumulation noise of its terms
        assert (diff <= np.abs(b) * 2.0 ** -7 + 1e-6 * np.abs(b).max()).all(), (i, diff.max())
        assert (diff > 0).mean() < 5e-3, (i, (diff > 0).mean())
    feat_d = dev[nl].reshape(batch, -1)[:, :feat.shape[1]].double().cpu().numpy()
    assert rel_err(feat_d, feat.detach().numpy()) < 1e-5
    emb_t = feat / feat.norm(dim=1, keepdim=True)
    assert rel_err(emb_d.cpu().numpy(), emb_t.detach().numpy()) < 1e-5

    # (3) backward through the whole chain with agreeing routes: fixed upstream gradient, then the loss's own
    demb = np.random.default_rng(9).uniform(-1, 1, size=tuple(emb_t.shape))
    emb_t.backward(torch.tensor(demb))
    tr.backward(demb=torch.as_tensor(demb, dtype=torch.float32).cuda())
    g = tr.plan.export_params(tr.grad)
    errs = {k: rel_err(g[k].cpu().numpy().reshape(tp[k].shape), tp[k].grad.numpy()) for k in tp}
    print("bf16 %s B=%d %dx%d: free-running emb %.2e; teacher-forced gradient worst %.2e" % (net, batch, image, image, e_free, max(errs.values())))
    # ~1e-5 at fc1, growing towards the input: a one-ulp difference in a bf16-stored gradient spreads over the 9*cin inputs of the
    # next data-gradient, some of which cross their own rounding boundary, ... until the difference saturates at the bf16 noise of
    # the gradients themselves (2.5e-3 over LightCNN-9's 10 layers, 1.2e-2 over the deeper CNN's 14).  A wiring or rounding-point
    # error would be O(1), and visible at fc1 first.
    assert max(errs.values()) < 2e-2, errs
    assert errs['fc1_weight'] < 1e-4 and errs['fc1_bias'] < 1e-4
    last = convs[-2].pname
    assert errs[last + '_weight'] < 1e-3 and errs[last + '_bias'] < 1e-3, (last, errs[last + '_weight'])
    tr.backward()
    assert torch.isfinite(tr.grad).all()
    tr.update()


def test_mfmb_kernels_bit_exact():
    """Stand-alone MFM on bf16 activations: forward is exact (max / min of stored values); backward routes / adds in fp32 and rounds once."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    rng = np.random.default_rng(3)
    for ways, c, order in ((3, 99, O.ORDER_GROUP), (3, 198, O.ORDER_RES), (2, 96, O.ORDER_GROUP)):
        b, h, w = 3, 5, 7
        p8 = lambda v: (v + 7) & ~7  # noqa: E731
        x = torch.zeros((b, h, w, p8(c)), dtype=torch.bfloat16, device="cuda")
        x[..., :c] = torch.as_tensor(rng.uniform(-1, 1, (b, h, w, c)), dtype=torch.float32).cuda().bfloat16()
        x[0, 0, 0, :c] = 0.25                      # ties across the slices
        co = (2 * c // 3) if ways == 3 else c // 2
        y = ops.mfmb_fwd(x, c, ways)
        xr = x[..., :c].float().cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)
        yr = O.mfm3(xr) if ways == 3 else O.mfm2(xr)
        assert np.array_equal(y[..., :co].float().cpu().numpy().transpose(0, 3, 1, 2), yr)
        if y.shape[-1] > co:
            assert float(y[..., co:].float().abs().max()) == 0.0
        dy = torch.zeros_like(y)
        dy[..., :co] = torch.as_tensor(rng.uniform(-1, 1, (b, h, w, co)), dtype=torch.float32).cuda().bfloat16()
        add = torch.zeros_like(x)
        add[..., :c] = torch.as_tensor(rng.uniform(-1, 1, (b, h, w, c)), dtype=torch.float32).cuda().bfloat16()
        dx = ops.mfmb_bwd(x, dy, c, ways, order, add=add)
        dyr = dy[..., :co].float().cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)
        ref = (O.mfm3_bwd(xr, dyr, order) if ways == 3 else O.mfm2_bwd(xr, dyr)) + add[..., :c].float().cpu().numpy().transpose(0, 3, 1, 2)
        want = torch.as_tensor(ref.transpose(0, 2, 3, 1), dtype=torch.float32).bfloat16().float().numpy()
        assert np.array_equal(dx[..., :c].float().cpu().numpy(), want)


def test_efm29_bf16_step_vs_emulation():
    """The headline network under the bf16 plan (stand-alone MFM3 of the residual-block inputs, residual adds in the conv epilogue,
    fp32 embedding head) against the free-running rounding emulation: embeddings / loss at the bf16 noise level (see the
    MFM2-stack test for why a free-running comparison cannot be tighter), gradients finite and of the right scale."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    from oracle import efm_oracle_torch as OT
    batch, image = 8, 32
    tr = TripletTrainer(batch, image=image, seed=3, dtype="bf16")
    allp = {k: v.cpu().numpy().astype(np.float64) for k, v in tr.plan.export_params(tr.flat).items()}
    w_head = torch.tensor(allp.pop("head_weight"))
    x = O.uniform01(batch * 3 * image * image, 21).reshape(batch, 3, image, image)
    labels = synth.parity_labels(batch, images_per_identity=2)
    neg = synth.negative_indices(labels, 5)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in allp.items()}
    feat = OT.efm29_forward_bf16(tp, torch.tensor(x))
    emb_r = (feat / feat.norm(dim=1, keepdim=True)) @ w_head.reshape(128, -1).T
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), neg.cuda())
    e_feat = rel_err(tr.last["feat"][:, :342].cpu().numpy(), feat.detach().numpy())
    e_emb = rel_err(tr.last["emb"].cpu().numpy(), emb_r.detach().numpy())
    print("EFM-29 bf16 free-running: feature %.2e, embedding %.2e" % (e_feat, e_emb))
    assert e_feat < 3e-2 and e_emb < 3e-2
    h = batch // 2
    loss_r = OT.triplet_loss(emb_r[:h], emb_r[h:], emb_r[neg.long()].detach(), 0.2)
    assert rel_err(loss.cpu().numpy(), loss_r.detach().numpy()) < 3e-2
    demb = np.random.default_rng(4).uniform(-1, 1, size=tuple(emb_r.shape))
    emb_r.backward(torch.tensor(demb))
    tr.backward(demb=torch.as_tensor(demb, dtype=torch.float32).cuda())
    g = tr.plan.export_params(tr.grad)
    errs = {k: rel_err(g[k].cpu().numpy().reshape(tp[k].shape), tp[k].grad.numpy()) for k in tp}
    print("EFM-29 bf16 free-running gradient errors: fc1 %.2e, worst %.2e" % (errs["fc1_weight"], max(errs.values())))
    # 29 chaotic layers, free-running: the max-norm error is dominated by a few flipped routes (0.2 at fc1 already, through the
    # ill-conditioned L2-norm backward), so the wiring check is the direction of every gradient: cosine similarity with the emulation
    cos = {k: float(np.dot(g[k].cpu().numpy().ravel(), tp[k].grad.numpy().ravel()) /
                    (np.linalg.norm(g[k].cpu().numpy()) * np.linalg.norm(tp[k].grad.numpy()) + 1e-30)) for k in tp}
    print("EFM-29 bf16 gradient cosine vs emulation: min %.4f (%s)" % (min(cos.values()), min(cos, key=cos.get)))
    assert min(cos.values()) > 0.9, cos
    assert torch.isfinite(tr.grad).all()
    tr.update()


@pytest.mark.parametrize("net,batch", [("lightcnn9", 512), ("deepcnn", 128)])
def test_bf16_full_size_properties(net, batch):
    """BASELINE configs[2] (LightCNN-9, 512 images of 3x112x112, bf16, semi-hard mining on) and configs[4] (deeper CNN, 128 images)
    at their FULL sizes — properties that need no oracle run:
    (a) a batch permutation permutes the embeddings bit for bit;  (b) the step is bitwise reproducible;
    (c) scaling the upstream gradient by 2 scales every gradient by exactly 2 (power-of-two scaling commutes with every bf16 / fp32
        rounding on the way);  (d) the gradient of the batch = the sum of the gradients of its two halves (data-parallel contract;
        fp32 accumulation order only);  (e) the semi-hard miner returns a different-identity row for every anchor and the mined
        step's loss and gradients are finite."""
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import MiningTripletTrainer
    image = 112
    mk = efm_symbol.lightcnn9_embedding_net if net == "lightcnn9" else efm_symbol.deepcnn_embedding_net
    tr = MiningTripletTrainer(batch, image=image, outputs=mk(), seed=42, dtype="bf16")
    labels = (np.arange(batch) // 4).astype(np.int32)
    tr.set_labels(labels)
    x = synth.images(batch, 3, image, 1234)
    emb, _ = tr.plan.forward(x, tr.flat, train=False)
    emb = emb.clone()
    perm = torch.randperm(batch, generator=torch.Generator().manual_seed(0)).cuda()
    emb_p, _ = tr.plan.forward(x[perm].contiguous(), tr.flat, train=False)
    assert torch.equal(emb_p, emb[perm])                                                    # (a)
    assert torch.isfinite(emb).all() and float((emb.norm(dim=1) - 1).abs().max()) < 1e-3
    d = emb.shape[1]
    demb = (synth.uniform01(batch * d, 77).view(batch, d) * 2 - 1).contiguous()

    def grad_of(scale):
        tr.plan.forward(x, tr.flat, train=True)
        tr.plan.backward([demb * scale, None], tr.flat, tr.grad)
        return tr.grad.clone()
    g1 = grad_of(1.0)
    assert torch.equal(g1, grad_of(1.0))                                                    # (b)
    assert torch.equal(grad_of(2.0), 2 * g1)                                                # (c)
    half = batch // 2
    th = MiningTripletTrainer(half, image=image, outputs=mk(), seed=42, dtype="bf16")
    th.flat.copy_(tr.flat)
    gsum = torch.zeros_like(g1)
    for sidx in range(2):
        rows = slice(sidx * half, (sidx + 1) * half)
        th.plan.forward(x[rows].contiguous(), th.flat, train=True)
        th.plan.backward([demb[rows].contiguous(), None], th.flat, th.grad)
        gsum += th.grad
    e_sum = rel_err(gsum.cpu().numpy(), g1.cpu().numpy())
    assert e_sum < 1e-4, e_sum                                                              # (d)
    loss = tr.forward_loss(x)                                                               # (e) mining on
    neg = tr.last["neg"].cpu().numpy()
    assert (neg >= 0).all() and (labels[neg] != labels).all()
    tr.backward()
    assert torch.isfinite(loss).all() and torch.isfinite(tr.grad).all() and float(tr.grad.abs().max()) > 0
    print("bf16 %s full size B=%d: shard-sum rel err %.2e, mean loss %.4f" % (net, batch, e_sum, float(loss.mean())))


def test_convb_wgrad_of_the_rowpacked_first_convolution():
    """5 x 1 kernel, pad (2, 0), 15 -> 96 channels on a 16-channel row-packed input: the shape the first convolution's weight gradient
    has in the bf16 plans (one tap group of 5 kernel rows in the halo-tile form)."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w, cin, cout = 2, 11, 23, 15, 96
    x, dy = bf(rand((b, cin, h, w), 1)), bf(rand((b, cout, h, w), 2))
    d = ops.conv_desc(b, h, w, cin, cout, 5, 1, 2, 0)
    dw, db = ops.convb_bwd_weight(d, to_nhwc_bf16(ops, x), to_nhwc_bf16(ops, dy))
    xp = np.pad(x, ((0, 0), (0, 0), (2, 2), (0, 0)))
    dw_ref = np.stack([np.einsum("bchw,bnhw->nc", xp[:, :, a:a + h, :], dy) for a in range(5)], axis=2)[..., None]
    assert rel_err(ops.conv_unpack_weights(d, dw).cpu().numpy(), dw_ref) < 2e-5
    assert rel_err(db[:cout].cpu().numpy(), dy.sum(axis=(0, 2, 3))) < 2e-5


@pytest.mark.parametrize("shape", [(2, 22, 37), (1, 112, 112), (3, 17, 16)])
def test_convb_weight_gradient_straight_from_dz_and_route_bytes(shape):
    """efm_convb_mfm_bwd_weight (first convolution, row-packed 5 x 1, 15 -> 96, MFM2 + 2x2 pooling): the conv-output gradient is formed
    in LDS from dz + the route bytes instead of being written by efm_convb_mfm_pool_bwd and read back — same values in the same
    accumulation order, so the result equals the two-kernel path BIT FOR BIT (odd maps: floor pooling drops the last row / column,
    stage tiles overhang the image)."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    b, h, w = shape
    cin, cout = 15, 96
    d = ops.conv_desc(b, h, w, cin, cout, 5, 1, 2, 0)
    assert ops.convb_mfm_bwd_weight_supported(d, 2, True) and not ops.convb_mfm_bwd_weight_supported(d, 3, True)
    x = to_nhwc_bf16(ops, bf(rand((b, cin, h, w), 1)))
    wb, _ = ops.convb_cast_weights(d, ops.conv_pack_weights(d, dev(rand((cout, cin, 5, 1), 2, 0.3))), need_dgrad=False)
    bias = torch.zeros(d.n_pad16, device="cuda")
    z, route = ops.convb_mfm_fwd(d, x, wb, bias, 2, 0, True)
    dz = torch.as_tensor(rand(tuple(z.shape), 7), dtype=torch.float32).cuda().bfloat16()
    dz[..., cout // 2:] = 0
    dy = ops.convb_mfm_pool_bwd(d, route, dz, 2, True)
    dw_ref, db_ref = ops.convb_bwd_weight(d, x, dy)
    dw, db = ops.convb_mfm_bwd_weight(d, x, route, dz, 2, True)
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    assert float(dw.abs().max()) > 0
