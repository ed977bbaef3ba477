"""The kernel selection bench.py times is the selection tested here, at the benchmark's own size.

bench.py (BASELINE configs[1]: EFM-29, 256 images of 3x112x112, fp32) installs the COMMITTED table
improving_face_recognition_performance_using_triplet_loss_amd/tuning/efm_b256_112_f32.json — Winograd F(2x2,3x3) kernels (8-wave / 4-wave, plain and with the fused
bias+MFM+pool epilogue) for most 3x3 forwards and data gradients, per-layer tilings for the rest and for the weight gradients.
Every kernel in it is compared with its direct implicit-GEMM counterpart at the REAL layer shape (B = 256), the whole tuned step
with the untuned (direct) step, and the tuned plan with the committed config-1 fixture."""
import os
import sys

import numpy as np
import pytest
import torch

from tests.util import rel_err

pytestmark = pytest.mark.gpu
BATCH, IMAGE = 256, 112


def _table():
    from improving_face_recognition_performance_using_triplet_loss_amd import tuning
    table, src = tuning.load("efm", BATCH, IMAGE, "f32")
    assert table is not None, "no committed tuning table for the benchmark configuration (tools/make_tuning.py)"
    return table


def test_committed_table_is_what_bench_installs():
    import bench  # noqa: F401  (importable without side effects)
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    table = _table()
    plan = Plan(efm_symbol.embedding_net(), (BATCH, 3, IMAGE, IMAGE), "cuda")
    assert plan.apply_tuning(table) == sum(1 for s in plan.steps if s.op == "conv") == 31   # 29 convs + fc1 + head
    assert plan.tuning_table() == {k: dict(v) for k, v in table.items()}                     # round trip: nothing dropped / altered
    n_wino = sum(v["wino_fwd"] + v["wino_dgrad"] for v in table.values())
    assert n_wino >= 20, n_wino                                                              # the selection really is mostly Winograd


def test_every_selected_kernel_matches_its_direct_counterpart_at_the_real_shape():
    """Layer by layer at B = 256: forward (plain or fused epilogue), data gradient and weight gradient under the table's choice vs
    the heuristic direct kernel on the same random operands: <= 1e-5 of the largest magnitude (Winograd differs by fp32 summation
    order only); fused epilogues: identical route bytes wherever the conv values are not within rounding of a tie."""
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, ops
    from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan
    table = _table()
    tuned = Plan(efm_symbol.embedding_net(), (BATCH, 3, IMAGE, IMAGE), "cuda")
    tuned.apply_tuning(table)
    gen = torch.Generator(device="cuda").manual_seed(0)
    seen, report = set(), []
    for st in tuned.steps:
        if st.op != "conv":
            continue
        d = st.desc
        key = (d.hin, d.cin, d.cout, d.kh, st.epi is not None, st.inputs[0].needs_grad)
        if key in seen:
            continue
        seen.add(key)
        d0 = ops.conv_desc(d.batch, d.hin, d.win, d.cin, d.cout, d.kh, d.kw, d.pad_h, d.pad_w)   # heuristics, direct kernels
        x = torch.rand((d.batch, d.hin, d.win, d.cin_p), device="cuda", generator=gen) - 0.5
        x[..., d.cin:] = 0
        wt = (torch.rand((d.cout, d.cin, d.kh, d.kw), device="cuda", generator=gen) - 0.5) * (2.0 / np.sqrt(d.cin * d.kh * d.kw))
        w = ops.conv_pack_weights(d0, wt)
        bias = torch.zeros(d.n_pad16, device="cuda")
        bias[: d.cout] = torch.rand(d.cout, device="cuda", generator=gen) - 0.5
        errs = {}
        if st.epi is None:
            ref = ops.conv_fwd(d0, x, w, bias)
            got = ops.wino_fwd(d, x, ops.wino_make_u(d, w), bias) if st.wino_fwd else ops.conv_fwd(d, x, w, bias)
            errs["fwd"] = float((got - ref).abs().max() / ref.abs().max())
            if not st.wino_fwd:
                assert torch.equal(got, ref)            # direct tilings are bit-identical
            del ref, got
        else:
            e = st.epi
            zr, rr = ops.conv_mfm_fwd(d0, x, w, bias, e["ways"], e["order"], e["pool"])
            if st.wino_fwd:
                z, r = ops.wino_mfm_fwd(d, x, ops.wino_mfm_make_u(d, w, e["ways"]), bias, e["ways"], e["order"], e["pool"])
            else:
                z, r = ops.conv_mfm_fwd(d, x, w, bias, e["ways"], e["order"], e["pool"])
            errs["fused"] = float((z - zr).abs().max() / zr.abs().max())
            creal = ops.mfm_out_channels(d.cout, e["ways"])      # route bytes of the pad channels are never written: unspecified
            r, rr = r[..., :creal], rr[..., :creal]
            flips = float((r != rr).float().mean())
            errs["route_flips"] = flips
            assert flips < 1e-4, (st.pname, flips)      # a route differs only where two candidates agree to the last bits
            if not st.wino_fwd:
                assert torch.equal(z, zr) and torch.equal(r, rr)
            del z, r, zr, rr
        dy = torch.rand((d.batch, d.hout, d.wout, d.cout_p), device="cuda", generator=gen) - 0.5
        dy[..., d.cout:] = 0
        if st.inputs[0].needs_grad:
            ref = ops.conv_bwd_data(d0, dy, ops.conv_make_dgrad_weights(d0, w))
            got = ops.wino_bwd_data(d, dy, ops.wino_make_u(d, w, dgrad=True)) if st.wino_dgrad else \
                ops.conv_bwd_data(d, dy, ops.conv_make_dgrad_weights(d, w))
            errs["dgrad"] = float((got - ref).abs().max() / ref.abs().max())
            if not st.wino_dgrad:
                assert torch.equal(got, ref)
            del ref, got
        dwr, dbr = ops.conv_bwd_weight(d0, x, dy)
        dw, db = ops.conv_bwd_weight(d, x, dy)
        errs["wgrad"] = float((dw - dwr).abs().max() / dwr.abs().max())
        errs["bgrad"] = float((db - dbr).abs().max() / dbr.abs().max())
        report.append((st.pname, table[st.pname], errs))
        for k, v in errs.items():
            if k != "route_flips":
                assert v < 1e-5, (st.pname, k, v)
        del x, dy, dw, dwr
    worst = max(max(v for k, v in e.items() if k != "route_flips") for _, _, e in report)
    print("tuned kernels vs direct at B=%d: %d distinct layer shapes, worst rel err %.2e" % (BATCH, len(report), worst))


def test_tuned_step_matches_direct_step_at_benchmark_size():
    """The whole tuned step vs the untuned one, B = 256: embeddings and loss <= 1e-5; the parameter gradient agrees as far as two
    correct fp32 forwards can (max/min/pool routes flip where candidates tie to the last bit): cosine > 0.9999 per step, and the
    tuned step is bitwise reproducible."""
    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    x = synth.images(BATCH, 3, IMAGE, 1234)
    neg = synth.negative_indices(synth.parity_labels(BATCH), 77).cuda()
    demb = (synth.uniform01(BATCH * 128, 77).view(BATCH, 128) * 2 - 1).contiguous()
    res = {}
    for name, tab in (("direct", None), ("tuned", _table())):
        tr = TripletTrainer(BATCH, image=IMAGE, seed=42, tuning=tab)
        loss = tr.forward_loss(x, neg).clone()
        tr.backward(demb=demb)
        g1 = tr.grad.clone()
        tr.forward_loss(x, neg)
        tr.backward(demb=demb)
        assert torch.equal(g1, tr.grad)
        res[name] = (tr.last["emb"].clone(), loss, g1)
        del tr
        torch.cuda.empty_cache()
    e_emb = rel_err(res["tuned"][0].cpu().numpy(), res["direct"][0].cpu().numpy())
    e_loss = rel_err(res["tuned"][1].cpu().numpy(), res["direct"][1].cpu().numpy())
    gt, gd = res["tuned"][2].double(), res["direct"][2].double()
    cos = float((gt * gd).sum() / (gt.norm() * gd.norm()))
    e_g = float((gt - gd).abs().max() / gd.abs().max())
    print("tuned vs direct step at B=%d: emb %.2e loss %.2e grad max-rel %.2e cosine %.7f" % (BATCH, e_emb, e_loss, e_g, cos))
    assert e_emb < 1e-5 and e_loss < 1e-5
    assert cos > 0.9999 and e_g < 3e-2


def test_tuned_plan_vs_config1_golden():
    """BASELINE configs[0] (64 faces) through the TUNED kernels against the committed fp64 fixture: 1e-3."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_config1_golden as M
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "config1_efm112.npz"))
    params, w_head, x, neg = M.inputs()
    tr = TripletTrainer(M.BATCH, image=M.IMAGE, tuning=_table())
    assert sum(bool(getattr(s, "wino_fwd", False)) for s in tr.plan.steps) >= 10
    allp = dict(params)
    allp["head_weight"] = w_head
    tr.plan.load_params(tr.flat, allp)
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), torch.as_tensor(neg).cuda())
    assert rel_err(tr.last["emb"].cpu().numpy(), z["emb"]) < 1e-3
    assert rel_err(loss.cpu().numpy(), z["loss"]) < 1e-3
    dl, dr = loss.cpu().numpy() - 0.2, z["loss"] - 0.2
    assert np.abs(dl - dr).max() < 1e-3 * np.abs(dr).max() + 1e-7


def test_tuned_kernels_gradients_vs_fp64_oracle_with_route_handover():
    """The benchmarked selection against the fp64 oracle DIRECTLY (not through the direct kernels): 8 images of 3x112x112 — every
    real map size, the 7 -> 3 floor pooling — on a plan that applies the committed table with one kernel per graph node (so that
    the MFM / pooling inputs exist for the route hand-over, as in test_e2e_gpu.py::test_112_step_vs_oracles): the 3x3 forwards
    and data gradients run wino4_k / wino_fwd_k, the weight gradients wino_wgrad_k.  Embeddings / loss and all 60 backbone parameter
    gradients <= 1e-3 (north_star's tolerance) with the oracle following the device's arg-max routes."""
    from oracle import efm_oracle as O
    from improving_face_recognition_performance_using_triplet_loss_amd import ops, synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer
    batch, image = 8, IMAGE
    shapes = O.efm29_param_shapes(3, image)
    params = O.init_params(shapes, 42)
    w_head = O.uniform_pm((128, 342), 777, O.xavier_uniform_scale((128, 342)))
    x = O.uniform01(batch * 3 * image * image, 1234).reshape(batch, 3, image, image)
    tr = TripletTrainer(batch, image=image, optimizer="sgd", lr=0.05, wd=1e-5, fuse=False, tuning=_table())
    convs = [s for s in tr.plan.steps if s.op == "conv"]
    assert sum(bool(s.wino_fwd) for s in convs) >= 20 and sum(bool(s.wino_dgrad) for s in convs) >= 20
    assert sum(bool(ops.conv_kernel_info(s.desc, ops.PASS_WGRAD)[0].startswith("wino_wgrad_k")) for s in convs) >= 20
    allp = dict(params)
    allp["head_weight"] = w_head
    tr.plan.load_params(tr.flat, allp)
    neg = synth.negative_indices(synth.parity_labels(batch, images_per_identity=2), 99)
    demb = np.random.default_rng(6).uniform(-1, 1, size=(batch, 128))
    loss = tr.forward_loss(torch.as_tensor(x, dtype=torch.float32).cuda(), neg.cuda())
    routing = {k: v.cpu().numpy().astype(np.float64) for k, v in tr.plan.routing_inputs().items()}
    tr.backward(demb=torch.as_tensor(demb, dtype=torch.float32).cuda())
    g = tr.plan.export_params(tr.grad)
    loss_r, emb_r, _, grads_r, ghead_r = O.train_step_loss(params, w_head, x, neg.numpy(), 0.2, demb=demb, routing=routing)
    assert rel_err(tr.last["emb"].cpu().numpy(), emb_r) < 1e-3 and rel_err(loss.cpu().numpy(), loss_r) < 1e-3
    worst, worst_name = 0.0, None
    for name, r in grads_r.items():
        e = rel_err(g[name].cpu().numpy().reshape(r.shape), r)
        if e > worst:
            worst, worst_name = e, name
    print("tuned (Winograd) kernels vs fp64 oracle, same routes: %d gradients, worst %.3e (%s)" % (len(grads_r), worst, worst_name))
    assert len(grads_r) == 60 and worst < 1e-3, (worst, worst_name)
    assert rel_err(g["head_weight"].cpu().numpy().reshape(128, 342), ghead_r) < 1e-3
