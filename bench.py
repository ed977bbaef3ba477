#!/usr/bin/env python
"""Headline benchmark: triplets/s of the EFM-29 triplet training step, 112x112x3, 256 images per GPU, fp32.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one resident synthetic batch: forward of [128 anchors ; 128 positives]
-> in-batch negatives -> TripletLoss -> backward -> (N>1: RCCL all-reduce of the flat gradient, overlapped)
-> SGD update.  Inputs are generated on the device before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_IMAGE_STEP = 15256522908  # SURVEY.md §8d / BASELINE.md §2: 3*fwd - dgrad(conv1), EFM-29 @112, unpadded
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense v_mfma_f32_16x16x32_bf16: 256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD


def baseline_metric():
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "triplets/sec (whole node) EFM 112x112 bs256/GPU at 1/2/4/8 MI355X; LFW acc"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU (BASELINE: 256)")
    ap.add_argument("--image", type=int, default=112)
    ap.add_argument("--workload", choices=["efm", "lightcnn9", "deepcnn"], default="efm",
                    help="efm = BASELINE configs[1] (default, the headline); lightcnn9 = configs[2] geometry in fp32 with in-batch "
                         "semi-hard mining (every image an anchor) — a secondary line, not the headline metric")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 (lightcnn9 only) = BASELINE configs[2]: bf16 operands / activations, fp32 accumulate + master weights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bounded bf16 runs of BASELINE configs[2] / configs[4]")
    ap.add_argument("--cpu-batch", type=int, default=64, help="images of the CPU-baseline sample: BASELINE configs[0] = 64 faces")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="untimed CPU-baseline steps (SURVEY.md §8d: 3)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed CPU-baseline steps, median reported (SURVEY.md §8d: 10; ~7 s each)")
    ap.add_argument("--no-host-loops", action="store_true", help="skip the per-sample host loops vs device kernels comparison")
    return ap.parse_args()


def kernel_families(trainer, torch, iters=3, only=None, dedup=True):
    """The convolution launches of one training step, grouped by kernel instance (the names rocprofv3 reports): every distinct
    (kernel, layer shape) is timed in isolation with HIP events on the launch stream; per instance: launches per step, ms per step,
    ALGORITHMIC flops (unpadded direct-convolution 2*M*cout*cin*kh*kw — what `roofline.achieved` uses) and EXECUTED matrix-core
    flops (padded tiles; a Winograd launch executes its 16 transformed-domain GEMMs = 1/2.25 of the algorithmic multiplies plus
    padding), from efm_conv_kernel_info."""
    from improving_face_recognition_performance_using_triplet_loss_amd import ops
    plan, dev = trainer.plan, trainer.device
    fam, cache = {}, {}

    def timed(run):
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters

    def add(name, key, run, alg, exe, nbytes=0.0):
        if only is not None and not name.startswith(only):  # tools/family_probe.py: just this kernel instance (rocprofv3 --pmc passes)
            return
        if key not in cache or not dedup:  # dedup=False (profiling probe): launch mix = the step's, repeated shapes included
            cache[key] = timed(run)
        f = fam.setdefault(name, {"launches": 0, "ms": 0.0, "alg_flop": 0.0, "mfma_flop": 0.0, "alg_bytes": 0.0})
        f["launches"] += 1
        f["alg_bytes"] += nbytes
        f["ms"] += cache[key]
        f["alg_flop"] += alg
        f["mfma_flop"] += exe

    for st in plan.steps:
        if st.op != "conv":
            continue
        d = st.desc
        shape = (d.hin, d.win, d.cin, d.cout, d.kh, d.pad_h, st.epi is not None and (st.epi["ways"], st.epi["pool"]), d.tune_fwd, d.tune_dgrad,
                 d.tune_wgrad, bool(getattr(st, "wino_fwd", False)), bool(getattr(st, "wino_dgrad", False)))
        alg = 2.0 * d.batch * d.hout * d.wout * d.cout * d.cin * d.kh * d.kw
        x = torch.rand((d.batch, d.hin, d.win, d.cin_p), device=dev)
        x[..., d.cin:] = 0
        dy = torch.rand((d.batch, d.hout, d.wout, d.cout_p), device=dev)
        dy[..., d.cout:] = 0
        w = torch.rand((d.n_pad16, d.k_pad), device=dev)
        b = torch.zeros(d.n_pad16, device=dev)
        e = st.epi
        # algorithmic HBM bytes of a launch: every operand once (fp32): input + output (+ route bytes of a fused epilogue) + weights
        xb, yb, wb_ = 4.0 * d.batch * d.hin * d.win * d.cin_p, 4.0 * d.batch * d.hout * d.wout * d.cout_p, 4.0 * d.n_pad16 * d.k_pad
        if e is not None:
            co = ops.mfm_out_channels(d.cout, e["ways"])
            cop = (co + 3) & ~3
            zpix = d.batch * ((d.hout // 2) * (d.wout // 2) if e["pool"] else d.hout * d.wout)
            fwd_bytes = xb + wb_ + zpix * cop * 5.0
        else:
            fwd_bytes = xb + wb_ + yb
        if e is not None:
            if getattr(st, "wino_fwd", False):
                u = ops.wino_mfm_make_u(d, w, e["ways"])
                name, exe = ops.conv_kernel_info(d, ops.PASS_WINO_FUSED, e["ways"], e["pool"])
                add(name, ("f",) + shape, lambda: ops.wino_mfm_fwd(d, x, u, b, e["ways"], e["order"], e["pool"]), alg, exe, fwd_bytes)
            else:
                name, exe = ops.conv_kernel_info(d, ops.PASS_FUSED, e["ways"], e["pool"])
                add(name, ("f",) + shape, lambda: ops.conv_mfm_fwd(d, x, w, b, e["ways"], e["order"], e["pool"]), alg, exe, fwd_bytes)
        else:
            y = torch.empty_like(dy)
            if getattr(st, "wino_fwd", False):
                u = ops.wino_make_u(d, w)
                name, exe = ops.conv_kernel_info(d, ops.PASS_WINO_FWD)
                add(name, ("f",) + shape, lambda: ops.wino_fwd(d, x, u, b, out=y), alg, exe, fwd_bytes)
            else:
                name, exe = ops.conv_kernel_info(d, ops.PASS_FWD)
                add(name, ("f",) + shape, lambda: ops.conv_fwd(d, x, w, b, out=y), alg, exe, fwd_bytes)
        if st.inputs[0].needs_grad:
            dx = torch.empty_like(x)
            if getattr(st, "wino_dgrad", False):
                u = ops.wino_make_u(d, w, dgrad=True)
                name, exe = ops.conv_kernel_info(d, ops.PASS_WINO_DGRAD)
                add(name, ("d",) + shape, lambda: ops.wino_bwd_data(d, dy, u, out=dx), alg, exe, xb + yb + wb_)
            else:
                wd = torch.rand((d.dn_pad16, d.dk_pad), device=dev)
                name, exe = ops.conv_kernel_info(d, ops.PASS_DGRAD)
                add(name, ("d",) + shape, lambda: ops.conv_bwd_data(d, dy, wd, out=dx), alg, exe, xb + yb + wb_)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        name, exe = ops.conv_kernel_info(d, ops.PASS_WGRAD)
        add(name + " (+ wgrad_reduce_k)", ("w",) + shape, lambda: ops.conv_bwd_weight(d, x, dy, dw=dw, dbias=db), alg, exe, xb + yb + wb_)
        del x, dy, w
    return fam


def dominant_kernel_roofline(trainer, torch, iters=3):
    """`roofline` = the kernel instance that costs the most time per step (all its launches of a step together: sum of algorithmic
    flops / sum of launch durations, each timed with HIP events on the launch stream, kernel alone on the chip), with the executed
    matrix-core flops and the busy fraction they imply next to the algorithmic figure, and the per-instance table it was picked from."""
    fam = kernel_families(trainer, torch, iters)
    table = []
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        t = f["ms"] * 1e-3
        table.append({"kernel": name, "launches_per_step": f["launches"], "ms_per_step": round(f["ms"], 3),
                      "tflops_mfma_executed": round(f["mfma_flop"] / t / 1e12, 1),
                      "frac": round(f["mfma_flop"] / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                      "direct_equivalent_tflops": round(f["alg_flop"] / t / 1e12, 1)})
    top_name, top = max(fam.items(), key=lambda kv: kv[1]["ms"])
    t = top["ms"] * 1e-3
    direct_equiv = top["alg_flop"] / t / 1e12      # unpadded direct-convolution flops per second
    executed = top["mfma_flop"] / t / 1e12          # flops the matrix cores execute: a Winograd launch = its 16 transformed-domain GEMMs, padded tiles
    traffic, traffic_src = None, None
    for cand in ("round3_traffic.json", "round2_traffic.json"):
        # HBM bytes per launch of the SAME kernel instance from committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction +
        # WRITE_SIZE, separate passes) — read from the file, NOT measured in this run
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", cand)))
        except Exception:
            continue
        if prof.get("kernel", "").replace(" ", "") == top_name.replace(" ", ""):
            traffic, traffic_src = prof["traffic"], "profiles/%s (committed rocprofv3 --pmc passes; not measured in this run)" % cand
            break
    conv_ms = sum(f["ms"] for f in fam.values())
    return {"bound": "mfma", "kernel": "%s: the %d launches of one step (every layer that resolves to this instance), kernel alone on the chip"
                                       % (top_name, top["launches"]),
            "achieved": round(executed, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(executed / PEAK_FP32_MFMA_TFLOPS, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "flop_per_launch": top["mfma_flop"] / top["launches"], "ms_per_launch": round(top["ms"] / top["launches"], 4),
            "launches_per_step": top["launches"], "ms_per_step": round(top["ms"], 3),
            "direct_conv_flop_per_launch": top["alg_flop"] / top["launches"],
            "direct_equivalent_tflops": round(direct_equiv, 2),
            "direct_equivalent_over_peak": round(direct_equiv / PEAK_FP32_MFMA_TFLOPS, 4),
            "algorithmic_bytes_per_launch": top["alg_bytes"] / top["launches"],
            "note": "achieved / frac = matrix-core flops the launches EXECUTE / time / peak = utilisation of the MFMA pipe (for a Winograd "
                    "F(2x2,3x3) kernel: its 16 transformed-domain GEMMs incl. tile padding, 1/2.25 of the direct-convolution multiplies). "
                    "direct_equivalent_* prices the same launches at SURVEY.md §8d's unpadded direct-convolution flop count and can exceed "
                    "the peak — it is a speed-up figure, not a utilisation",
            "conv_kernel_ms_per_step_serial": round(conv_ms, 3),
            "mfma_flop_executed_per_step": sum(f["mfma_flop"] for f in fam.values()), "families": table[:8]}


def step_bound(plan, peak_tflops, hbm_tbs=8.0):
    """SURVEY.md §8d's per-layer bound of one training step of `plan`: for every convolution and pass (forward, data gradient, weight
    gradient) max(t_flops, t_bytes) with t_flops = unpadded 2*M*cout*cin*kh*kw / MFMA peak of the plan's dtype and t_bytes = the
    pass's ALGORITHMIC HBM bytes / 8 TB/s — every operand once: the layer's input, its stored output (for a fused conv -> MFM [-> pool]
    layer that is z + one route byte per element, NOT the full-resolution conv output: a design that materialises the conv-output
    gradient moves more than this bound prices) and the weights.  Returns the sums and which side governs."""
    es_act = 2 if plan.dtype == "bf16" else 4
    t_f = t_b = t_max = 0.0
    layers = []
    for st in plan.steps:
        if st.op != "conv":
            continue
        d = st.desc
        f32 = getattr(st, "f32", False) or plan.dtype != "bf16"
        es = 4 if f32 else es_act
        pad = (lambda c: (c + 3) // 4 * 4) if es == 4 else (lambda c: (c + 7) // 8 * 8)
        m = d.batch * d.hout * d.wout
        fl = 2.0 * m * d.cout * d.cin * d.kh * d.kw
        xb = d.batch * d.hin * d.win * pad(d.cin) * es
        if st.epi is not None:
            co = d.cout // 2 if st.epi["ways"] == 2 else 2 * d.cout // 3
            zp = d.batch * (d.hout // 2) * (d.wout // 2) if st.epi["pool"] else m
            zb = zp * pad(co) * (es + 1)
        else:
            zb = m * pad(d.cout) * es
        wn = d.cout * d.cin * d.kh * d.kw
        peak = (157.3 if f32 else peak_tflops) * 1e12
        passes = [("fwd", xb + zb + wn * es)]
        if st.inputs[0].needs_grad:
            passes.append(("dgrad", xb + zb + wn * es))
        passes.append(("wgrad", xb + zb + wn * 4))
        row = {"layer": st.pname}
        for name, nbytes in passes:
            tf, tb = fl / peak * 1e3, nbytes / (hbm_tbs * 1e12) * 1e3
            t_f, t_b, t_max = t_f + tf, t_b + tb, t_max + max(tf, tb)
            row[name] = [round(tf, 4), round(tb, 4)]
        layers.append(row)
    return {"t_flops_ms": round(t_f, 3), "t_bytes_ms": round(t_b, 3), "bound_ms": round(t_max, 3), "layers": layers}


def secondary_configs(torch, device, image, steps=10, warmup=3):
    """BASELINE configs[2] and configs[4] (one GPU each), a few seconds in all, inside the same JSON line so that a driver-timed
    figure exists for them: LightCNN-9 256-d, 512 images, bf16, in-batch semi-hard mining (every image an anchor);
    the build-defined deeper CNN 512-d, 128 images, bf16, same step.  Not the headline metric."""
    from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import MiningTripletTrainer
    res = {}
    for key, batch, outputs, flop in (("configs[2] LightCNN-9 256-d B=512 bf16 semi-hard", 512, efm_symbol.lightcnn9_embedding_net, 4667572224),
                                      ("configs[4] deeper CNN 512-d B=128 bf16 semi-hard", 128, efm_symbol.deepcnn_embedding_net, 3 * 5199839232 - 180633600)):
        tr = MiningTripletTrainer(batch, image=image, optimizer="sgd", lr=2.4e-4, wd=1e-5, margin=0.2, device=device, seed=42,
                                  outputs=outputs(), dtype="bf16")
        tr.set_labels(torch.arange(batch) // 4)
        xs = [synth.images(batch, 3, image, 1234 + s, device) for s in range(2)]
        for i in range(warmup):
            tr.step(xs[i % 2], None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = tr.step(xs[i % 2], None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        bound = step_bound(tr.plan, PEAK_BF16_MFMA_TFLOPS)
        res[key] = {"roofline": {"bound": "sum over layers and passes of max(t_flops, t_bytes) (SURVEY.md §8d): MFMA 2.5 PFLOP/s dense bf16 / HBM 8 TB/s",
                                 "t_flops_ms": bound["t_flops_ms"], "t_bytes_ms": bound["t_bytes_ms"], "bound_ms": bound["bound_ms"],
                                 "achieved_ms": round(dt * 1e3, 3), "frac": round(bound["bound_ms"] / (dt * 1e3), 4),
                                 "governs": "mfma" if bound["t_flops_ms"] >= bound["t_bytes_ms"] else "hbm"},
                    "triplets_per_s": round(batch / dt, 1), "images_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 3), "steps": steps,
                    "warmup": warmup, "dtype": "bf16 operands / activations, fp32 accumulate + master weights",
                    "step_mfma_roofline_frac": round(batch / dt * flop / (PEAK_BF16_MFMA_TFLOPS * 1e12), 4), "peak_tflops": PEAK_BF16_MFMA_TFLOPS,
                    "loss": round(float(loss.mean().item()), 6)}
        del tr, xs
        torch.cuda.empty_cache()
    return res


def cpu_baseline(batch, image, torch, warmup=3, steps=10):
    """The CPU restatement of the reference graph (oracle/efm_oracle_torch.py, torch-CPU fp32 / oneDNN) timed on this
    box's host cores on a bounded sample: BASELINE configs[0] — `batch` = 64 images per step (32 triplets), fwd + bwd + SGD, ID head
    off — `warmup` untimed + `steps` timed steps, median (3 + 10; ~100 s of host work at ~7 s per step)."""
    from oracle import efm_oracle as O
    from oracle import efm_oracle_torch as OT
    threads = torch.get_num_threads()
    shapes = O.efm29_param_shapes(3, image)
    g = torch.Generator().manual_seed(0)
    p = {}
    for name, shp in shapes.items():
        if name.endswith("_bias"):
            p[name] = torch.zeros(shp, requires_grad=True)
        else:
            s = O.xavier_uniform_scale(shp)
            p[name] = ((torch.rand(shp, generator=g) * 2 - 1) * s).requires_grad_(True)
    wh = ((torch.rand((128, 342), generator=g) * 2 - 1) * O.xavier_uniform_scale((128, 342))).requires_grad_(True)
    x = torch.rand((batch, 3, image, image), generator=g)
    h = batch // 2
    neg = (torch.arange(h) + 1) % h
    times = []
    cpu_model = "unknown CPU"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    for i in range(warmup + steps):
        for t in list(p.values()) + [wh]:
            t.grad = None
        t0 = time.perf_counter()
        OT.train_step(p, wh, x, neg, 0.2)
        with torch.no_grad():
            for t in list(p.values()) + [wh]:
                t -= 2.4e-4 * (t.grad / h + 1e-5 * t)
        dt = time.perf_counter() - t0
        if i >= warmup:
            times.append(dt)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(h / dt, 3), "unit": "triplets/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
            "logical_cpus": os.cpu_count(),
            "sample": "torch-CPU fp32 restatement (oracle/efm_oracle_torch.py) of the same EFM-29 step on %d images of "
                      "%dx%dx3 (= %d triplets; BASELINE configs[0]), %d warm-up + %d timed steps, median %.2f s/step (min %.2f, max %.2f), "
                      "%d torch threads on %s" % (batch, image, image, h, warmup, len(times), dt, min(times), max(times), threads, cpu_model)}


def host_loops(torch, device, cases=((128, 128, "train_efm.py / bench step: 128 anchors of a 256-image batch, 128-d"),
                                     (16384, 128, "pre-trained_efm_v3.py:132: 16 384 anchors, Dense(128) output"))):
    """SURVEY.md §8d's second comparison: the reference's literal per-sample HOST loops of one step — negative pick
    (train_efm.py:234-239 = pre-trained_efm_v3.py:202-207), `cosine_dist` (train_efm.py:26-34) and the per-value read-backs of the
    CSV writer (:254-255), restated on CPU tensors in oracle/host_loops.py — timed on this box's host beside what replaces them here:
    `data.pick_negatives` (vectorised rejection sampling on the label vector) + `efm_gather_rows` + `efm_cosine_pairs` on the device
    + ONE copy of the two similarity vectors to the host.  Same labels and embeddings on both sides; seconds per step."""
    import random
    import numpy as np
    from oracle import host_loops as H
    from improving_face_recognition_performance_using_triplet_loss_amd import data, ops, synth
    res = []
    warm = torch.rand(16, 8)   # first-call costs of the torch CPU ops (dispatcher, thread pool) stay out of the timed loops
    H.cosine_dist_loop(warm[:8], warm[8:], warm[:8], 8)
    for b, dim, what in cases:
        lab = (torch.arange(b, dtype=torch.int64) % max(b // 4, 2)).to(torch.float32)
        lab2 = torch.cat([lab, lab])
        fc = synth.uniform01(2 * b * dim, 4242, device="cpu").view(2 * b, dim) - 0.5
        t0 = time.perf_counter()
        neg, _ = H.pick_negatives_loop(lab2, fc, b, random.Random(1))
        t1 = time.perf_counter()
        pd, nd = H.cosine_dist_loop(fc[:b], fc[b:], neg, b)
        t2 = time.perf_counter()
        rows = H.csv_rows(pd, nd, b)
        t3 = time.perf_counter()
        emb = fc.to(device)
        rng = np.random.default_rng(1)
        times = []
        for _ in range(6):
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            idx = data.pick_negatives(lab2, b, b, rng).to(device)
            n = ops.gather_rows(emb, idx)
            s_ap, s_an = ops.cosine_pairs(emb[:b], emb[b:], n)
            host = torch.stack([s_ap, s_an]).cpu()          # what the CSV writer needs, in one copy
            times.append(time.perf_counter() - s0)
        dev_s = sorted(times[1:])[len(times[1:]) // 2]
        # same arithmetic on both sides: the anchor-positive similarities do not depend on the draw
        err = float((host[0] - torch.tensor([r[0] for r in rows])).abs().max())
        cpu_s = t3 - t0
        res.append({"anchors": b, "dim": dim, "case": what, "reference_loops_s": round(cpu_s, 5),
                    "negative_pick_s": round(t1 - t0, 5), "cosine_dist_s": round(t2 - t1, 5), "csv_readbacks_s": round(t3 - t2, 5),
                    "device_path_s": round(dev_s, 6), "ratio": round(cpu_s / dev_s, 1), "s_ap_max_abs_diff": err})
    return {"unit": "seconds per step", "kind": "port", "note": "reference loops restated on torch CPU tensors (oracle/host_loops.py; MXNet "
            "absent); on the reference's GPU run every scalar read-back is additionally a device synchronisation", "cases": res}


def self_launch(args):
    """`python bench.py --gpus N` started plainly (no torchrun environment): start the N ranks as a CHILD process — one process per
    GPU, `torch.distributed.run`, rendezvous on 127.0.0.1 — and exit with its return code.  Nothing in this process has touched the
    GPU yet (torch is not even imported), so no program is replaced after GPU initialisation."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # rehearsal knobs (not used by the driver): all ranks on one device / gloo instead of RCCL, to exercise the
    # multi-process path on a 1-GPU box
    if os.environ.get("EFM_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("EFM_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("EFM_FORCE_ALLREDUCE"):
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from improving_face_recognition_performance_using_triplet_loss_amd import synth
    from improving_face_recognition_performance_using_triplet_loss_amd.trainer import TripletTrainer

    flop_per_image = FLOP_PER_IMAGE_STEP
    if args.workload in ("lightcnn9", "deepcnn"):
        from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol
        from improving_face_recognition_performance_using_triplet_loss_amd.trainer import MiningTripletTrainer
        deep = args.workload == "deepcnn"
        # 3*fwd - dgrad(conv1): LightCNN-9 @112 / the deeper CNN (fwd 5 199 839 232, conv1 dgrad 180 633 600)
        flop_per_image = 3 * 5199839232 - 180633600 if deep else 4667572224
        tr = MiningTripletTrainer(args.batch, image=args.image, optimizer="sgd", lr=2.4e-4, wd=1e-5, margin=0.2, device=device, seed=42,
                                  outputs=efm_symbol.deepcnn_embedding_net() if deep else efm_symbol.lightcnn9_embedding_net(),
                                  dtype=args.dtype, autotune=args.dtype == "f32" and os.environ.get("EFM_AUTOTUNE", "1") != "0")
        ids = (torch.arange(args.batch) // 4) + rank * (args.batch // 4)  # P = B/4 identities x K = 4 images
        tr.set_labels(ids)
        batches = [(synth.images(args.batch, 3, args.image, 1234 + 1000 * rank + s, device), None) for s in range(2)]
        triplets_per_step = args.batch
    else:
        # kernel selection: the COMMITTED table for this configuration (what tests/test_tuned_gpu.py exercises at full size) unless
        # EFM_TUNING_FILE names another one or EFM_AUTOTUNE=live asks for a fresh timing run; EFM_AUTOTUNE=0 = heuristics only
        from improving_face_recognition_performance_using_triplet_loss_amd import tuning as tuning_mod
        mode = os.environ.get("EFM_AUTOTUNE", "1")
        table, tuning_src = (None, None)
        if args.dtype == "f32" and mode not in ("0", "live"):
            table, tuning_src = tuning_mod.load("efm", args.batch, args.image, "f32", file=os.environ.get("EFM_TUNING_FILE"))
        tr = TripletTrainer(args.batch, image=args.image, optimizer="sgd", lr=2.4e-4, wd=1e-5, margin=0.2, device=device, seed=42,
                            n_buckets=int(os.environ.get("EFM_BUCKETS", "6")), dtype=args.dtype, tuning=table,
                            autotune=args.dtype == "f32" and mode != "0")
        if table is None:
            tuning_src = "live autotune" if (args.dtype == "f32" and mode != "0") else "heuristics"
        labels = synth.parity_labels(args.batch, rank=rank)
        batches = []
        for s in range(2):  # resident synthetic batches, seed = 1234 + 1000*rank + step
            x = synth.images(args.batch, 3, args.image, 1234 + 1000 * rank + s, device)
            neg = synth.negative_indices(labels, 77 + 1000 * rank + s).to(device)
            batches.append((x, neg))
        triplets_per_step = args.batch // 2

    for i in range(args.warmup):
        tr.step(*batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = tr.step(*batches[i % 2])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_mean = float(loss.mean().item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        triplets = world * triplets_per_step * args.steps / dt
        images = world * args.batch * args.steps / dt
        out = {
            "metric": baseline_metric(), "value": round(triplets, 2), "unit": "triplets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: EFM-29 128-d embedding, %d images/GPU of %dx%dx3, fp32, "
                                   "fwd+bwd+SGD, reference batch layout (1 triplet per anchor)" % (args.batch, args.image, args.image),
                       "images_per_gpu": args.batch, "parallelism": "dp%d" % world},
            "images_per_s": round(images, 1),
            # whole step, SURVEY.md §8d's unpadded direct-convolution flop count / time / fp32 MFMA peak (north_star's ">= 40 %" figure);
            # the Winograd kernels execute 2.25x fewer multiplies for those flops, so this is a throughput figure, not a utilisation
            "step_mfma_roofline_frac": round(images / world * flop_per_image / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
            "collectives_per_step": tr.reducer.last_collectives,
            "loss": round(loss_mean, 6),
        }
        if args.workload in ("lightcnn9", "deepcnn"):
            name, cfg = (("deeper CNN 512-d", 4) if args.workload == "deepcnn" else ("LightCNN-9 256-d", 2))
            out["metric"] = "triplets/sec %s 112x112, in-batch semi-hard mining, %s (secondary: BASELINE configs[%d])" % (name, args.dtype, cfg)
            out["dtype"] = args.dtype
            out["config"] = {"workload": "BASELINE configs[%d]: %s (MFM2), %d images/GPU, every image an anchor, semi-hard negatives "
                                         "mined on device, %s" % (cfg, name, args.batch, "bf16 operands + fp32 accumulate / master weights"
                                                                  if args.dtype == "bf16" else "fp32"),
                             "images_per_gpu": args.batch, "parallelism": "dp%d" % world}
            peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_FP32_MFMA_TFLOPS
            out["step_mfma_roofline_frac"] = round(images / world * flop_per_image / (peak * 1e12), 4)
        elif args.dtype == "bf16":  # not a BASELINE configuration: the headline network under the bf16 plan, for reference
            out["metric"] = "triplets/sec EFM 112x112 bs%d/GPU under the bf16 plan (NOT the BASELINE metric, which is fp32)" % args.batch
            out["dtype"] = "bf16"
            out["config"]["workload"] = out["config"]["workload"].replace("fp32", "bf16 operands + fp32 accumulate / master weights")
            out["step_mfma_roofline_frac"] = round(images / world * flop_per_image / (PEAK_BF16_MFMA_TFLOPS * 1e12), 4)
        else:
            out["roofline"] = dominant_kernel_roofline(tr, torch)
            # utilisation of the matrix pipe over the WHOLE step: executed matrix-core flops of all conv launches / step time / peak
            out["step_mfma_executed_frac"] = round(out["roofline"]["mfma_flop_executed_per_step"] / (ms * 1e-3) / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4)
            # the kernel selection that was timed, as data: feed it back with EFM_TUNING_FILE=<json with {"table": ...}> to reproduce
            out["tuning"] = {"source": tuning_src, "table": {k: [v["tune_fwd"], v["tune_dgrad"], v["tune_wgrad"], int(v["wino_fwd"]), int(v["wino_dgrad"])]
                                                               for k, v in tr.plan.tuning_table().items()},
                             "columns": ["tune_fwd", "tune_dgrad", "tune_wgrad", "wino_fwd", "wino_dgrad"]}
        if world == 1 and args.workload == "efm" and args.dtype == "f32" and not args.no_secondary:
            del tr
            torch.cuda.empty_cache()
            out["secondary"] = secondary_configs(torch, device, args.image)
        if world == 1 and not args.no_cpu_baseline and args.workload == "efm" and args.dtype == "f32":
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.image, torch, args.cpu_warmup, args.cpu_steps)
            if not args.no_host_loops:
                # the second, host-side comparison of SURVEY.md §8d: same leg (it times the oracle's literal loops on the host cores)
                out["host_loops"] = out["cpu_baseline"]["host_loops"] = host_loops(torch, device)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
