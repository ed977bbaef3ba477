#!/usr/bin/env python
"""Per-layer timing of the EFM-29 convolution kernels (HIP events on the launch stream).

    python tools/conv_bench.py [--batch 256] [--image 112] [--layers conv2,conv3] [--iters 5] [--what fwd,dgrad,wgrad]
Prints ms and achieved TFLOP/s (unpadded algorithmic flops) per layer and pass, and the step total.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from improving_face_recognition_performance_using_triplet_loss_amd import efm_symbol, ops  # noqa: E402
from improving_face_recognition_performance_using_triplet_loss_amd.plan import Plan  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--image", type=int, default=112)
    ap.add_argument("--layers", default="")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--net", choices=["efm", "lightcnn9"], default="efm")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--tuned", action="store_true", help="apply the committed kernel-selection table (tuning/*.json): 'fwd' / 'dgrad' then "
                                                         "time whatever kernel the table names for the layer, Winograd included")
    a = ap.parse_args()
    net = efm_symbol.embedding_net() if a.net == "efm" else efm_symbol.lightcnn9_embedding_net()
    plan = Plan(net, (a.batch, 3, a.image, a.image), fuse=True, dtype=a.dtype)
    if a.tuned:
        from improving_face_recognition_performance_using_triplet_loss_amd import tuning
        table, src = tuning.load(a.net, a.batch, a.image, a.dtype)
        assert table is not None, "no committed table for this configuration"
        plan.apply_tuning(table)
        print("# kernel selection:", src)
    bf = a.dtype == "bf16"
    want = set(a.layers.split(",")) if a.layers else None
    what = a.what.split(",")
    tot = {k: 0.0 for k in what}
    totf = {k: 0.0 for k in what}
    seen = set()
    for st in plan.steps:
        if st.op != "conv" or (want and st.pname not in want):
            continue
        d = st.desc
        key = (d.hin, d.win, d.cin, d.cout, d.kh, d.pad_h)
        dup = key in seen
        seen.add(key)
        x = torch.rand((d.batch, d.hin, d.win, d.cin_p), device="cuda") - 0.5
        x[..., d.cin:] = 0
        dy = torch.rand((d.batch, d.hout, d.wout, d.cout_p), device="cuda") - 0.5
        dy[..., d.cout:] = 0
        w = torch.rand((d.n_pad16, d.k_pad), device="cuda") - 0.5
        b = torch.zeros(d.n_pad16, device="cuda")
        wd = torch.rand((d.dn_pad16, d.dk_pad), device="cuda") - 0.5
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        if bf:
            c8 = lambda c: (c + 7) & ~7  # noqa: E731
            xb = torch.zeros((d.batch, d.hin, d.win, c8(d.cin)), device="cuda", dtype=torch.bfloat16)
            xb[..., :d.cin] = x[..., :d.cin].bfloat16()
            dyb = torch.zeros((d.batch, d.hout, d.wout, c8(d.cout)), device="cuda", dtype=torch.bfloat16)
            dyb[..., :d.cout] = dy[..., :d.cout].bfloat16()
            wb, wdb = ops.convb_cast_weights(d, w)
        flops = 2.0 * d.batch * d.hout * d.wout * d.cout * d.cin * d.kh * d.kw
        line = "%-14s %3dx%-3d %3d->%-3d k%d M=%-7d" % (st.pname, d.hin, d.win, d.cin, d.cout, d.kh, d.batch * d.hout * d.wout)
        for k in what:
            if k in ("dgrad", "wdgrad") and not st.inputs[0].needs_grad:
                line += "  dgrad   --           "
                continue
            if bf:
                if k == "fwd":
                    ms = timeit(lambda: ops.convb_fwd(d, xb, wb, b), a.iters)
                    if st.epi is not None:
                        ms2 = timeit(lambda: ops.convb_mfm_fwd(d, xb, wb, b, st.epi["ways"], st.epi["order"], st.epi["pool"]), a.iters)
                        line += "  fused %7.3f ms %6.1f TF |" % (ms2, flops / ms2 / 1e9)
                elif k == "dgrad":
                    ms = timeit(lambda: ops.convb_bwd_data(d, dyb, wdb), a.iters)
                else:
                    ms = timeit(lambda: ops.convb_bwd_weight(d, xb, dyb, dw=dw, dbias=db), a.iters)
            elif k in ("wfwd", "wdgrad", "wwgrad"):
                if not ops.wino_supported(d):
                    continue
                if k == "wwgrad":
                    ms = timeit(lambda: ops.wino_bwd_weight(d, x, dy, dw=dw, dbias=db), a.iters)
                elif k == "wfwd":
                    u = ops.wino_make_u(d, w)
                    ms = timeit(lambda: ops.wino_fwd(d, x, u, b, out=y), a.iters)
                else:
                    u = ops.wino_make_u(d, w, dgrad=True)
                    ms = timeit(lambda: ops.wino_bwd_data(d, dy, u, out=dx), a.iters)
            elif k == "fwd" and a.tuned:   # the launch the plan makes for this layer's forward
                e = st.epi
                if e is not None and getattr(st, "wino_fwd", False):
                    u = ops.wino_mfm_make_u(d, w, e["ways"])
                    ms = timeit(lambda: ops.wino_mfm_fwd(d, x, u, b, e["ways"], e["order"], e["pool"]), a.iters)
                elif e is not None:
                    ms = timeit(lambda: ops.conv_mfm_fwd(d, x, w, b, e["ways"], e["order"], e["pool"]), a.iters)
                elif getattr(st, "wino_fwd", False):
                    u = ops.wino_make_u(d, w)
                    ms = timeit(lambda: ops.wino_fwd(d, x, u, b, out=y), a.iters)
                else:
                    ms = timeit(lambda: ops.conv_fwd(d, x, w, b, out=y), a.iters)
                line += " [%s]" % ops.conv_kernel_info(d, (5 if getattr(st, "wino_fwd", False) else 1) if e is not None else (4 if getattr(st, "wino_fwd", False) else 0),
                                                       e["ways"] if e else 0, e["pool"] if e else False)[0]
            elif k == "dgrad" and a.tuned:
                if getattr(st, "wino_dgrad", False):
                    u = ops.wino_make_u(d, w, dgrad=True)
                    ms = timeit(lambda: ops.wino_bwd_data(d, dy, u, out=dx), a.iters)
                else:
                    ms = timeit(lambda: ops.conv_bwd_data(d, dy, wd, out=dx), a.iters)
                line += " [%s]" % ops.conv_kernel_info(d, 6 if getattr(st, "wino_dgrad", False) else 2)[0]
            elif k == "fwd":
                ms = timeit(lambda: ops.conv_fwd(d, x, w, b, out=y), a.iters)
                if st.epi is not None:
                    ms2 = timeit(lambda: ops.conv_mfm_fwd(d, x, w, b, st.epi["ways"], st.epi["order"], st.epi["pool"]), a.iters)
                    line += "  fused %7.3f ms %6.1f TF |" % (ms2, flops / ms2 / 1e9)
            elif k == "dgrad":
                ms = timeit(lambda: ops.conv_bwd_data(d, dy, wd, out=dx), a.iters)
            else:
                ms = timeit(lambda: ops.conv_bwd_weight(d, x, dy, dw=dw, dbias=db), a.iters)
            tot[k] += ms
            totf[k] += flops
            line += "  %s %7.3f ms %6.1f TF" % (k, ms, flops / ms / 1e9)
        print(line + ("  (dup shape)" if dup else ""), flush=True)
    for k in what:
        if tot[k]:
            print("TOTAL %-6s %8.3f ms  %6.1f TF/s" % (k, tot[k], totf[k] / tot[k] / 1e9))
    print("TOTAL all %8.3f ms" % sum(tot.values()))


if __name__ == "__main__":
    main()
