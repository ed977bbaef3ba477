#!/usr/bin/env python
"""Condense gpurun_out/<tag> (written by tools/profile_round3.sh) into profiles/round3_*.{md,csv,json}.   usage: summarize_profile3.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], "round3"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    f = glob.glob(pattern)
    return max(f, key=os.path.getmtime) if f else None


def clean(k):
    import subprocess
    k = k.strip('"')
    if k.startswith("_Z"):
        try:
            k = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
        except OSError:
            pass
    return k.replace("(anonymous namespace)::", "").replace("void ", "").replace("DF16b", "__bf16")


def stats_table(path, top=22, steps=7):
    """steps = None: total ms of the run (the fp32 run also holds the roofline probe's isolated launches, so a per-step figure would mislead)."""
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["| %% | %s | calls | avg us | kernel |" % ("ms / step" if steps else "total ms"), "|---|---|---|---|---|"]
    for r in rows[:top]:
        out.append("| %.2f | %.3f | %s | %.1f | `%s` |" % (float(r["Percentage"]), float(r["TotalDurationNs"]) / 1e6 / (steps or 1), r["Calls"],
                                                         float(r["AverageNs"]) / 1e3, clean(r["Name"])[:120]))
    return rows, total, out


def bench_line(fn):
    return json.loads(open(os.path.join(src, fn)).read().strip().splitlines()[-1])


lines = ["# round 3 — profiles of one MI355X (rocprofv3; un-profiled figures are in BENCH_r03 / DESIGN.md)", ""]
stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, "profiles/%s_kernel_stats.csv" % name)
rows, total, table = stats_table(stats, steps=None)
bench = bench_line("bench_line.json")
roof = bench["roofline"]
KEY = open(os.path.join(src, "dominant.txt")).read().strip()
lines += ["## fp32 headline: `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary` under --kernel-trace --stats", "",
          "bench line under the profiler: %.2f ms/step, %.1f triplets/s; kernel selection = %s; 7 steps + the roofline probe's isolated launches; "
          "total GPU kernel time %.1f ms (two backward streams overlap: per-kernel durations add up to more than wall time)"
          % (bench["ms_per_step"], bench["value"], bench["tuning"]["source"], total / 1e6), "",
          "`roofline` of that line: " + json.dumps({k: v for k, v in roof.items() if k not in ("families", "note")}), ""] + table
st = [r for r in rows if clean(r["Name"]).startswith(KEY)]
if st:
    lines += ["", "### Dominant kernel instance `%s`" % KEY, "",
              "kernel-trace average over all %s launches of the run: **%.1f us**; bench.py's live HIP-event figure (kernel alone on the chip): %.1f us "
              "per launch, %.2f TFLOP/s executed = %.3f of the fp32 MFMA peak (direct-convolution equivalent %.1f TFLOP/s)"
              % (st[0]["Calls"], float(st[0]["AverageNs"]) / 1e3, roof["ms_per_launch"] * 1e3, roof["achieved"], roof["frac"], roof["direct_equivalent_tflops"])]


def pmc(dirname, counter, key):
    f = newest(os.path.join(src, dirname, "*", "*counter_collection.csv"))
    if not f:
        return None
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and clean(r["Kernel_Name"]).startswith(key)]


fetch, write = pmc("pmc_fetch", "FETCH_SIZE", KEY), pmc("pmc_write", "WRITE_SIZE", KEY)
if fetch and write:
    f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
    fetch_b, write_b = f_kb * 1024 * 2, w_kb * 1024   # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide streams -> x2; KiB units
    lines += ["separate --pmc passes (`tools/family_probe.py`): FETCH_SIZE %.0f KiB/launch (x2 gfx950 correction -> %.1f MB), WRITE_SIZE %.0f KiB/launch "
              "(%.1f MB); **HBM-side bytes per launch = %.1f MB** vs %.1f MB algorithmic" % (f_kb, fetch_b / 1e6, w_kb, write_b / 1e6,
                                                                                           (fetch_b + write_b) / 1e6, roof["algorithmic_bytes_per_launch"] / 1e6)]
    json.dump({"kernel": KEY, "fetch_bytes": fetch_b, "write_bytes": write_b, "traffic": fetch_b + write_b, "launches_averaged": len(fetch)},
              open("profiles/%s_traffic.json" % name, "w"))
sq = newest(os.path.join(src, "pmc_sq", "*", "*counter_collection.csv"))
if sq:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sq)):
        if clean(r["Kernel_Name"]).startswith(KEY):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += ["", "SQ counters (mean per launch): " + ", ".join("%s = %.4g" % (k, sum(v) / len(v)) for k, v in sorted(agg.items()))]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "GRBM_GUI_ACTIVE" in agg:
        busy, gui = sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(agg["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(agg["GRBM_GUI_ACTIVE"]) / len(agg["GRBM_GUI_ACTIVE"])
        lines.append("matrix-pipe busy from the counters: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = %.3f" % (busy / (1024 * gui / 8)))
lines += ["", "### Per-instance table of the bench line (`roofline.families`: isolated HIP-event timing)", "",
          "| kernel | launches/step | ms/step | TF/s executed | frac of peak | TF/s direct-conv equivalent |", "|---|---|---|---|---|---|"]
for f in roof.get("families", []):
    lines.append("| `%s` | %d | %.3f | %.1f | %.3f | %.1f |" % (f["kernel"], f["launches_per_step"], f["ms_per_step"], f["tflops_mfma_executed"], f["frac"],
                                                             f["direct_equivalent_tflops"]))
for cfg, key, cmd in (("configs[2] LightCNN-9, 512 images, bf16", "lc9", "--workload lightcnn9 --dtype bf16 --batch 512"),
                      ("configs[4] deeper CNN, 128 images, bf16", "deep", "--workload deepcnn --dtype bf16 --batch 128")):
    bf = newest(os.path.join(src, "stats_" + key, "*", "*kernel_stats.csv"))
    if not bf:
        continue
    shutil.copy(bf, "profiles/%s_bf16_%s_kernel_stats.csv" % (name, "lightcnn9" if key == "lc9" else "deepcnn"))
    rows2, total2, table2 = stats_table(bf, 18)
    b2 = bench_line("bench_line_%s.json" % key)
    lines += ["", "## BASELINE %s: `python3 bench.py %s --steps 5 --warmup 2`" % (cfg, cmd), "",
              "%.2f ms/step under the profiler, %.0f triplets/s; total GPU kernel time %.1f ms over 7 steps (%.2f ms per step: the two backward "
              "streams overlap)" % (b2["ms_per_step"], b2["value"], total2 / 1e6, total2 / 7e6), ""] + table2
# bf16 counters
rowsb = collections.defaultdict(lambda: collections.defaultdict(list))
for d_, cs in (("bf_fetch", ["FETCH_SIZE"]), ("bf_write", ["WRITE_SIZE"]),
               ("bf_sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS",
                          "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"])):
    f = newest(os.path.join(src, d_, "*", "*counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k = clean(r["Kernel_Name"])
        if ("conv" in k or "mfm" in k or "wgrad" in k or "slab" in k) and r["Counter_Name"] in cs:
            rowsb[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
kt = newest(os.path.join(src, "bf_sq", "*", "*kernel_trace.csv"))
durs = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt)):
        durs[clean(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if rowsb:
    lines += ["", "## bf16 kernels of LightCNN-9's two widest layers (conv2 48->192 @56x56, conv3 96->384 @28x28; 512 images): --pmc passes over "
              "`tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --layers conv2,conv3 --iters 3`", "",
              "FETCH_SIZE x2 (gfx950 correction) and WRITE_SIZE in MB per launch (mean over the instance's launches: both layers, all passes); matrix-pipe "
              "busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "",
              "| kernel | launches | avg us (under --pmc) | HBM read MB | HBM write MB | MFMA busy | LDS conflict share |", "|---|---|---|---|---|---|---|"]
    mean = lambda v: sum(v) / len(v) if v else float("nan")  # noqa: E731
    for k, c in sorted(rowsb.items(), key=lambda kv: -sum(durs.get(kv[0], [0]))):
        busy = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * mean(c["GRBM_GUI_ACTIVE"]) / 8) if c["GRBM_GUI_ACTIVE"] else float("nan")
        conf = mean(c["SQ_LDS_BANK_CONFLICT"]) / mean(c["SQ_LDS_IDX_ACTIVE"]) if c["SQ_LDS_IDX_ACTIVE"] and mean(c["SQ_LDS_IDX_ACTIVE"]) > 0 else float("nan")
        lines.append("| `%s` | %d | %.1f | %.1f | %.1f | %.3f | %.3f |" % (k[:100], len(durs.get(k, [])), mean(durs.get(k, [])), mean(c["FETCH_SIZE"]) * 2048 / 1e6,
                                                                        mean(c["WRITE_SIZE"]) * 1024 / 1e6, busy, conf))
if rowsb:
    lines += ["", "Legend (rocprofv3 leaves some bf16 instance names mangled): `conv_fwd_kI__bf16Li<MT>ELi<NT>ELb1ELi<EPI>E` = `conv_fwd_k<__bf16, MT, NT, true, EPI>` "
              "(EPI 1 = fused bias + MFM2 (+ pool) forward, EPI 0 = plain forward / data gradient): `Li2ELi13ELb1ELi1E` = the fused forwards of conv2 / conv3, "
              "`Li2ELi13ELb1ELi0E` = their plain forwards (timed by conv_bench beside the fused ones), `Li2ELi6ELb1ELi0E` = conv3's data gradient (N = 96), the "
              "garbled `conv_fwd_k<bool _Accum, ..., E, 0>` = `<__bf16, 2, 3, true, 0>` = conv2's data gradient (N = 48).  Before the chunk-major K order "
              "(first profile of the round, `gpurun_out/r3prof`) those two data gradients read 6 203 MB and 2 800 MB from HBM per launch at 0.135 / 0.246 busy."]
for fn, title in (("per_layer.log", "fp32 EFM-29, benchmarked kernel selection, each launch alone (tools/conv_bench.py --tuned)"),
                  ("per_layer_lc9.log", "bf16 LightCNN-9, 512 images (tools/conv_bench.py --net lightcnn9 --dtype bf16)")):
    f = os.path.join(src, fn)
    if os.path.exists(f):
        shutil.copy(f, "profiles/%s_%s.txt" % (name, fn.replace(".log", "")))
open("profiles/%s_summary.md" % name, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
