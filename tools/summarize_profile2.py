#!/usr/bin/env python
"""Condense gpurun_out/<tag> (written by tools/profile_round2.sh) into profiles/<name>_*.{md,csv,json}."""
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    f = glob.glob(pattern)
    return max(f, key=os.path.getmtime) if f else None


def clean(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "")


def stats_table(path, top=24):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["| % | total ms | calls | avg us | kernel |", "|---|---|---|---|---|"]
    for r in rows[:top]:
        out.append("| %.2f | %.3f | %s | %.1f | `%s` |" % (float(r["Percentage"]), float(r["TotalDurationNs"]) / 1e6, r["Calls"],
                                                         float(r["AverageNs"]) / 1e3, clean(r["Name"])[:110]))
    return rows, total, out


stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, "profiles/%s_kernel_stats.csv" % name)
rows, total, table = stats_table(stats)
bench = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
roof = bench["roofline"]
KEY = open(os.path.join(src, "dominant.txt")).read().strip()
lines = ["# %s — rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary`" % name, "",
         "bench line under the profiler: %.2f ms/step, %.1f triplets/s; kernel selection = %s" % (bench["ms_per_step"], bench["value"], bench["tuning"]["source"]), "",
         "`roofline` of that line: " + json.dumps({k: v for k, v in roof.items() if k not in ("families", "note")}), "",
         "7 steps (2 warm-up + 5 timed) + the roofline probe's isolated launches (every distinct conv launch of the step x 4); total GPU kernel time "
         "%.1f ms (weight-gradient and data-gradient kernels overlap on two streams, so per-kernel durations add up to more than wall time)" % (total / 1e6), ""]
lines += table
st = [r for r in rows if clean(r["Name"]).startswith(KEY)]
if st:
    lines += ["", "## Dominant kernel instance `%s` (most time per step; all layers that resolve to it)" % KEY, "",
              "kernel-trace average over all %s launches of the run above: **%.1f us**; bench.py's live HIP-event figure (kernel alone on the chip): "
              "%.1f us per launch, %.2f TFLOP/s algorithmic = %.3f of peak, matrix pipe busy %.3f"
              % (st[0]["Calls"], float(st[0]["AverageNs"]) / 1e3, roof["ms_per_launch"] * 1e3, roof["achieved"], roof["frac"], roof["mfma_busy_frac"])]


def pmc(dirname, counter):
    f = newest(os.path.join(src, dirname, "*", "*counter_collection.csv"))
    if not f:
        return None
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and clean(r["Kernel_Name"]).startswith(KEY)]


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
kt = newest(os.path.join(src, "pmc_fetch", "*", "*kernel_trace.csv"))
if kt:
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if clean(r["Kernel_Name"]).startswith(KEY)]
    if durs:
        lines.append("kernel-only probe (`tools/family_probe.py \"%s\"`: the step's %d launches of this instance, x3): mean %.1f us per launch"
                     % (KEY, roof["launches_per_step"], sum(durs) / len(durs)))
if fetch and write:
    f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
    # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md, HBM / rocprofv3); units are KiB
    fetch_b, write_b = f_kb * 1024 * 2, w_kb * 1024
    lines += ["separate --pmc passes: FETCH_SIZE %.0f KiB/launch (x2 gfx950 correction -> %.1f MB), WRITE_SIZE %.0f KiB/launch (%.1f MB); "
              "**HBM-side bytes per launch = %.1f MB** (average over the instance's launches)" % (f_kb, fetch_b / 1e6, w_kb, write_b / 1e6, (fetch_b + write_b) / 1e6)]
    json.dump({"kernel": KEY, "fetch_bytes": fetch_b, "write_bytes": write_b, "traffic": fetch_b + write_b,
               "launches_averaged": len(fetch)}, open("profiles/%s_traffic.json" % name, "w"))
sq = newest(os.path.join(src, "pmc_sq", "*", "*counter_collection.csv"))
if sq:
    agg = {}
    for r in csv.DictReader(open(sq)):
        if clean(r["Kernel_Name"]).startswith(KEY):
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    lines += ["", "SQ counters (mean per launch): " + ", ".join("%s = %.4g" % (k, sum(v) / len(v)) for k, v in sorted(agg.items()))]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg and "GRBM_GUI_ACTIVE" in agg:
        busy = sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(agg["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(agg["GRBM_GUI_ACTIVE"]) / len(agg["GRBM_GUI_ACTIVE"])
        lines.append("matrix-pipe busy from the counters: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = %.3f" % (busy / (1024 * gui / 8)))
lines += ["", "## Per-instance table of the bench line (`roofline.families`: isolated HIP-event timing, algorithmic vs executed flops)", "",
          "| kernel | launches/step | ms/step | TF/s algorithmic | TF/s executed | frac algorithmic | MFMA busy |", "|---|---|---|---|---|---|---|"]
for f in roof.get("families", []):
    lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.3f | %.3f |" % (f["kernel"], f["launches_per_step"], f["ms_per_step"], f["tflops_algorithmic"],
                                                                   f["tflops_mfma_executed"], f["frac_algorithmic"], f["mfma_busy_frac"]))
bf = newest(os.path.join(src, "stats_bf16", "*", "*kernel_stats.csv"))
if bf:
    shutil.copy(bf, "profiles/%s_bf16_lightcnn9_kernel_stats.csv" % name)
    rows2, total2, table2 = stats_table(bf, 16)
    b2 = json.loads(open(os.path.join(src, "bench_line_bf16.json")).read().strip().splitlines()[-1])
    lines += ["", "## BASELINE configs[2]: `python3 bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 5 --warmup 2`", "",
              "%.2f ms/step, %.0f triplets/s, step at %.3f of the bf16 MFMA peak; total GPU kernel time %.1f ms over 7 steps"
              % (b2["ms_per_step"], b2["value"], b2["step_mfma_roofline_frac"], total2 / 1e6), ""] + table2
open("profiles/%s_summary.md" % name, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
