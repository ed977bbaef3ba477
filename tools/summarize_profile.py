#!/usr/bin/env python
"""Condense gpurun_out/<tag> (written by tools/profile_round.sh) into profiles/<name>_*.{md,csv}."""
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
stats = max(glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")), key=os.path.getmtime)  # a re-used tag keeps older runs
shutil.copy(stats, "profiles/%s_kernel_stats.csv" % name)
rows = list(csv.DictReader(open(stats)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
bench = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
lines = ["# %s — rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline`" % name, "",
         "bench line under the profiler: %.2f ms/step, %.1f triplets/s, dominant-kernel roofline %s" %
         (bench["ms_per_step"], bench["value"], json.dumps(bench["roofline"])), "",
         "7 steps (2 warm-up + 5 timed) + 7 extra conv2-forward launches of the roofline probe + the start-up tiling autotune "
         "(every convolution x up to 7 tilings + 2 Winograd variants x 4 launches; EFM_AUTOTUNE=0 skips it); total GPU kernel time %.1f ms "
         "(weight-gradient and data-gradient kernels overlap on two streams, so per-kernel durations add up to more than wall time)" % (total / 1e6), "",
         "| % | total ms | calls | avg us | kernel |", "|---|---|---|---|---|"]
for r in rows[:30]:
    lines.append("| %.2f | %.3f | %s | %.1f | `%s` |" % (float(r["Percentage"]), float(r["TotalDurationNs"]) / 1e6, r["Calls"],
                                                       float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "")[:100]))


roof = bench["roofline"]
if roof.get("winograd"):
    KEY = "wino4_k<" if (roof["tune_fwd"] >> 8) & 3 == 2 else "wino_fwd_k<"
else:
    KEY = "conv_fwd_k<float, 1, 13, true, 1>"


def newest(pattern):
    f = glob.glob(pattern)
    return max(f, key=os.path.getmtime) if f else None


def pmc(dirname, counter):
    f = newest(os.path.join(src, dirname, "*", "*counter_collection.csv"))
    if not f:
        return None
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and KEY in r["Kernel_Name"]]


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
lines += ["", "## Dominant kernel: conv2 forward (66->198 3x3 @56x56, B=256, fused bias+MFM3+pool epilogue) = `%s...>` as chosen by the "
          "bench run's autotune; kernel-only runs (`tools/dominant_probe.py %d %d`), separate --pmc passes" % (KEY, int(bool(roof.get("winograd"))), roof["tune_fwd"]), ""]
kt = newest(os.path.join(src, "pmc_fetch", "*", "*kernel_trace.csv"))
if kt:
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt)) if KEY in r["Kernel_Name"]]
    if durs:
        lines.append("kernel-trace duration of that instance: mean %.1f us over %d launches (min %.1f) -> %.1f TFLOP/s algorithmic"
                     % (sum(durs) / len(durs), len(durs), min(durs), 188841590784.0 / (sum(durs) / len(durs)) / 1e6))
st_rows = [r for r in rows if KEY in r["Name"]]
if st_rows:
    lines.append("in the --stats run above: `%s` %s calls, average %.1f us (all layers that use this instance)"
                 % (st_rows[0]["Name"].replace("(anonymous namespace)::", "")[:60], st_rows[0]["Calls"], float(st_rows[0]["AverageNs"]) / 1e3))
if fetch and write:
    f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
    # gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> x2 (MI355X_MICROARCH.md §HBM);
    # units are KiB
    fetch_b, write_b = f_kb * 1024 * 2, w_kb * 1024
    lines += ["FETCH_SIZE %.0f KiB/launch (x2 gfx950 correction -> %.1f MB), WRITE_SIZE %.0f KiB/launch (%.1f MB); "
              "HBM bytes per launch = %.1f MB" % (f_kb, fetch_b / 1e6, w_kb, write_b / 1e6, (fetch_b + write_b) / 1e6),
              "algorithmic bytes: x 802816*68*4 = 218.4 MB read + pooled z 200704*132*4 = 106.0 MB + route 26.5 MB written + weights 0.5 MB"
              " (Winograd: U 3.3 MB) = 351 MB (an unfused kernel would write the 642 MB conv result instead)"]
    json.dump({"kernel": KEY, "fetch_bytes": fetch_b, "write_bytes": write_b, "traffic": fetch_b + write_b}, open("profiles/%s_traffic.json" % name, "w"))
sq = newest(os.path.join(src, "pmc_sq", "*", "*counter_collection.csv"))
if sq:
    agg = {}
    for r in csv.DictReader(open(sq)):
        if KEY in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    lines += ["", "## SQ counters, conv2 forward (mean per launch)", ""]
    for k, v in agg.items():
        lines.append("- %s = %.4g" % (k, sum(v) / len(v)))
open("profiles/%s_summary.md" % name, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
