#!/bin/bash
# rocprofv3 kernel trace of the Winograd weight gradient on a few EFM-29 layers (kernel vs slab reduction vs bias gradient)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/ww_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py --tuned --what wwgrad --iters 10 --layers "${1:-conv2,conv3,conv4_res_r,conv4_res,conv5_res}" > $OUT/run.log 2>&1
cat $OUT/run.log | tail -8
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-90s calls %5s avg %9.1f us total %9.1f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
