#!/bin/bash
# Run on the GPU box: per-kernel time of one bench configuration.  usage: trace_bench.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/stats.log 2>&1 || exit 1
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%7.3f ms %5.1f%% n=%5s avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print("total %.3f ms" % (tot / 1e6))
PY
