#!/bin/bash
# Run on the GPU box: PMC counters for one conv_bench configuration.  usage: pmc_probe.sh <tag> "<counters>" <conv_bench args...>
set -o pipefail
TAG=$1; CNT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/pmc -- python3 $R/tools/conv_bench.py "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
F=$(find $OUT/pmc -name '*counter_collection.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "at::" in k or "elementwise" in k:
        continue
    print(k, {c: "%.3g" % (sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
