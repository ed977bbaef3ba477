#!/bin/bash
# PMC counters for one kernel instance of the step: pmc_family.sh <tag> "<kernel prefix>" "<counters pass 1>" ["<counters pass 2>" ...]
set -o pipefail
TAG=$1; KER=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
i=0
for CNT in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/p$i -- python3 $R/tools/family_probe.py "$KER" 1 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; continue; }
  F=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$KER" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if k.startswith(sys.argv[2]):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: "%.4g" % (sum(v) / len(v)) for c, v in agg.items()}, "launches", len(next(iter(agg.values()))) if agg else 0)
PY
done
