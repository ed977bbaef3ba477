#!/bin/bash
# SQ counters of the Winograd weight-gradient kernel on one layer (separate --pmc passes, kernel trace only)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/ww_pmc
rm -rf $OUT; mkdir -p $OUT
L=${1:-conv3}
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/p$i -- python3 $R/tools/conv_bench.py --tuned --what wwgrad --iters 3 --layers $L > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $CNT"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wino_wgrad_k" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-32s mean %14.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
