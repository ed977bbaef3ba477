#!/bin/bash
# Run on the GPU box (through gpurun): (1) kernel-trace stats of the default bench command, (2) HBM byte counters and SQ counters of
# the dominant kernel instance (the one bench.py's `roofline` names) in separate --pmc passes, as MI355X_MICROARCH.md prescribes,
# (3) kernel-trace stats of the bf16 secondary configuration.  Output: gpurun_out/<tag>/...   usage: profile_round2.sh <tag>
set -o pipefail
TAG=${1:-r2prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
grep '"metric"' $OUT/stats.log > $OUT/bench_line.json
KERNEL=$(python3 -c "import json; r=json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1])['roofline']; print(r['kernel'].split(':')[0])")
echo "dominant kernel instance: $KERNEL"
echo "$KERNEL" > $OUT/dominant.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/family_probe.py "$KERNEL" 2 > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/family_probe.py "$KERNEL" 2 > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/family_probe.py "$KERNEL" 2 > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bf16 -- python3 $R/bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 5 --warmup 2 > $OUT/stats_bf16.log 2>&1 || { tail -5 $OUT/stats_bf16.log; exit 1; }
grep '"metric"' $OUT/stats_bf16.log > $OUT/bench_line_bf16.json
ls $OUT
