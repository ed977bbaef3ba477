#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats of the bench command + HBM byte counters of the dominant
# kernel (conv2 forward) in separate --pmc passes, as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/<tag>/...
set -o pipefail
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
grep '"metric"' $OUT/stats.log > $OUT/bench_line.json
WINO=$(python3 -c "import json,sys; r=json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1])['roofline']; print(int(r['winograd']))")
TUNE=$(python3 -c "import json,sys; r=json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1])['roofline']; print(r['tune_fwd'])")
echo "dominant kernel: winograd=$WINO tune_fwd=$TUNE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/dominant_probe.py $WINO $TUNE 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/dominant_probe.py $WINO $TUNE 3 > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/dominant_probe.py $WINO $TUNE 3 > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed"
ls -R $OUT | head -40
