#!/bin/bash
# Round-3 profile, run on the GPU box through gpurun: usage  profile_round3.sh <tag>   -> gpurun_out/<tag>/...
#  (1) kernel-trace stats of the default bench command (fp32 headline);  (2) HBM byte counters + SQ counters of the dominant fp32
#  kernel instance in SEPARATE --pmc passes (MI355X_MICROARCH.md);  (3) kernel-trace stats of BASELINE configs[2] / [4] (bf16);
#  (4) --pmc passes (FETCH_SIZE | WRITE_SIZE | MFMA busy + LDS conflicts) over the bf16 kernels of the two biggest LightCNN-9 layers.
source tools/gpu_steps.sh ${1:-r3prof}
cd /tmp
step stats 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary
grep '"metric"' $OUT/stats.log > $OUT/bench_line.json
KERNEL=$(python3 -c "import json; r=json.loads(open('$OUT/bench_line.json').read().strip().splitlines()[-1])['roofline']; print(r['kernel'].split(':')[0])")
echo "$KERNEL" > $OUT/dominant.txt
step pmc_fetch 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/family_probe.py "$KERNEL" 2
step pmc_write 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/family_probe.py "$KERNEL" 2
step pmc_sq 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/family_probe.py "$KERNEL" 2
step stats_lc9 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lc9 -- python3 $R/bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 5 --warmup 2
grep '"metric"' $OUT/stats_lc9.log > $OUT/bench_line_lc9.json
step stats_deep 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_deep -- python3 $R/bench.py --workload deepcnn --dtype bf16 --batch 128 --steps 5 --warmup 2
grep '"metric"' $OUT/stats_deep.log > $OUT/bench_line_deep.json
BF="python3 $R/tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --layers conv2,conv3 --iters 3"
step bf_fetch 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/bf_fetch -- $BF
step bf_write 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bf_write -- $BF
step bf_sq 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/bf_sq -- $BF
cd $R
step per_layer 400 python tools/conv_bench.py --tuned --what fwd,dgrad,wgrad --iters 10
step per_layer_lc9 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 10
ls $OUT
