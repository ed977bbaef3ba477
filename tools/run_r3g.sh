source tools/gpu_steps.sh r3g
cd /tmp
step prof_lc9 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lc9 -- python3 $R/bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 5 --warmup 2
step prof_deep 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_deep -- python3 $R/bench.py --workload deepcnn --dtype bf16 --batch 128 --steps 5 --warmup 2
cd $R
step make_tuning 600 python tools/make_tuning.py --rounds 3 --out gpurun_out/r3g/efm_tuning_new.json
