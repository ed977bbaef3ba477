source tools/gpu_steps.sh r3h
step tests_all 1000 python -m pytest tests -m gpu -q --timeout 900
step bench 900 python bench.py --no-cpu-baseline --steps 10
step lc9_layers 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5 --what wgrad
cd /tmp
step prof_lc9 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lc9 -- python3 $R/bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 5 --warmup 2
