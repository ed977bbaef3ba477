source tools/gpu_steps.sh r3j
step tests_bf16 900 python -m pytest tests/test_bf16_gpu.py tests/test_lightcnn9_gpu.py -m gpu -q --timeout 900
step lc9_wgrad 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5 --what wgrad
step bench_lc9 300 python bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 10 --warmup 3
export EFM_WGRAD2_NBUF=2
step lc9_wgrad_nbuf2 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5 --what wgrad
