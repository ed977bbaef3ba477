source tools/gpu_steps.sh r3n
step tests_bf16 900 python -m pytest tests/test_bf16_gpu.py tests/test_lightcnn9_gpu.py tests/test_scripts_gpu.py -m gpu -q --timeout 900
step lc9_dgrad 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5 --what dgrad
step bench_lc9 300 python bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 10 --warmup 3
step bench_deep 300 python bench.py --workload deepcnn --dtype bf16 --batch 128 --steps 10 --warmup 3
