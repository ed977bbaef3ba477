source tools/gpu_steps.sh r3f
step tests_a 900 python -m pytest tests/test_bf16_gpu.py tests/test_lightcnn9_gpu.py -m gpu -q --timeout 900
step lc9_layers 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5 --what wgrad
step bench 900 python bench.py --no-cpu-baseline --steps 10
step tests_rest 900 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_bf16_gpu.py --deselect tests/test_lightcnn9_gpu.py
