source tools/gpu_steps.sh r3a
step tests 900 python -m pytest tests -m gpu -q -x --timeout 900
step bench 900 python bench.py
step iter 300 python tools/iter_bench.py --threads 0,1,4,8,16
step iter_rgb 300 python tools/iter_bench.py --threads 0,8,16 --channels 3 --size 128 --crop 112
step lc9_layers 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5
