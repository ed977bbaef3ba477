source tools/gpu_steps.sh r3m
step tests_all 1000 python -m pytest tests -m gpu -q --timeout 900
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step lc9_wgrad 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5 --what wgrad
step bench 900 python bench.py
