source tools/gpu_steps.sh r3o
step tests_all 1000 python -m pytest tests -m gpu -q --timeout 900
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench 900 python bench.py
