source tools/gpu_steps.sh r3i
step two_streams_on 300 python bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 10 --warmup 3
export EFM_TWO_STREAMS=0
step two_streams_off 300 python bench.py --workload lightcnn9 --dtype bf16 --batch 512 --steps 10 --warmup 3
step deep_off 300 python bench.py --workload deepcnn --dtype bf16 --batch 128 --steps 10 --warmup 3
