source tools/gpu_steps.sh r3e
export EFM_CONVB_MT_MAX=4
step lc9_mt4 400 python tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --iters 5 --what fwd,dgrad
