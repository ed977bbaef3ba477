set -o pipefail
mkdir -p gpurun_out/r2d
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -x -q > gpurun_out/r2d/tests.log 2>&1 || { tail -30 gpurun_out/r2d/tests.log; exit 1; }
tail -3 gpurun_out/r2d/tests.log
for B in 768 1024 1536 2560 3840; do
  for K in 1 2; do
    echo "== blocks $B KPW $K" >> gpurun_out/r2d/wgrad_sweep.txt
    EFM_WGRAD_BLOCKS=$B EFM_WGRAD_KPW=$K python tools/conv_bench.py --what wgrad >> gpurun_out/r2d/wgrad_sweep.txt 2>&1
  done
done
grep -E "==|TOTAL wgrad" gpurun_out/r2d/wgrad_sweep.txt
