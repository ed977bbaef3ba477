# A/B of two builds on the same box: tools/ab/libefm_old.so (EFM_LIB_PATH) vs the in-tree library, weight gradient per layer.
# Make the baseline first: check out / stash to the older tree, `python -m improving_face_recognition_performance_using_triplet_loss_amd.build --force`,
# copy libefm_hip.so to tools/ab/libefm_old.so (git-ignored, travels with gpurun), return to the new tree and rebuild.
set -o pipefail
mkdir -p gpurun_out/ab
rm -f gpurun_out/ab/wgrad.txt
python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad or conv or bias" > gpurun_out/ab/wgrad_tests.log 2>&1 || { tail -30 gpurun_out/ab/wgrad_tests.log; exit 1; }
tail -2 gpurun_out/ab/wgrad_tests.log
for CFG in "0 0" "3840 1"; do
  set -- $CFG
  for L in old new; do
    if [ $L = old ]; then export EFM_LIB_PATH=$PWD/tools/ab/libefm_old.so; else unset EFM_LIB_PATH; fi
    if [ $1 != 0 ]; then export EFM_WGRAD_BLOCKS=$1 EFM_WGRAD_KPW=$2; else unset EFM_WGRAD_BLOCKS EFM_WGRAD_KPW; fi
    echo "== $L blocks $1 KPW $2" >> gpurun_out/ab/wgrad.txt
    python tools/conv_bench.py --what wgrad --iters 10 >> gpurun_out/ab/wgrad.txt 2>&1
  done
done
grep -E "==|TOTAL wgrad" gpurun_out/ab/wgrad.txt
