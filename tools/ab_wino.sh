# A/B of two builds on the same box: tools/ab/libefm_old.so (EFM_LIB_PATH) vs the in-tree library, Winograd forward / data gradient.
# Make the baseline first: check out / stash to the older tree, `python -m improving_face_recognition_performance_using_triplet_loss_amd.build --force`,
# copy libefm_hip.so to tools/ab/libefm_old.so (git-ignored, travels with gpurun), return to the new tree and rebuild.
set -o pipefail
mkdir -p gpurun_out/ab
rm -f gpurun_out/ab/wino.txt
python -m pytest tests/test_winograd_gpu.py -x -q > gpurun_out/ab/wino_tests.log 2>&1 || { tail -20 gpurun_out/ab/wino_tests.log; exit 1; }
tail -2 gpurun_out/ab/wino_tests.log
for V in 1 2; do
  for L in old new; do
    if [ $L = old ]; then export EFM_LIB_PATH=$PWD/tools/ab/libefm_old.so; else unset EFM_LIB_PATH; fi
    echo "== $L variant $V" >> gpurun_out/ab/wino.txt
    EFM_WINO_VARIANT=$((V==1?8:4)) python tools/conv_bench.py --what wfwd,wdgrad --iters 10 >> gpurun_out/ab/wino.txt 2>&1
  done
done
grep -E "==|TOTAL w" gpurun_out/ab/wino.txt
