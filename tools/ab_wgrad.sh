# A/B of two builds on the same box: tools/ab/libefm_old.so (EFM_LIB_PATH) vs the in-tree library.
set -o pipefail
mkdir -p gpurun_out/ab
for CFG in "0 0" "3840 1" "2560 2"; do
  set -- $CFG
  for L in old new; do
    if [ $L = old ]; then export EFM_LIB_PATH=$PWD/tools/ab/libefm_old.so; else unset EFM_LIB_PATH; fi
    if [ $1 != 0 ]; then export EFM_WGRAD_BLOCKS=$1 EFM_WGRAD_KPW=$2; else unset EFM_WGRAD_BLOCKS EFM_WGRAD_KPW; fi
    echo "== $L blocks $1 KPW $2" >> gpurun_out/ab/wgrad.txt
    python tools/conv_bench.py --what wgrad --iters 10 >> gpurun_out/ab/wgrad.txt 2>&1
  done
done
grep -E "==|TOTAL wgrad" gpurun_out/ab/wgrad.txt
