#!/bin/bash
# Ablation builds of the Winograd weight-gradient kernel (profiles/round2_wino_wgrad.md).
#   HERE, before gpurun:  tools/ww_ablate.sh            builds tools/ab/libefm_ww_<variant>.so = the in-tree objects with
#                                                       efm_wino_wgrad.hip recompiled under the variant's -DEFM_WW_* flags
#   on the box:           tools/ww_ablate.sh run [layers] [shape]   times every variant with and without the LDS-DMA staging
#                         tools/ww_ablate.sh stamps [layer] [shape] cycles per phase of the stamped build
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/improving_face_recognition_performance_using_triplet_loss_amd
declare -A FLAGS=( [nobarrier]="-DEFM_WW_NOBARRIER" [noload]="-DEFM_WW_NOLOAD" [noxform]="-DEFM_WW_NOXFORM" [nomfma]="-DEFM_WW_NOMFMA"
                   [nomfma_noload]="-DEFM_WW_NOMFMA -DEFM_WW_NOLOAD" [mfmaonly]="-DEFM_WW_NOLOAD -DEFM_WW_NOXFORM"
                   [mfmaonly_nobarrier]="-DEFM_WW_NOLOAD -DEFM_WW_NOXFORM -DEFM_WW_NOBARRIER" [stamps]="-DEFM_WW_STAMPS" )
ORDER="nobarrier noload noxform nomfma nomfma_noload mfmaonly mfmaonly_nobarrier stamps"
if [ "$1" = stamps ]; then  # in-kernel cycle stamps per phase (s_memtime), printed by the 8 waves of block 64
  EFM_LIB_PATH=$R/tools/ab/libefm_ww_stamps.so EFM_WINO_WGRAD_SHAPE=${3:-5x3} python $R/tools/conv_bench.py --tuned --what wwgrad --iters 1 --layers ${2:-conv3} 2>&1 | grep "^wave" | sort | awk 'NR%2==1'
  exit 0
fi
if [ "$1" = run ]; then
  L=${2:-conv3}
  S=${3:-5x3}
  for V in base $(echo $ORDER | sed 's/ stamps//'); do
    for D in 0 1; do
      unset EFM_LIB_PATH
      [ $V != base ] && export EFM_LIB_PATH=$R/tools/ab/libefm_ww_$V.so
      echo "== $V staging $([ $D = 1 ] && echo off || echo on)"
      EFM_WINO_DBG=$D EFM_WINO_WGRAD_SHAPE=$S python $R/tools/conv_bench.py --tuned --what wwgrad --iters 10 --layers $L 2>&1 | grep "k3"
    done
  done
  exit 0
fi
mkdir -p $R/tools/ab /tmp/ww_ab
for V in $ORDER; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize ${FLAGS[$V]} -c $P/csrc/efm_wino_wgrad.hip -o /tmp/ww_ab/ww_$V.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libefm_ww_$V.so $(ls $P/csrc/_obj/*.o | grep -v efm_wino_wgrad.o) /tmp/ww_ab/ww_$V.o
done
ls -la $R/tools/ab/
