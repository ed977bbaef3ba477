#!/bin/bash
# Ablation builds of the Winograd weight-gradient kernel (run HERE, before gpurun): tools/ab/libefm_ww_{noload,nomfma}.so = the in-tree
# objects with efm_wino_wgrad.hip recompiled under -DEFM_WW_NOLOAD / -DEFM_WW_NOMFMA.  On the box: tools/ww_ablate.sh run [layers]
set -e
R=$(cd $(dirname $0)/.. && pwd)
P=$R/improving_face_recognition_performance_using_triplet_loss_amd
if [ "$1" = run ]; then
  L=${2:-conv3,conv4_res_r,conv2}
  for V in base noload nomfma nodma; do
    unset EFM_LIB_PATH EFM_WINO_DBG
    case $V in
      noload) export EFM_LIB_PATH=$R/tools/ab/libefm_ww_noload.so;;
      nomfma) export EFM_LIB_PATH=$R/tools/ab/libefm_ww_nomfma.so;;
      nodma) export EFM_WINO_DBG=1;;
    esac
    echo "== $V"
    python $R/tools/conv_bench.py --tuned --what wwgrad --iters 10 --layers $L 2>&1 | grep "k3"
  done
  exit 0
fi
mkdir -p $R/tools/ab /tmp/ww_ab
for V in NOLOAD NOMFMA; do
  v=$(echo $V | tr A-Z a-z)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -DEFM_WW_$V -c $P/csrc/efm_wino_wgrad.hip -o /tmp/ww_ab/ww_$v.o
  OBJS=$(ls $P/csrc/_obj/*.o | grep -v efm_wino_wgrad.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libefm_ww_$v.so $OBJS /tmp/ww_ab/ww_$v.o
done
ls -la $R/tools/ab/
