source tools/gpu_steps.sh r3k
cd /tmp
BF="python3 $R/tools/conv_bench.py --net lightcnn9 --dtype bf16 --batch 512 --layers conv2,conv3 --iters 3"
step bf_sq 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/bf_sq -- $BF
step bf_sq2 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/bf_sq2 -- $BF
