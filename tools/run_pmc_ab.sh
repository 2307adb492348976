set -u
mkdir -p gpurun_out/r02c
for K in 3 5; do
  RT_KERNEL=$K bash tools/pmc_gpu.sh k${K} "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" > gpurun_out/r02c/pmc_k$K.log 2>&1
done
timeout -k 10 300 python tools/exp_kernels.py "k3id:RT_KERNEL=3,RT_ORDER=identity" "k5id:RT_KERNEL=5,RT_ORDER=identity" "k5:RT_KERNEL=5" "k5t32:RT_KERNEL=5,RT_SCHED_THRESH=32" > gpurun_out/r02c/ab.log 2>&1
tail -4 gpurun_out/r02c/ab.log
