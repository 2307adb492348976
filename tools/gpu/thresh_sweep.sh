#!/bin/bash
# tools/gpu/thresh_sweep.sh: lanes that must wait for the S block (RT_SCHED_THRESH) and the same once the tile is exhausted (RT_DRAIN_THRESH),
# swept through the diagnostic library (same kernel, knobs from the environment) on config #3 -> gpurun_out/r04ai/sweep.log
out=gpurun_out/r04ai; mkdir -p $out; : > $out/sweep.log
export RT_LIB_PATH=raytracing_c_amd/librt_hip_diag.so RT_EXP_ROUNDS=1 RT_EXP_RANKS="" RT_EXP_REPS=3
for st in 36 44 48 52 56; do
  for dt in 24 48; do
    RT_SCHED_THRESH=$st RT_DRAIN_THRESH=$dt timeout -k 10 120 python tools/exp_kernels.py "thresh$st-drain$dt:" 2>/dev/null | grep round >> $out/sweep.log || exit 1
  done
done
RT_SCHED_THRESH=48 RT_DRAIN_THRESH=48 timeout -k 10 120 python tools/exp_kernels.py "again48-48:" 2>/dev/null | grep round >> $out/sweep.log
cat $out/sweep.log
