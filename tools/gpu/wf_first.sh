#!/bin/bash
# first GPU run of the wavefront pipeline: frame parity (both geometries), then an interleaved timing A/B
out=gpurun_out/${1:-r03a}; mkdir -p $out
export RT_PIPELINE=wf
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "frame_bit_exact" > $out/parity_wf0.log 2>&1
rc=$?; tail -5 $out/parity_wf0.log
if [ $rc -ne 0 ]; then echo "parity wf0 failed rc=$rc"; exit 1; fi
RT_WF_GEOMETRY=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "frame_bit_exact" > $out/parity_wf1.log 2>&1
rc=$?; tail -3 $out/parity_wf1.log
if [ $rc -ne 0 ]; then echo "parity wf1 failed rc=$rc"; exit 1; fi
unset RT_PIPELINE
RT_EXP_RANKS="" RT_EXP_REPS=3 timeout -k 10 300 python tools/exp_kernels.py "k5:" "wf0:RT_PIPELINE=wf" "wf1:RT_PIPELINE=wf,RT_WF_GEOMETRY=1" > $out/ab.log 2>&1
cat $out/ab.log
